"""CPU oracle for the OVMono3D-LIFT inference path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain fp32 PyTorch-CPU restatement of the reference algorithm (nightgoodl/ovmono3d), each
function citing the reference file:line it follows. Only ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` may import this package; nothing under
``ovmono3d_amd/`` does.

PARITY PINNING STATUS: the reference ships no tests with value assertions, no golden vectors and
no result files for this path, and its own code cannot be imported here
(detectron2 / pytorch3d / torchvision / groundingdino / cv2 are not installed - ordinary
ModuleNotFoundError, nothing was denied). Therefore:
  * reference-owned arithmetic (cube decode, allocentric pose, cuboid corners, virtual depth,
    score fusion, fast-rcnn inference glue, GDINO phrase-logit glue) is restated line by line;
  * third-party arithmetic (DINOv2 blocks, detectron2 SFP/RPN/ROIPooler/box2box, torchvision
    ROIAlign/NMS, pytorch3d rotation conversions) is restated from the published algorithms and
    cross-checked where an independent implementation exists in this container
    (HF ``Dinov2Model`` layers, ``torch.nn.functional`` primitives) - see tests/test_oracle_*.py;
  * everything else is "parity unpinned" and says so where it is tested.
"""
