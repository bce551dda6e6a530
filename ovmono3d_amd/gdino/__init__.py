"""Native GroundingDINO branch (scope row a10): the text-conditioned 2D detector that
``ROIHeads3DGDINO`` calls (reference cubercnn/modeling/roi_heads/roi_heads_gdino.py:186,
IDEA-Research/GroundingDINO @856dde2 - third-party, not in the reference tree).

The whole network - BERT, Swin, fusion encoder, two-stage selection, decoder - is sequenced inside libovm3d
(``ovm_gdino_create`` / ``ovm_gdino_forward``, csrc/gdino.hip); this package is its ctypes front end (``engine.py``), the
architecture record (``config.py``) and the detector object the RoI head plugs in (``detector.py``: tokenisation, the
upstream -> port parameter-name map). Weights use the parameter names of the Hugging Face port of the model, which is also the
independent CPU implementation the parity tests compare against (tests/, never imported from here).
"""
