"""Native GroundingDINO branch (scope row a10): the text-conditioned 2D detector that
``ROIHeads3DGDINO`` calls (reference cubercnn/modeling/roi_heads/roi_heads_gdino.py:186,
IDEA-Research/GroundingDINO @856dde2 - third-party, not in the reference tree).

The network is sequenced on the host, module by module, in the structure of the published model; every
arithmetic op is a libovm3d call (ovm_g_*: MFMA GEMM projections, LayerNorm, batched matmul, softmax,
gathers, GroupNorm, multi-scale deformable sampling, ...). Weights use the parameter names of the
Hugging Face port (``transformers`` ``GroundingDinoForObjectDetection``), which is also the independent
CPU implementation the parity tests compare against.
"""
