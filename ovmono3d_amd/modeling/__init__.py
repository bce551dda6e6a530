from .backbone import build_dino_backbone  # noqa: F401
from .meta_arch import RCNN3D, build_model, build_backbone  # noqa: F401
from .proposal_generator import RPNWithIgnore  # noqa: F401
from .roi_heads import ROIHeads3D, ROIHeads3DGDINO, CubeHead, build_roi_heads, build_cube_head  # noqa: F401
