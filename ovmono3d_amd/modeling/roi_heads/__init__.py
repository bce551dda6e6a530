from .cube_head import CubeHead, build_cube_head  # noqa: F401
from .roi_heads import ROIHeads3D, build_roi_heads  # noqa: F401
from .roi_heads_gdino import ROIHeads3DGDINO  # noqa: F401
