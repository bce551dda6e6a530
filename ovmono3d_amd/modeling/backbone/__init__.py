from .dino import DINOBackbone, SimpleFeaturePyramidWithDepth, build_dino_backbone  # noqa: F401
from .clip import CLIPBackbone, SimpleFeaturePyramid, build_clip_backbone  # noqa: F401
from .mae import MAEBackbone, build_mae_backbone  # noqa: F401
from .midas import MIDASBackbone, build_midas_backbone  # noqa: F401
from .sam import SAMBackbone, build_sam_backbone  # noqa: F401
