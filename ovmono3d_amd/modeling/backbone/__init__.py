from .dino import DINOBackbone, SimpleFeaturePyramidWithDepth, build_dino_backbone  # noqa: F401
