"""Cube head plugin (reference cubercnn/modeling/roi_heads/cube_head.py:18-208).
The FC stack and the five output linears run as MFMA GEMMs inside ``ovm_cube_forward``; this class
only validates the configuration the native kernels implement (shared FC, 6d pose, class-agnostic)."""
from __future__ import annotations

from ...registry import ROI_CUBE_HEAD_REGISTRY


@ROI_CUBE_HEAD_REGISTRY.register()
class CubeHead:
    def __init__(self, cfg, input_shape=None):
        H = cfg.MODEL.ROI_CUBE_HEAD
        self.num_classes = cfg.MODEL.ROI_HEADS.NUM_CLASSES
        self.use_conf = H.USE_CONFIDENCE
        self.z_type = H.Z_TYPE
        self.pose_type = H.POSE_TYPE
        self.cluster_bins = H.CLUSTER_BINS
        self.shared_fc = H.SHARED_FC
        self.use_prior = H.DIMS_PRIORS_ENABLED
        if not self.shared_fc or self.pose_type != "6d" or self.use_prior or self.cluster_bins > 1:
            raise NotImplementedError("native CubeHead: SHARED_FC, POSE_TYPE '6d', no dims priors, 1 cluster bin "
                                      "(reference configs/Base.yaml:71-86)")


def build_cube_head(cfg, input_shape=None):
    return ROI_CUBE_HEAD_REGISTRY.get(cfg.MODEL.ROI_CUBE_HEAD.NAME)(cfg, input_shape)
