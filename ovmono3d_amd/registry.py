"""Name -> builder registries with the reference's names (detectron2 ``Registry`` semantics):
BACKBONE_REGISTRY (reference dino.py:123), META_ARCH_REGISTRY (rcnn3d.py:25), ROI_HEADS_REGISTRY
(roi_heads.py:39, roi_heads_gdino.py:26), PROPOSAL_GENERATOR_REGISTRY (rpn.py:19),
ROI_CUBE_HEAD_REGISTRY (cube_head.py:18). Selected by MODEL.BACKBONE.NAME, MODEL.META_ARCHITECTURE,
MODEL.ROI_HEADS.NAME, MODEL.PROPOSAL_GENERATOR.NAME, MODEL.ROI_CUBE_HEAD.NAME."""
from __future__ import annotations


class Registry:
    def __init__(self, name: str):
        self._name = name
        self._obj_map = {}

    def register(self, obj=None):
        if obj is None:
            def deco(fn_or_cls):
                self._do_register(fn_or_cls.__name__, fn_or_cls)
                return fn_or_cls
            return deco
        self._do_register(obj.__name__, obj)
        return obj

    def _do_register(self, name, obj):
        assert name not in self._obj_map, f"An object named '{name}' was already registered in '{self._name}' registry!"
        self._obj_map[name] = obj

    def get(self, name):
        ret = self._obj_map.get(name)
        if ret is None:
            raise KeyError(f"No object named '{name}' found in '{self._name}' registry!")
        return ret

    def __contains__(self, name):
        return name in self._obj_map


BACKBONE_REGISTRY = Registry("BACKBONE")
META_ARCH_REGISTRY = Registry("META_ARCH")
ROI_HEADS_REGISTRY = Registry("ROI_HEADS")
PROPOSAL_GENERATOR_REGISTRY = Registry("PROPOSAL_GENERATOR")
ROI_CUBE_HEAD_REGISTRY = Registry("ROI_CUBE_HEAD")
