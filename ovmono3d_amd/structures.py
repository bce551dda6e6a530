"""Minimal ``Boxes`` / ``Instances`` / ``ImageList`` with the detectron2 semantics the path relies on.

Consumers in the reference: demo/demo.py:93-96 (iterates ``dets.pred_bbox3D`` ...),
cubercnn/evaluation/omni3d_evaluation.py:1219-1232 (``instances.has('pred_bbox3D')``, ``.tensor``).
These are host-side containers around torch tensors (device memory handles); no arithmetic
of the hot path lives here.
"""
from __future__ import annotations

from typing import Any, Dict, List, Tuple

import torch


class Boxes:
    """Nx4 XYXY boxes."""

    def __init__(self, tensor: torch.Tensor):
        if not isinstance(tensor, torch.Tensor):
            tensor = torch.as_tensor(tensor, dtype=torch.float32)
        tensor = tensor.to(torch.float32)
        if tensor.numel() == 0:
            tensor = tensor.reshape((-1, 4))
        assert tensor.dim() == 2 and tensor.size(-1) == 4, tensor.size()
        self.tensor = tensor

    def to(self, *a, **k) -> "Boxes":
        return Boxes(self.tensor.to(*a, **k))

    def clone(self) -> "Boxes":
        return Boxes(self.tensor.clone())

    @property
    def device(self):
        return self.tensor.device

    def __len__(self) -> int:
        return self.tensor.shape[0]

    def __getitem__(self, item) -> "Boxes":
        if isinstance(item, int):
            return Boxes(self.tensor[item].view(1, -1))
        return Boxes(self.tensor[item])

    def __iter__(self):
        yield from self.tensor

    def __repr__(self):
        return "Boxes(" + str(self.tensor) + ")"


class Instances:
    """Per-image container of equally long fields (detectron2.structures.Instances semantics)."""

    def __init__(self, image_size: Tuple[int, int], **kwargs: Any):
        self._image_size = tuple(int(x) for x in image_size)
        self._fields: Dict[str, Any] = {}
        for k, v in kwargs.items():
            self.set(k, v)

    @property
    def image_size(self) -> Tuple[int, int]:
        return self._image_size

    def __setattr__(self, name: str, val: Any) -> None:
        if name.startswith("_"):
            super().__setattr__(name, val)
        else:
            self.set(name, val)

    def __getattr__(self, name: str) -> Any:
        if name == "_fields" or name not in self._fields:
            raise AttributeError("Cannot find field '{}' in the given Instances!".format(name))
        return self._fields[name]

    def set(self, name: str, value: Any) -> None:
        data_len = len(value)
        if len(self._fields):
            assert len(self) == data_len, \
                "Adding a field of length {} to a Instances of length {}".format(data_len, len(self))
        self._fields[name] = value

    def has(self, name: str) -> bool:
        return name in self._fields

    def remove(self, name: str) -> None:
        del self._fields[name]

    def get(self, name: str) -> Any:
        return self._fields[name]

    def get_fields(self) -> Dict[str, Any]:
        return self._fields

    def to(self, *args: Any, **kwargs: Any) -> "Instances":
        ret = Instances(self._image_size)
        for k, v in self._fields.items():
            if hasattr(v, "to"):
                v = v.to(*args, **kwargs)
            ret.set(k, v)
        return ret

    def __getitem__(self, item) -> "Instances":
        if isinstance(item, int):
            if item >= len(self) or item < -len(self):
                raise IndexError("Instances index out of range!")
            item = slice(item, None, len(self))
        ret = Instances(self._image_size)
        for k, v in self._fields.items():
            ret.set(k, v[item])
        return ret

    def __len__(self) -> int:
        for v in self._fields.values():
            return v.__len__()
        raise NotImplementedError("Empty Instances does not support __len__!")

    def __iter__(self):
        raise NotImplementedError("`Instances` object is not iterable!")

    def __repr__(self) -> str:
        s = self.__class__.__name__ + "("
        s += "num_instances={}, ".format(len(self) if self._fields else 0)
        s += "image_height={}, image_width={}, ".format(*self._image_size)
        s += "fields=[{}])".format(", ".join(f"{k}: {v}" for k, v in self._fields.items()))
        return s


class ImageList:
    """Batched padded images + the unpadded (H, W) of each (detectron2.structures.ImageList).

    In this build ``tensor`` is the NHWC uint8 canvas-less batch handle used by the native path:
    the zero padding to the square canvas happens inside the patch-gather kernel, so ``tensor``
    may be ``None`` when the caller only needs ``image_sizes``.
    """

    def __init__(self, tensor, image_sizes: List[Tuple[int, int]]):
        self.tensor = tensor
        self.image_sizes = [tuple(int(v) for v in s) for s in image_sizes]

    def __len__(self) -> int:
        return len(self.image_sizes)
