"""Per-rank inference loop, record -> COCO-style JSON conversion and the single result gather.

Mirrors reference cubercnn/evaluation/omni3d_evaluation.py:626-734 (``inference_on_dataset``: warm-up reset
after 5 iterations, data / compute / eval timers, ``torch.cuda.synchronize()`` before stopping the compute
timer, depth stacking :661-665, gather to rank 0 :717-720) and :1200-1252 (``instances_to_coco_json``).
The AP computation (Omni3Deval with true 3D IoU) lives in ``omni3d_eval.py``.
"""
from __future__ import annotations

import datetime
import logging
import time
from typing import Dict, List

import torch

from .distributed import gather_records, get_rank, get_world_size

logger = logging.getLogger("cubercnn")


def instances_to_coco_json(instances, img_id) -> List[Dict]:
    """reference omni3d_evaluation.py:1200-1252 (XYXY->XYWH; one dict per detection)."""
    n = len(instances) if instances.get_fields() else 0
    if n == 0:
        return []
    if isinstance(img_id, str):
        img_id = hash(img_id.split("/")[-1]) % (10 ** 8)            # fallback of :1209-1217 when no id map is known
    boxes = instances.pred_boxes.tensor.cpu().clone()
    boxes[:, 2] -= boxes[:, 0]
    boxes[:, 3] -= boxes[:, 1]
    boxes = boxes.tolist()
    scores = instances.scores.cpu().tolist()
    classes = instances.pred_classes.cpu().tolist()
    has_3d = instances.has("pred_bbox3D")
    if has_3d:
        bbox3D = instances.pred_bbox3D.cpu().numpy()
        center_cam = instances.pred_center_cam.cpu().numpy()
        center_2D = instances.pred_center_2D.cpu().numpy()
        dims = instances.pred_dimensions.cpu().numpy()
        pose = instances.pred_pose.cpu().numpy()
    out = []
    for k in range(n):
        r = {"image_id": img_id, "category_id": classes[k], "bbox": boxes[k], "score": scores[k]}
        if has_3d:
            r["bbox3D"] = bbox3D[k].tolist()
            r["center_cam"] = center_cam[k].tolist()
            r["center_2D"] = center_2D[k].tolist()
            r["dimensions"] = dims[k].tolist()
            r["pose"] = pose[k].tolist()
            r["depth"] = float(center_cam[k][2])
        out.append(r)
    return out


class Omni3DEvaluator:
    def __init__(self, dataset_name: str = "", output_dir: str = "", filter_settings=None):
        self.dataset_name, self.output_dir = dataset_name, output_dir

    def instances_to_coco_json(self, instances, img_id):
        return instances_to_coco_json(instances, img_id)


def _records_of(instances, image_index: int, device: torch.device) -> torch.Tensor:
    """[n,48] float32 record tensor of one image's Instances (layout OvmDet3D) on ``device`` - the device the run gathers
    on, NOT the device of the instance fields: an image with no detection (empty oracle list, no GroundingDINO box, the
    n == 0 early return of ``_forward_cube``) carries CPU or no fields and must still concatenate with its neighbours."""
    n = len(instances) if instances.get_fields() else 0
    rec = torch.zeros((n, 48), dtype=torch.float32, device=device)
    if n == 0:
        return rec
    rec[:, 0:4] = instances.pred_boxes.tensor.to(device)
    rec[:, 4] = instances.scores.to(device)
    rec[:, 5] = instances.pred_classes.to(device, torch.int32).contiguous().view(torch.float32)
    if instances.has("pred_bbox3D"):
        rec[:, 6:30] = instances.pred_bbox3D.reshape(n, 24).to(device)
        rec[:, 30:33] = instances.pred_center_cam.to(device)
        rec[:, 33:35] = instances.pred_center_2D.to(device)
        rec[:, 35:38] = instances.pred_dimensions.to(device)
        rec[:, 38:47] = instances.pred_pose.reshape(n, 9).to(device)
    rec[:, 47] = torch.full((n,), image_index, dtype=torch.int32, device=device).view(torch.float32)
    return rec


def _json_of_records(rec: torch.Tensor, img_id) -> List[Dict]:
    rec = rec.cpu()
    cls = rec[:, 5].contiguous().view(torch.int32).tolist()
    out = []
    for k in range(rec.shape[0]):
        r = rec[k]
        b = r[0:4].tolist()
        out.append({"image_id": img_id, "category_id": cls[k], "bbox": [b[0], b[1], b[2] - b[0], b[3] - b[1]],
                    "score": float(r[4]), "bbox3D": r[6:30].view(8, 3).tolist(), "center_cam": r[30:33].tolist(),
                    "center_2D": r[33:35].tolist(), "dimensions": r[35:38].tolist(), "pose": r[38:47].view(3, 3).tolist(),
                    "depth": float(r[32])})
    return out


def inference_on_dataset(model, data_loader, evaluator=None) -> List[Dict]:
    """Returns the list of per-image prediction dicts on rank 0 (``[]`` elsewhere), in dataset order."""
    world = get_world_size()
    total = len(data_loader)
    num_warmup = min(5, max(total - 1, 0))
    start_time = time.perf_counter()
    total_data = total_compute = total_eval = 0.0
    metas: List[Dict] = []
    recs: List[torch.Tensor] = []
    n_img = 0
    # records live on the model's device (with world > 1 the nccl gather needs device tensors on every rank, also on a rank
    # whose images all came back empty)
    rec_dev = torch.device(getattr(model, "device", None) or "cpu")
    model.eval()
    with torch.no_grad():
        start_data = time.perf_counter()
        for idx, inputs in enumerate(data_loader):
            total_data += time.perf_counter() - start_data
            if idx == num_warmup:
                start_time = time.perf_counter()
                total_data = total_compute = total_eval = 0.0
            t0 = time.perf_counter()
            if "depth" in inputs[0]:
                depth = torch.stack([x["depth"] for x in inputs])                     # :661-665
                outputs = model(inputs, prompt_depth=depth)
            else:
                outputs = model(inputs)
            if torch.cuda.is_available():
                torch.cuda.synchronize()                                               # :669
            total_compute += time.perf_counter() - t0
            t0 = time.perf_counter()
            for inp, out in zip(inputs, outputs):
                inst = out["instances"]
                metas.append({"image_id": inp.get("image_id", inp.get("file_name", str(idx))), "K": inp["K"],
                              "width": inp["width"], "height": inp["height"], "n": len(inst) if inst.get_fields() else 0})
                recs.append(_records_of(inst, n_img, rec_dev))
                n_img += 1
            total_eval += time.perf_counter() - t0
            iters = idx + 1 - num_warmup * int(idx >= num_warmup)
            if idx >= num_warmup * 2 and (idx % 50 == 0):
                tot = (time.perf_counter() - start_time) / iters
                eta = datetime.timedelta(seconds=int(tot * (total - idx - 1)))
                logger.info(f"Inference done {idx + 1}/{total}. Dataloading: {total_data / iters:.4f} s/iter. "
                            f"Inference: {total_compute / iters:.4f} s/iter. Eval: {total_eval / iters:.4f} s/iter. "
                            f"Total: {tot:.4f} s/iter. ETA={eta}")
            start_data = time.perf_counter()
    total_time = time.perf_counter() - start_time
    denom = max(total - num_warmup, 1)
    logger.info(f"Total inference time: {datetime.timedelta(seconds=int(total_time))} ({total_time / denom:.6f} s / iter per device, "
                f"on {world} devices)")
    logger.info(f"Total inference pure compute time: {datetime.timedelta(seconds=int(total_compute))} "
                f"({total_compute / denom:.6f} s / iter per device, on {world} devices)")
    mine = torch.cat(recs) if recs else torch.zeros((0, 48), dtype=torch.float32, device=rec_dev)
    if world > 1:
        import torch.distributed as dist
        allrec, _ = gather_records(mine, dst=0)                                        # :717-720, one gather per dataset
        gathered = [None] * world if get_rank() == 0 else None
        dist.gather_object(metas, gathered, dst=0)
        if get_rank() != 0:
            return []
        metas = [m for part in gathered for m in part]
        mine = allrec
    results, ofs = [], 0
    mine = mine.cpu()
    for m in metas:
        n = m.pop("n")
        pred = dict(m)
        pred["instances"] = _json_of_records(mine[ofs: ofs + n], m["image_id"])
        ofs += n
        results.append(pred)
    return results
