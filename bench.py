#!/usr/bin/env python3
"""Benchmark of the MI355X-native OVMono3D-LIFT inference path.

  python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

A "step" = one pass of the hot path (preprocess -> DINOv2 ViT + SFP -> ROIAlign -> cube head -> decode
-> postprocess -> per-image detection counts back on the host) over one batch of synthetic images that
are already resident in HBM. Workload = BASELINE.json configs[1]: DINOv2-L/14 + SFP, batch 1, a
512x512x3 synthetic image at network resolution 532x532 (ResizeShortestEdge(532) is host data feeding,
outside the path) on the reference's 896x896 SQUARE_PAD canvas (T = 4097 tokens). The 2D boxes come from
ROIHeads3DGDINO's native GroundingDINO network (Swin-B + BERT-base + 6/6 deformable transformer, 900 queries,
random-init weights, 6 prompted categories) -> phrase scores -> NMS; `--proposals oracle2d` measures the
oracle-2D branch instead (32 given boxes/image, SURVEY.md §8d mode A).

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel, HIP
events on the launch stream) and `cpu_baseline` (the fp32 torch CPU oracle on the host cores, rank 0,
N=1 only, bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

PEAK_F16_TFLOPS = 2516.6        # 256 CU x 2.4 GHz x 4096 FLOP/clk/CU, dense (MI355X_MICROARCH.md)


def vit_flops(T, D, L, G2):
    per_layer = 24.0 * T * D * D + 4.0 * T * T * D
    return L * per_layer + 2.0 * G2 * 588 * D


def kernel_flops(T, D):
    """Algorithmic FLOPs per launch (batch 1) of each profiled kernel category."""
    return {"attn": 4.0 * T * T * D, "qkv": 6.0 * T * D * D, "proj": 2.0 * T * D * D,
            "fc1": 8.0 * T * D * D, "fc2": 8.0 * T * D * D}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--precision", default="f16x3", choices=["f16x3", "f16"])
    ap.add_argument("--model", default="vitl14", help="DINOv2 arch (vitl14 ...) or, with --tower clip, an open_clip arch (ViT-B-16)")
    ap.add_argument("--tower", default="dinov2", choices=["dinov2", "clip", "mae", "midas", "sam"],
                    help="the other ViT towers of the reference's configs (parity-test cases, not the headline): clip = BASELINE configs[3], "
                         "use --model ViT-B-16 --canvas 1024 --net-res 608 --proposals oracle2d; mae: --model facebook/vit-mae-base; "
                         "midas: --model DPT_Large; sam: --model vit_b")
    ap.add_argument("--canvas", type=int, default=896)
    ap.add_argument("--net-res", type=int, default=532)
    ap.add_argument("--boxes", type=int, default=32, help="boxes per image for --proposals oracle2d")
    ap.add_argument("--proposals", default="gdino", choices=["gdino", "oracle2d", "rpn"],
                    help="gdino: ROIHeads3DGDINO (the headline); oracle2d: given boxes; rpn: the route the reference's --eval-only takes without "
                         "oracle boxes (RPN -> box head -> Fast R-CNN inference -> cube head, rcnn3d.py:105-111; its published number, "
                         "nohup.out:939, is ViT-B: --model vitb14)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt", action="store_true", help="skip the extra measurements (one-pass fp16; tight 518 canvas)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start one rank per GPU through torch.distributed.run as a CHILD process,
        # before anything in this process has touched the GPU (the reference's launch(), tools/train_net.py:563-570, does the same
        # with mp.spawn); rank 0's JSON line comes through the inherited stdout, the child's exit code becomes ours. Never re-exec.
        import socket
        import subprocess
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    import torch.distributed as dist
    # OVM_BENCH_BACKEND=gloo rehearses the N>1 code path on a box with fewer GPUs than ranks (ranks then share devices)
    backend = os.environ.get("OVM_BENCH_BACKEND", "nccl")
    dev_index = local_rank % max(torch.cuda.device_count(), 1)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
    if os.environ.get("OVM_BENCH_DRYRUN") == "1":
        # Rehearsal of everything AROUND the hot path on a box without a GPU (tests/test_distributed_cpu.py): launcher, rendezvous,
        # barrier, max-over-ranks and rank 0's single line. No step runs, and the line says so: value is null, never a measurement.
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        if world > 1:
            dist.barrier()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if rank == 0:
            print(json.dumps({"metric": "images/sec/GPU @512x512 DINOv2-L SFP; AP3D delta vs reference", "value": None, "unit": "images/sec",
                              "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "dry_run": True,
                              "note": "OVM_BENCH_DRYRUN=1: launcher / rendezvous rehearsal, no step was run"}))
        if world > 1:
            dist.destroy_process_group()
        return
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        # N ranks build their synthetic checkpoints on the host at the same time: share the cores instead of oversubscribing them N-fold
        try:
            avail_cores = len(os.sched_getaffinity(0))
        except AttributeError:
            avail_cores = os.cpu_count() or 1
        torch.set_num_threads(max(1, avail_cores // world))

    from common import build_cfg, build_clip_cfg, build_mae_cfg, build_midas_cfg, build_sam_cfg, oracle_params
    # experiment knobs of libovm3d (ovm_tune_set), e.g. OVM_TUNE=gdino_branches=0 - never set by the driver's runs
    from ovmono3d_amd import lib as _ovm_lib
    for kv in os.environ.get("OVM_TUNE", "").split(","):
        if "=" in kv:
            k, v = kv.split("=")
            if _ovm_lib.load().ovm_tune_set(k.encode(), int(v)) != 0:
                raise SystemExit(f"OVM_TUNE: unknown key {k!r}")
    from ovmono3d_amd.modeling import build_model
    from ovmono3d_amd.util.synth_weights import CLIP_ARCH, MAE_ARCH, MIDAS_ARCH, SAM_ARCH, VIT_ARCH, synth_state_dict

    clip = args.tower != "dinov2"                        # any of the patch-16 towers behind the 4-level pyramid
    arch_table = {"dinov2": VIT_ARCH, "clip": CLIP_ARCH, "mae": MAE_ARCH, "midas": MIDAS_ARCH, "sam": SAM_ARCH}[args.tower]
    D, L, heads = arch_table[args.model][:3]
    if args.tower == "mae":
        L -= 1                                           # the reference taps the state before the last block (backbone/mae.py:43-55)
    cfg_builder = {"dinov2": build_cfg, "clip": build_clip_cfg, "mae": build_mae_cfg, "midas": build_midas_cfg, "sam": build_sam_cfg}[args.tower]
    G = args.canvas // (16 if clip else 14)
    T = G * G + (0 if args.tower == "sam" else 1)
    B = args.batch

    def make_inputs(seed):
        g = torch.Generator().manual_seed(seed)
        out = []
        for i in range(B):
            img = torch.randint(0, 256, (3, args.net_res, args.net_res), dtype=torch.uint8, generator=g)
            oh = ow = 512
            K = [[1024.0, 0.0, 256.0], [0.0, 1024.0, 256.0], [0.0, 0.0, 1.0]]      # demo.py:63-76 focal 4.0*h/2
            x1 = torch.rand(args.boxes, generator=g) * 384
            y1 = torch.rand(args.boxes, generator=g) * 384
            w = 32 + torch.rand(args.boxes, generator=g) * 96
            h = 32 + torch.rand(args.boxes, generator=g) * 96
            boxes = torch.stack([x1, y1, x1 + w, y1 + h], 1)
            d = {"image": img, "height": oh, "width": ow, "K": K, "image_id": i}
            if use_gdino:
                d["category_list"] = list(CATEGORIES)
            elif not use_rpn:
                d["oracle2D"] = {"gt_bbox2D": boxes, "gt_classes": torch.randint(0, 50, (args.boxes,), generator=g),
                                 "gt_scores": 0.3 + 0.7 * torch.rand(args.boxes, generator=g)}
            out.append(d)
        return out

    sd = synth_state_dict(args.model, seed=0)
    use_gdino = args.proposals == "gdino"
    use_rpn = args.proposals == "rpn"
    CATEGORIES = ("chair", "dining table", "sofa", "potted plant", "television", "bookcase")
    gd_hf = gd_sd = None
    gd_events = []
    if use_gdino:
        if B != 1:
            raise SystemExit("ROIHeads3DGDINO processes one image per batch (reference rcnn3d.py:108): use --batch 1")
        from ovmono3d_amd.gdino.detector import HashTokenizer, NativeGroundingDino
        from synth_gdino import synth_gdino_model
        gd_hf, gd_sd = synth_gdino_model(0)

    def run(precision, steps, warmup, profile):
        cfg = cfg_builder(args.model, args.canvas, precision, max_batch=B, max_rois=1000 if (use_gdino or use_rpn) else max(64, args.boxes),
                        roi_heads="ROIHeads3DGDINO" if use_gdino else "ROIHeads3D",
                        extra=["MODEL.AMD.GDINO_CORUN", os.environ.get("OVM_BENCH_CORUN", "0") == "1"])
        model = build_model(cfg, device=dev)
        model.load_state_dict(sd)
        if use_gdino:
            # the detector object (C++ engine inside): RCNN3D.inference then takes the one-call path (ovm_infer)
            model.roi_heads.detector = NativeGroundingDino(dev, gd_sd, HashTokenizer(), cfg.MODEL.PIXEL_MEAN, cfg.MODEL.PIXEL_STD,
                                                           precision=3 if precision == "f16x3" else 1)
        host_inputs = make_inputs(1000 + rank)
        inputs = []
        for d in host_inputs:                      # images resident in HBM before the timed region
            d = dict(d)
            d["image"] = d["image"].to(dev)
            inputs.append(d)
        ndet = 0
        c_time = [0.0]
        if os.environ.get("OVM_BENCH_HOSTTIME") == "1" and use_gdino:      # diagnostic: seconds inside the one C call vs the whole step
            _inner = model.engine.infer_gdino
            def _timed(*a, **k):
                t = time.perf_counter(); r = _inner(*a, **k); c_time[0] += time.perf_counter() - t; return r
            model.engine.infer_gdino = _timed
        for _ in range(warmup):
            out = model(inputs)
        torch.cuda.synchronize()
        if profile:
            model.engine.profile_enable(True)      # HIP-event brackets on the launch stream; same-session A/B: they cost nothing (21.4-21.7 ms with all, attention only or none)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        c_time[0] = 0.0
        t0 = time.perf_counter()
        for _ in range(steps):
            out = model(inputs)
            ndet += sum(len(o["instances"]) for o in out)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        if os.environ.get("OVM_BENCH_HOSTTIME") == "1" and use_gdino and profile:
            print(f"[hosttime] step {dt / steps * 1e3:.3f} ms, inside ovm_infer {c_time[0] / steps * 1e3:.3f} ms", file=sys.stderr)
        prof = model.engine.profile_read() if profile else None
        if profile:
            model.engine.profile_enable(False)
        if use_gdino and profile:
            # the GroundingDINO network alone (nothing else on the chip), after the timed region: device time per forward
            det = model.roi_heads.detector
            from ovmono3d_amd.modeling.roi_heads.gdino_glue import build_caption
            cap = build_caption(list(CATEGORIES))[0]
            for _ in range(3):
                det(inputs[0]["image"], cap)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                det(inputs[0]["image"], cap)
            e1.record()
            torch.cuda.synchronize()
            gd_events.clear()
            gd_events.append((e0.elapsed_time(e1) / 10, det.engine.launches() if det.engine is not None else None))
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, prof, ndet, model, host_inputs, cfg, out, inputs

    dt, prof, ndet, model, host_inputs, cfg, last_out, inputs_dev = run(args.precision, args.steps, args.warmup, True)
    ms_per_step = dt / args.steps * 1e3
    value = world * B * args.steps / dt

    # ---- roofline of the dominant kernel (HIP events, launch stream) ----
    kf = kernel_flops(T, D)
    dominant = max((k for k in kf), key=lambda k: prof[k][0])
    tot_ms, launches = prof[dominant]
    avg_ms = tot_ms / max(launches, 1)
    flops_launch = kf[dominant] * B
    achieved = flops_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
    # HBM-side bytes per launch of the dominant kernel come from a separate rocprofv3 --pmc pass over this same command
    # (profiles/rNN/pmc_traffic_*.json documents the command and the gfx950 FETCH_SIZE correction; newest round first); null if none matches
    traffic = traffic_source = None
    for tdir in ("r03", "r02", "r01"):
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", tdir, "pmc_traffic_f16x3_vitl14_T4097_b1.json")))
            wl = tj["workload"]
            # per-launch HBM bytes of a kernel do not depend on what runs beside it, but the figure is only quoted for the same
            # model / canvas / precision / batch, and its provenance is stated next to it
            if (wl["model"], wl["canvas"], wl["precision"], wl["batch"]) == (args.model, args.canvas, args.precision, B):
                key = {"attn": "attn_kernel"}.get(dominant)
                for kn, kv in tj["kernels"].items():
                    if key and key in kn:
                        traffic = kv["traffic_bytes_per_launch"]
                        traffic_source = (f"profiles/{tdir}/pmc_traffic_f16x3_vitl14_T4097_b1.json: separate rocprofv3 --pmc FETCH_SIZE / "
                                          f"WRITE_SIZE passes over bench.py --proposals {wl.get('proposals', 'oracle2d')} (per-launch bytes of the "
                                          "same kernel at the same shapes, counters-only run of its own: PMC collection cannot share the timed run)")
                if traffic is not None:
                    break
        except (OSError, KeyError, ValueError):
            continue
    roofline = {"bound": "mfma", "kernel": dominant, "achieved": round(achieved, 2), "peak": PEAK_F16_TFLOPS,
                "unit": "TFLOP/s", "frac": round(achieved / PEAK_F16_TFLOPS, 4), "traffic": traffic,
                "traffic_source": traffic_source,
                "avg_launch_ms": round(avg_ms, 4), "launches": launches,
                "algorithmic_flops_per_launch": flops_launch,
                "mfma_passes_per_product": 3 if args.precision == "f16x3" else 1}
    if args.tower == "sam":
        roofline["note"] = ("the per-launch FLOP model assumes global attention over all T tokens; 8 of SAM's 12 blocks attend inside 14 x 14 "
                            "windows and all of them run on the fp32-MFMA attention kernel: read `value`, not this fraction")
    if use_gdino:
        roofline["note"] = ("launch durations are measured while the GroundingDINO graph runs concurrently on a side stream "
                            "(the kernels share the CUs); --proposals oracle2d gives the uncontended figure")
    kernels = {k: {"ms_per_step": round(prof[k][0] / args.steps, 4), "launches_per_step": prof[k][1] // args.steps,
                   **({"tflops": round(kf[k] * B * prof[k][1] / (prof[k][0] * 1e-3) / 1e12, 2)} if k in kf and prof[k][0] > 0 else {})}
               for k in prof}
    if use_gdino and gd_events:
        kernels["gdino_network_alone"] = {"ms_per_forward": round(gd_events[0][0], 4), "launches_per_forward": gd_events[0][1],
                                          "note": "the C++ GroundingDINO engine by itself (graph replay), measured after the timed region; "
                                                  "inside a step it runs on a side stream beside the ViT"}
    total_flops = vit_flops(T, D, L, G * G) * B
    e2e_tflops = total_flops * args.steps / dt / 1e12

    alt = None
    if not args.no_alt and args.precision == "f16x3" and world == 1:
        # the one-pass fp16 mode is measured in a child process of its own (fresh handle, graphs and allocator state), after
        # this process has finished its timed region
        import subprocess
        torch.cuda.synchronize()
        cmd = [sys.executable, os.path.abspath(__file__), "--precision", "f16", "--no-alt", "--no-cpu-baseline", "--steps", str(args.steps),
               "--warmup", str(args.warmup), "--proposals", args.proposals, "--model", args.model, "--canvas", str(args.canvas),
               "--net-res", str(args.net_res), "--boxes", str(args.boxes), "--batch", str(args.batch)]
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
            d2 = json.loads(r.stdout.strip().splitlines()[-1])
            alt = {"precision": "f16 (one MFMA pass; outside the 1e-3 parity band on the synthetic checkpoint)",
                   "value": d2["value"], "ms_per_step": d2["ms_per_step"], "dominant_kernel": d2["roofline"]["kernel"],
                   "dominant_tflops": d2["roofline"]["achieved"]}
        except Exception as e:                                   # the extra measurement must never break the contract line
            alt = {"precision": "f16", "error": repr(e)[:200]}
    alt_canvas = None
    if not args.no_alt and world == 1 and args.canvas == 896 and args.net_res == 532:
        # SURVEY.md 8(d) second canvas mode, a declared deviation from the reference's plumbing: 512x512 fed unresized on a
        # tight 518x518 canvas (G = 37, T = 1370) instead of ResizeShortestEdge(532) on SQUARE_PAD 896
        import subprocess
        cmd = [sys.executable, os.path.abspath(__file__), "--precision", args.precision, "--no-alt", "--no-cpu-baseline", "--steps",
               str(args.steps), "--warmup", str(args.warmup), "--proposals", args.proposals, "--model", args.model, "--canvas", "518",
               "--net-res", "512", "--boxes", str(args.boxes), "--batch", str(args.batch)]
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
            d3 = json.loads(r.stdout.strip().splitlines()[-1])
            alt_canvas = {"mode": "tight: 512x512 unresized on a 518 canvas (T=1370); NOT the reference's plumbing", "value": d3["value"],
                          "ms_per_step": d3["ms_per_step"]}
        except Exception as e:
            alt_canvas = {"mode": "tight", "error": repr(e)[:200]}

    # ---- a dataset does not come in one shape: 90 synthetic images of 45 distinct network resolutions through the same model
    # (what ResizeShortestEdge(532, 896) emits for aspect ratios 0.59 .. 1.68), one image per step as the evaluation loop feeds
    # them. The detector builds one plan + HIP graph per padded shape on first sight (plan cache: LRU bounded by bytes), so pass 1
    # pays the captures and the later passes are the steady state. The same images GROUPED by shape (every plan switch removed) are
    # timed beside the mixed order: their ratio is what switching plans costs. Larger images carry more detector work than the
    # 532 x 532 headline, so neither figure is comparable with `value`; reported next to it, never as `value`.
    mixed = None
    if not args.no_alt and world == 1 and args.canvas == 896 and args.net_res == 532:
        try:
            shapes = [(532, w) for w in range(532, 897, 16)] + [(h, 532) for h in range(548, 897, 16)]
            g = torch.Generator().manual_seed(77)
            order = torch.randperm(2 * len(shapes), generator=g).tolist()
            mixed_inputs = []
            for i in order:
                hh, ww = shapes[i % len(shapes)]
                d = {"image": torch.randint(0, 256, (3, hh, ww), dtype=torch.uint8, generator=g).to(dev), "height": hh, "width": ww,
                     "K": [[2.0 * hh, 0.0, ww / 2], [0.0, 2.0 * hh, hh / 2], [0.0, 0.0, 1.0]], "image_id": i}
                if use_gdino:
                    d["category_list"] = list(CATEGORIES)
                elif not use_rpn:
                    d["oracle2D"] = host_inputs[0]["oracle2D"]
                mixed_inputs.append(d)
            grouped = sorted(mixed_inputs, key=lambda d: tuple(d["image"].shape[1:]))

            def one_pass(seq):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for d in seq:
                    model([d])
                torch.cuda.synchronize()
                return time.perf_counter() - t0
            n = len(mixed_inputs)
            first = one_pass(mixed_inputs)
            t_mixed = min(one_pass(mixed_inputs) for _ in range(2))
            t_grouped = min(one_pass(grouped) for _ in range(2))
            mixed = {"images": n, "distinct_shapes": len({tuple(d["image"].shape[1:]) for d in mixed_inputs}),
                     "images_per_sec_first_pass": round(n / first, 2), "images_per_sec": round(n / t_mixed, 2),
                     "images_per_sec_grouped_by_shape": round(n / t_grouped, 2), "mixed_over_grouped": round(t_grouped / t_mixed, 4),
                     "note": "network resolutions 532 x 532 .. 532 x 896 / 896 x 532 on the 896 canvas, one image per step; the first pass "
                             "includes one plan + graph capture per new shape; mixed_over_grouped = the steady-state rate in shuffled "
                             "dataset order relative to the same images grouped by shape (1.0 = switching plans is free)"}
            if use_gdino:
                eng = model.roi_heads.detector.engine
                mixed["plans_held"] = int(eng.debug_scalar("plans"))
                mixed["plan_cache_mb"] = round(eng.debug_scalar("plan_bytes") / 2 ** 20, 1)
        except Exception as e:                                   # the extra measurement must never break the contract line
            mixed = {"error": repr(e)[:300]}

    cpu_baseline = parity = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle.pipeline import inference
        from parity import parity_ok, parity_report
        try:
            avail = len(os.sched_getaffinity(0))
        except AttributeError:
            avail = os.cpu_count() or 1
        # the GPU box gives a 1-GPU job a 16-core share; more torch threads than that only oversubscribe
        ncores = int(os.environ.get("OVM_CPU_THREADS", min(avail, 16)))
        torch.set_num_threads(ncores)
        P = oracle_params(cfg)
        cpu_idx = [0] if B == 1 else [0, B - 1]              # the images the CPU leg computes: first and last of the batch
        cpu_in = {i: dict(host_inputs[i]) for i in cpu_idx}
        if use_gdino:
            # CPU leg of the GroundingDINO branch: the Hugging Face port (fp32, eager; patched in the three places where
            # transformers 5.x departs from upstream, tests/hf_gdino_patches.py) as the stand-in for the upstream network the
            # reference calls, + the oracle's restatement of the reference glue (oracle/gdino_glue.py)
            from hf_gdino_patches import patch_hf_to_upstream, upstream_position_ids
            from oracle import gdino_glue as og
            patch_hf_to_upstream(gd_hf)
            tok = HashTokenizer()
            caption, cap_list = og.build_caption(list(CATEGORIES))
            ids = tok.encode(caption)
            spans = og.phrase_spans(ids, [tok.encode(c, add_special_tokens=False) for c in cap_list])
            mean = torch.tensor(cfg.MODEL.PIXEL_MEAN).view(3, 1, 1)
            std = torch.tensor(cfg.MODEL.PIXEL_STD).view(3, 1, 1)
            cpu_in[0].pop("category_list")

            def cpu_gdino():
                x = ((cpu_in[0]["image"].float() - mean) / std)[[2, 1, 0]]
                with upstream_position_ids():
                    o = gd_hf(pixel_values=x[None], input_ids=torch.tensor(ids)[None], return_dict=True)
                lg = torch.full((o.logits.shape[1], 256), float("-inf"))
                lg[:, :len(ids)] = o.logits[0][:, :len(ids)]
                return og.gdino_postprocess(lg, o.pred_boxes[0], spans, cap_list, [[c] for c in CATEGORIES], x.shape[1:])
        refs, boxes2d = {}, {}
        with torch.no_grad():
            t0 = time.perf_counter()
            nimg = 0
            todo = list(cpu_idx)
            while True:
                i = todo.pop(0) if todo else 0
                given = None
                if use_gdino:
                    bx, sc, cl = cpu_gdino()
                    given = [dict(pred_boxes=bx, pred_classes=cl, scores=sc)]
                ref, aux = inference(sd, [cpu_in[i]], P, given_boxes=given, return_aux=True)
                if i not in refs:
                    refs[i] = ref[0]
                    boxes2d[i] = aux["instances_2d"][0]      # the 2D detections (network resolution) the oracle handed to its cube head
                nimg += 1
                el = time.perf_counter() - t0
                if not todo and (el > 12.0 or nimg >= 3):
                    break
        cpu_baseline = {"value": round(nimg / el, 4), "unit": "images/sec", "cores": ncores, "kind": "port",
                        "sample": f"{nimg} image(s) of the same workload through oracle/ (fp32 torch CPU restatement"
                                  + (" + Hugging Face GroundingDINO fp32 on CPU for the text-prompted boxes" if use_gdino else "")
                                  + f"), {el:.1f} s, torch threads={ncores}"}
        # parity of the timed configuration itself: the HIP outputs of the timed batch against the CPU leg's outputs.
        # (a) end to end, images 0 and B-1 against the oracle. Behind a proposal stage (900-query detector, or RPN top-1000 / NMS /
        #     box head / per-class NMS) the two routes take hundreds of discrete decisions on scores that agree to ~1e-6, so a box on
        #     an edge may flip and the 2D boxes they hand to the cube head differ in the last bits; detections are paired by box.
        # (b) same boxes: the CPU leg's 2D boxes through the HIP cube branch on the HIP features - no discrete decision in
        #     between, identity pairing, the strict 1e-3 check of the float path.
        # (c) batch > 1: EVERY image of the batch against its own batch-1 run on the HIP path (ids exact, floats <= 1e-5: a batch
        #     changes which rows take the GEMMs' leftover-row path, i.e. fp32 summation order, never the arithmetic; measured 2-4e-6
        #     on pred_pose, less elsewhere); with (a) on the first and last image this covers all B.
        proposal_stage = use_gdino or use_rpn
        parity = {"end_to_end": {}, "images_vs_oracle": cpu_idx}
        ok_all = True
        for i in cpu_idx:
            e2e = parity_report(last_out[i]["instances"], refs[i])
            parity["end_to_end"][f"image{i}"] = e2e
            n_ref = max(e2e["n_det_oracle"], 1)
            if proposal_stage:
                from ovmono3d_amd.structures import Boxes, Instances
                b2 = boxes2d[i]
                with torch.no_grad():
                    images = model.preprocess_image([inputs_dev[i]])
                    model.backbone(images)
                    t_in = Instances(images.image_sizes[0])
                    t_in.pred_boxes, t_in.scores, t_in.pred_classes = Boxes(b2["pred_boxes"].to(dev)), b2["scores"].to(dev), b2["pred_classes"].to(dev)
                    got_b = model.roi_heads._forward_cube(None, [t_in], None, list(images.image_sizes),
                                                          [host_inputs[i]["height"] / images.image_sizes[0][0]], images=images, postprocess=True)[0]
                strict = parity_report(got_b, refs[i], box_tol=1e-4)
                parity.setdefault("same_boxes", {})[f"image{i}"] = strict
                flips_ok = e2e["unmatched_oracle"] <= max(2, n_ref // 100) and e2e["unmatched_hip"] <= max(2, n_ref // 100)
                e2e_pairs = dict(e2e, unmatched_oracle=0, unmatched_hip=0)
                ok_all &= bool(flips_ok and parity_ok(e2e_pairs, 1e-3, pose_by_conditioning=True) and parity_ok(strict, 1e-3))
            else:
                ok_all &= bool(parity_ok(e2e, 1e-3))
        if B > 1:
            from parity import FIELDS as _PF
            worst, worst_pose, ids_ok, counts_ok = 0.0, 0.0, True, True
            with torch.no_grad():
                for i in range(B):
                    solo = model([inputs_dev[i]])[0]["instances"]
                    inst = last_out[i]["instances"]
                    if len(solo) != len(inst):
                        counts_ok = False
                        continue
                    if len(inst) == 0:
                        continue
                    ids_ok &= bool(torch.equal(solo.pred_classes, inst.pred_classes))
                    for f in _PF:
                        a_, b_ = inst.get(f), solo.get(f)
                        a_, b_ = (a_.tensor if hasattr(a_, "tensor") else a_).double(), (b_.tensor if hasattr(b_, "tensor") else b_).double()
                        e_ = float((a_ - b_).abs().max() / b_.abs().max().clamp_min(1e-30))
                        if f == "pred_pose":
                            worst_pose = max(worst_pose, e_)
                        else:
                            worst = max(worst, e_)
            parity["batch_vs_batch1"] = {"images": B, "same_counts": counts_ok, "class_ids_exact": ids_ok, "worst_rel_err": worst,
                                         "worst_rel_err_pose": worst_pose}
            ok_all &= bool(counts_ok and ids_ok and worst <= 1e-5 and worst_pose <= 1e-4)
        parity["ok_1e-3"] = bool(ok_all)
        parity["criteria"] = ("max_rel_err = max|a-b| / max|b| per field; max_elem_rel_err = max_i |a_i-b_i| / max(|b_i|, floor) with the floors of "
                              "tests/parity.py:ELEM_FLOOR. Gate: class ids exact, every float field <= 1e-3 (max_rel_err)"
                              + ("; behind the proposal stage: same_boxes strict (identity pairing), end_to_end with <= 1 % of the detections "
                                 "flipped by discrete near-ties, pred_pose <= 1e-3 on every detection whose 6-D -> R map is well conditioned "
                                 "(amplification <= 5) and, for near-degenerate Gram-Schmidt inputs, held to the angle a 1e-3-relative "
                                 "perturbation of the raw 6-D output causes at that conditioning (pose.* fields; tests/parity.py)" if proposal_stage else "")
                              + ("; batch_vs_batch1: every image of the batch equals its own batch-1 run within 1e-5 (pred_pose, which carries the 6-D head's conditioning, 1e-4), ids exact" if B > 1 else ""))
        parity["oracle"] = ("oracle/ restatement" + (" + Hugging Face GroundingDINO port" if use_gdino else "")
                            + "; unpinned vs the reference itself (no reference fixtures exist, DESIGN.md 5)")

    if rank == 0:
        line = {
            "metric": "images/sec/GPU @512x512 DINOv2-L SFP; AP3D delta vs reference",
            "value": round(value, 3), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16x3" if args.precision == "f16x3" else "f16", "data": "synthetic",
            "config": {"workload": {"dinov2": "DINOv2 ", "clip": "CLIP image tower ", "mae": "MAE encoder ", "midas": "MiDaS DPT ViT ",
                                    "sam": "SAM image encoder "}[args.tower] + f"{args.model} + SFP + "
                                   + ("ROIHeads3DGDINO (native GroundingDINO Swin-B/BERT-base, 900 queries, 6 categories -> NMS)" if use_gdino
                                      else ("RPN (top-1000 / level, NMS 0.7) -> box head -> Fast R-CNN inference (<= 100 detections)" if use_rpn
                                            else f"oracle-2D boxes ({args.boxes}/img)"))
                                   + f" + ROIAlign + CubeHead + decode, batch {B}/GPU, 512x512 synthetic -> network res {args.net_res} "
                                     f"-> canvas {args.canvas} (T={T}), random-init weights (seed 0)",
                       "proposal_source": "ROIHeads3DGDINO native GroundingDINO" if use_gdino else ("RPN + box head" if use_rpn else "oracle2D"),
                       "precision": args.precision, "parallelism": f"dp{world} (image-sharded, no data-path collective)",
                       "ap3d_delta": "not measurable offline (no Omni3D data / checkpoint); proxy = tensor parity of THIS configuration vs "
                                     "the CPU oracle, see `parity` (" + ("not run" if parity is None else
                                                                         ("within 1e-3, class ids exact" if parity["ok_1e-3"] else "FAILED")) + ")"},
            "images_per_sec_per_gpu": round(value / world, 3),
            "vit_tflops_end_to_end": round(e2e_tflops, 2),
            "detections_per_step": ndet // max(args.steps, 1),
            "roofline": roofline, "kernels": kernels, "cpu_baseline": cpu_baseline, "parity": parity, "mixed_shapes": mixed,
        }
        if alt:
            line["alt_precision"] = alt
        if alt_canvas:
            line["alt_canvas"] = alt_canvas
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
