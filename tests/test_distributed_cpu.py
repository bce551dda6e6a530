"""world_size-2 gloo rehearsal of the multi-GPU path: contiguous image shards per rank (the reference's
InferenceSampler, cubercnn/data/build.py:320) and one gather of detection records to rank 0 in rank order
(comm.gather + itertools.chain, omni3d_evaluation.py:717-720)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_items, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ovmono3d_amd import lib
    from ovmono3d_amd.evaluation.distributed import gather_records
    b, e = lib.shard_range(n_items, rank, world)
    # fake "detections": image i yields (i % 3) records whose first float encodes (image, k)
    recs = [torch.tensor([[i * 10.0 + k] + [float(rank)] * 47]) for i in range(b, e) for k in range(i % 3)]
    mine = torch.cat(recs) if recs else torch.zeros(0, 48)
    allrec, counts = gather_records(mine, dst=0)
    if rank == 0:
        q.put((allrec.numpy(), counts))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_inference_gathers_in_dataset_order():
    world, n_items = 2, 11
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_items, q)) for r in range(world)]
    for p in procs:
        p.start()
    rec, counts = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    expect = [i * 10.0 + k for i in range(n_items) for k in range(i % 3)]
    assert rec[:, 0].tolist() == expect                 # concatenation of contiguous shards == dataset order
    assert sum(counts) == len(expect) and len(counts) == world


def test_bench_gpus2_launches_its_own_ranks_dryrun():
    """`python bench.py --gpus 2` without a launcher must start its two ranks itself (a child torch.distributed.run, before anything
    touches a GPU - the reference's launch(), tools/train_net.py:563-570) and print exactly ONE JSON line with n_gpus 2 from rank 0.
    Here, without a GPU, the hot path cannot run (there is no CPU fallback), so OVM_BENCH_DRYRUN=1 rehearses launcher, gloo rendezvous,
    barriers and the max-over-ranks reduction only - the line says dry_run and carries no value. The same command with real steps on
    two ranks runs under -m gpu (tests/test_gpu_entrypoints.py::test_bench_gpus2_real_steps_gloo)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, OVM_BENCH_BACKEND="gloo", OVM_BENCH_DRYRUN="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["dry_run"] is True and d["value"] is None and d["steps"] == 2
