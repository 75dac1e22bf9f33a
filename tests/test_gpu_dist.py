"""The complete sharded path (ShardedMatcher: HIP kernels per shard -> all-gather of top-K -> merge kernel) with
two ranks.  The GPU box has one device, so both ranks use cuda:0 and the collective travels over gloo (host
staging); on a multi-GPU node the same code runs over RCCL (backend "nccl")."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, P, Q, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
        from helpers import pretrained_example
        from lens_amd.dist import ShardedMatcher
        from lens_amd.engine import LensEngine, seqmatch_topk
        from lens_amd.synthetic import synthetic_frames, synthetic_database
        w1, w2p = pretrained_example()
        w2 = synthetic_database(w2p, P)
        frames = synthetic_frames(Q, 128, seed=3).cuda()
        m = ShardedMatcher(torch.from_numpy(w1), w2, 128, 10, 250, L=2, K=25, rank=rank, world=world, device="cuda:0")
        idx, val = m.match(frames)
        full = LensEngine(w1, w2, 128, 10, 250, device="cuda:0")
        fi, fv = seqmatch_topk(full.forward_counts(frames), 2, 25)
        ok = bool(torch.equal(idx, fi) and torch.equal(val, fv))
        t = torch.tensor([int(ok)])
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        if rank == 0:
            ret.put(int(t.item()))
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_matcher_equals_single_device():
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 1500, 24, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert ret.get(timeout=5) == 1


@pytest.mark.parametrize("shard", ["queries", "db"])
def test_bench_two_ranks_the_way_the_driver_launches_it(shard):
    """bench.py under torch.distributed.run with 2 ranks.  The box has one GPU, so the test hook
    LENS_BENCH_SHARED_DEVICE=1 puts both ranks on cuda:0 over gloo; everything else (sharding, barrier + sync
    bracket, max over ranks, gather + merge for the DB-sharded mode, the single JSON line from rank 0) is the code
    the 8-GPU run executes."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LENS_BENCH_SHARED_DEVICE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--queries", "256", "--places", "20000", "--shard", shard, "--no-cpu-baseline"]
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["value"] > 0
    assert d["scaling"] == ("weak" if shard == "queries" else "strong")
    assert d["config"]["queries_per_step"] == (512 if shard == "queries" else 256)
