"""The complete sharded path (ShardedMatcher / ShardedStep: HIP kernels per shard -> all-gather of top-K -> merge
kernel) with two ranks.  On a one-GPU box both ranks use cuda:0 and the collective travels over gloo (host
staging); where torch.cuda.device_count() >= 2 the `rccl` cases run the same code one rank per GPU over RCCL
(backend "nccl") and are skipped otherwise.  A group of ONE rank over RCCL runs everywhere: it takes the device-tensor
collective path that the multi-GPU runs take."""
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


MULTI_GPU = torch.cuda.device_count() >= 2      # counting devices does not initialise the GPU
needs_two_gpus = pytest.mark.skipif(not MULTI_GPU, reason="needs two GPUs (RCCL between ranks)")


def _worker(rank, world, port, P, Q, ret, backend="gloo"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = f"cuda:{rank}" if backend == "nccl" else "cuda:0"
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
        from helpers import pretrained_example
        from lens_amd.dist import ShardedMatcher, ShardedStep
        from lens_amd.engine import LensEngine, seqmatch_topk
        from lens_amd.synthetic import synthetic_frames, synthetic_database
        w1, w2p = pretrained_example()
        w2 = synthetic_database(w2p, P)
        frames = synthetic_frames(Q, 128, seed=3).to(dev)
        m = ShardedMatcher(torch.from_numpy(w1), w2, 128, 10, 250, L=2, K=25, rank=rank, world=world, device=dev)
        idx, val = m.match(frames)
        full = LensEngine(w1, w2, 128, 10, 250, device=dev)
        fi, fv = seqmatch_topk(full.forward_counts(frames), 2, 25)
        ok = bool(torch.equal(idx, fi) and torch.equal(val, fv))
        # the overlapped form: three different batches through the two alternating buffer sets
        step = ShardedStep(m.engine, 2, 25, m.n_places, m.offset, Q, world)
        batches = [synthetic_frames(Q, 128, seed=40 + i).to(dev) for i in range(3)]
        first = step.submit(batches[0])
        step.synchronize()
        got = [tuple(t.clone() for t in first)]
        pend = [step.submit(b) for b in batches[1:]]          # two in flight: the second reuses no live buffer
        step.synchronize()
        got += [tuple(t.clone() for t in p) for p in pend]
        for b, (gi, gv) in zip(batches, got):
            ri, rv = seqmatch_topk(full.forward_counts(b), 2, 25)
            ok = ok and bool(torch.equal(gi, ri) and torch.equal(gv, rv))
        ok = ok and step.gather_merge_ms() > 0
        t = torch.tensor([int(ok)], device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        if rank == 0:
            ret.put(int(t.item()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("backend", ["gloo", pytest.param("nccl", marks=needs_two_gpus)])
def test_two_rank_sharded_matcher_equals_single_device(backend):
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 1500, 24, ret, backend)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert ret.get(timeout=5) == 1


def test_single_rank_group_over_rccl():
    """The device-tensor collective path (no host staging) through RCCL itself, on one GPU: a process group of one
    rank with backend "nccl".  The all-gather of the packed top-K block, the all-reduce and the ordering of the side
    stream against the kernels' stream are the calls a multi-GPU run makes; only the peers are missing."""
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    p = ctx.Process(target=_worker, args=(0, 1, _free_port(), 1500, 24, ret, "nccl"))
    p.start()
    p.join(300)
    assert p.exitcode == 0
    assert ret.get(timeout=5) == 1


@pytest.mark.parametrize("backend", ["gloo", pytest.param("nccl", marks=needs_two_gpus)])
@pytest.mark.parametrize("shard", ["queries", "db"])
def test_bench_two_ranks_the_way_the_driver_launches_it(shard, backend):
    """bench.py under torch.distributed.run with 2 ranks.  On a one-GPU box the test hook
    LENS_BENCH_SHARED_DEVICE=1 puts both ranks on cuda:0 over gloo; with two GPUs the `nccl` case runs it exactly as
    the driver does, one rank per GPU over RCCL.  Everything else (sharding, barrier + sync bracket, max over ranks,
    overlapped gather + merge for the DB-sharded mode, the in-run parity of the merged top-25 against the oracle, the
    single JSON line from rank 0) is the code the 8-GPU run executes."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    if backend == "gloo":
        env["LENS_BENCH_SHARED_DEVICE"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--queries", "256", "--places", "20000", "--shard", shard]
    if shard == "queries":
        cmd.append("--no-cpu-baseline")
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["value"] > 0
    assert d["scaling"] == ("weak" if shard == "queries" else "strong")
    assert d["config"]["queries_per_step"] == (512 if shard == "queries" else 256)
    assert d["multi_gpu"]["ranks_seen"] == [0, 1] and len(d["multi_gpu"]["per_rank_elapsed_s"]) == 2
    assert d["multi_gpu"]["backend"] == backend
    assert "cpu_baseline" not in d          # timed at N=1 only: here the peers spin in the final barrier meanwhile
    if shard == "db":
        assert d["multi_gpu"]["gather_merge_ms"] > 0
        assert d["parity"]["mismatches"] == 0 and d["parity"]["top25_idx_identical"]


def test_bench_db_sharded_step_over_rccl_with_one_rank():
    """bench.py's DB-sharded mode under torch.distributed.run with ONE rank that forms an RCCL group of its own
    (test hook LENS_BENCH_GROUP_OF_ONE=1): barriers, the all-reduce / all-gather of the timings on device tensors and
    the overlapped top-K all-gather + merge all go through RCCL, which the two-rank cases cannot on a one-GPU box."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", LENS_BENCH_GROUP_OF_ONE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
           "--queries", "256", "--places", "20000", "--shard", "db", "--no-extras"]
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["shard"] == "db"
    assert d["multi_gpu"]["backend"] == "nccl" and d["multi_gpu"]["ranks_seen"] == [0]
    assert d["multi_gpu"]["gather_merge_ms"] > 0
    assert d["parity"]["mismatches"] == 0 and d["parity"]["top25_idx_identical"]
