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
