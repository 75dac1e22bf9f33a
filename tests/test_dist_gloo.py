"""World-size-2 (and 3) gloo test of the sharded-database path on CPU: shard arithmetic with the
(L-1)-column halo, the all-gather of per-shard top-K, and the merge rule.  The per-shard ranking and
the merge use the oracle here (no GPU); on the GPU the same plumbing (lens_amd.dist) carries the HIP
kernels' outputs over RCCL (tests/test_gpu_parity.py::test_sharded_topk_merge_equals_unsharded)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _merge_lists(idx_all, val_all, K):
    """Oracle-side merge: rank all candidates by (score desc, index desc), keep K."""
    R, Q, _ = idx_all.shape
    out_i = np.full((Q, K), -1, np.int32)
    out_v = np.full((Q, K), -np.inf, np.float32)
    for q in range(Q):
        cand = [(float(val_all[r, q, k]), int(idx_all[r, q, k])) for r in range(R) for k in range(idx_all.shape[2])
                if idx_all[r, q, k] >= 0]
        cand.sort(key=lambda t: (-t[0], -t[1]))
        for k, (v, i) in enumerate(cand[:K]):
            out_i[q, k], out_v[q, k] = i, v
    return out_i, out_v


def _worker(rank, world, port, out, L, K, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from lens_amd.dist import shard_rows, gather_topk
        from oracle import oracle as O
        P = out.shape[1]
        r0, r1, n_places, offset = shard_rows(P, L, rank, world)
        shard = np.ascontiguousarray(out[:, r0:r1])               # counts this rank would have computed
        S = O.seqmatch(shard, L)[:n_places]                        # only its own places are ranked
        idx, val = O.topk(S, K)
        idx = np.where(idx >= 0, idx + offset, -1).astype(np.int32)
        ia, va = gather_topk(torch.from_numpy(idx), torch.from_numpy(val))
        mi, mv = _merge_lists(ia.numpy(), va.numpy(), K)
        full_i, full_v = O.topk(O.seqmatch(out, L), K)
        ok = np.array_equal(mi, full_i) and np.array_equal(mv, full_v)
        t = torch.tensor([int(ok)])
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        if rank == 0:
            ret.put(int(t.item()))
    finally:
        dist.destroy_process_group()


# the last case is the benchmark's multi-GPU shape: 100 000 places over 8 ranks (12 500 places + 1 halo row each,
# except that the 99 999 sequence-matched places do not divide evenly), L = 2, top-25
@pytest.mark.parametrize("world,P,L,K", [(2, 301, 2, 25), (3, 97, 3, 10), (2, 40, 1, 25), (8, 100000, 2, 25)])
def test_sharded_topk_allgather_merge(world, P, L, K):
    rng = np.random.default_rng(world * 100 + P)
    out = rng.integers(0, 5, size=(17, P)).astype(np.float32)       # many ties
    if P >= 1000:                                                   # ... and winners that straddle shard borders
        from lens_amd.dist import shard_bounds
        for b in shard_bounds(P, L, world)[1:-1]:
            out[:, b - 2:b + 2] += rng.integers(20, 23, size=(17, 4)).astype(np.float32)
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, out, L, K, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    assert ret.get(timeout=5) == 1
