"""Multi-GPU matching: the place database (rows of output_layer.w.weight) sharded across ranks.

One process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm).  The reference is a
single process (lens/run_model.py); this is the build's MI355X-native addition:

  * every output neuron is one place and is independent given the feature spikes, so W2 rows are
    split contiguously; each shard carries L-1 extra rows so that the diagonal sequence filter
    (lens/run_model.py:248-252) needs no exchange;
  * the feature layer (80 KB of weights) is replicated, every rank recomputes the feature spikes;
  * each rank ranks its own places (top-K with global indices), then ONE collective: an all-gather
    of [Q', K] x (float32 score, int32 index), followed by a K-way merge on every rank.
"""
import torch
import torch.distributed as dist


def shard_bounds(P, L, world):
    """Split the P-L+1 sequence-matched places into `world` contiguous ranges.
    Returns b with len world+1; rank r ranks places [b[r], b[r+1]) and needs W2 rows
    [b[r], b[r+1] + L - 1)."""
    Ps = P - max(L, 1) + 1
    return [r * Ps // world for r in range(world + 1)]


def shard_rows(P, L, rank, world):
    """(first_row, last_row_exclusive, n_ranked_places, place_offset) of this rank's shard."""
    b = shard_bounds(P, L, world)
    p0, p1 = b[rank], b[rank + 1]
    return p0, p1 + max(L, 1) - 1, p1 - p0, p0


def gather_topk(idx, val, group=None):
    """All-gather per-shard top-K lists: idx int32 [Q',K], val float32 [Q',K] -> [world,Q',K] each.
    Works with any backend (RCCL on device tensors; gloo on CPU tensors in the tests)."""
    world = dist.get_world_size(group)
    q, k = idx.shape
    dev = idx.device
    if dev.type == "cuda" and dist.get_backend(group) == "gloo":      # test transport: stage through the host
        idx, val = idx.cpu(), val.cpu()
    idx_all = torch.empty((world * q, k), dtype=idx.dtype, device=idx.device)   # concatenation along dim 0
    val_all = torch.empty((world * q, k), dtype=val.dtype, device=val.device)
    dist.all_gather_into_tensor(idx_all, idx.contiguous(), group=group)
    dist.all_gather_into_tensor(val_all, val.contiguous(), group=group)
    return idx_all.view(world, q, k).to(dev), val_all.view(world, q, k).to(dev)


class ShardedMatcher:
    """This rank's shard of the place database plus the replicated feature layer."""

    def __init__(self, w1, w2_full, roi_dim, dims, timebin, L, K, rank, world, device, group=None, max_batch=4096):
        from .engine import LensEngine
        P = w2_full.shape[0]
        r0, r1, self.n_places, self.offset = shard_rows(P, L, rank, world)
        self.L, self.K, self.group, self.world = L, K, group, world
        self.engine = LensEngine(w1, w2_full[r0:r1], roi_dim, dims, timebin, device=device, max_batch=max_batch)

    def local_topk(self, frames, carry=False):
        from .engine import seqmatch_topk
        counts = self.engine.forward_counts(frames, carry=carry, check_status=False)
        return seqmatch_topk(counts, self.L, self.K, n_places=self.n_places, place_offset=self.offset)

    def match(self, frames, carry=False):
        """frames uint8 [Q,H,W] (same on every rank) -> global top-K idx/val [Q-L+1, K] on every rank."""
        from .engine import topk_merge
        idx, val = self.local_topk(frames, carry=carry)
        if self.world == 1:
            return idx, val
        idx_all, val_all = gather_topk(idx, val, self.group)
        return topk_merge(idx_all, val_all)
