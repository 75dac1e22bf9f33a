"""Multi-GPU matching: the place database (rows of output_layer.w.weight) sharded across ranks.

One process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm).  The reference is a
single process (lens/run_model.py); sharding is this build's MI355X-native addition:

  * every output neuron is one place and is independent given the feature spikes, so W2 rows are
    split contiguously; each shard carries L-1 extra rows so that the diagonal sequence filter
    (lens/run_model.py:248-252) needs no exchange;
  * the feature layer (80 KB of weights) is replicated, every rank recomputes the feature spikes;
  * each rank ranks its own places (top-K with global indices), then ONE collective: an all-gather
    of [2, Q', K] int32 (indices, and the float32 scores as bits), followed by a K-way merge on every
    rank.  `ShardedStep` runs the collective and the merge of batch i on a second HIP stream beside
    the kernels of batch i+1.
"""
import torch
import torch.distributed as dist


def shard_bounds(P, L, world):
    """Split the P-L+1 sequence-matched places into `world` contiguous ranges.
    Returns b with len world+1; rank r ranks places [b[r], b[r+1]) and needs W2 rows
    [b[r], b[r+1] + L - 1)."""
    Ps = P - max(L, 1) + 1
    return [r * Ps // world for r in range(world + 1)]


def check_shardable(P, L, world):
    """Raise the same ValueError on every rank (before any collective) when a rank would own no place."""
    if L < 1:
        raise ValueError("sharded matching needs sequence_length >= 1")
    if world < 1 or P - L + 1 < world:
        raise ValueError(f"{P} places with sequence_length {L} cannot be split over {world} ranks "
                         f"(every rank needs at least one sequence-matched place)")


def shard_rows(P, L, rank, world):
    """(first_row, last_row_exclusive, n_ranked_places, place_offset) of this rank's shard."""
    b = shard_bounds(P, L, world)
    p0, p1 = b[rank], b[rank + 1]
    return p0, p1 + max(L, 1) - 1, p1 - p0, p0


def _use_host_staging(t, group):
    return t.device.type == "cuda" and dist.get_backend(group) == "gloo"


def gather_topk(idx, val, group=None):
    """All-gather per-shard top-K lists: idx int32 [Q',K], val float32 [Q',K] -> [world,Q',K] each, with ONE
    collective over the packed pair.  RCCL on device tensors; with the gloo backend (the test transport) the
    pair is staged through host memory with synchronous copies."""
    world = dist.get_world_size(group)
    q, k = idx.shape
    dev = idx.device
    pack = torch.stack((idx, val.view(torch.int32)))                    # [2,Q',K]
    if _use_host_staging(pack, group):
        pack = pack.cpu()
    allp = torch.empty((world * 2, q, k), dtype=torch.int32, device=pack.device)
    dist.all_gather_into_tensor(allp, pack.contiguous(), group=group)
    allp = allp.view(world, 2, q, k).to(dev)
    return allp[:, 0].contiguous(), allp[:, 1].contiguous().view(torch.float32)


class ShardedMatcher:
    """This rank's shard of the place database plus the replicated feature layer."""

    def __init__(self, w1, w2_full, roi_dim, dims, timebin, L, K, rank, world, device, group=None, max_batch=4096):
        from .engine import LensEngine
        P = w2_full.shape[0]
        check_shardable(P, L, world)
        r0, r1, self.n_places, self.offset = shard_rows(P, L, rank, world)
        self.L, self.K, self.group, self.world = L, K, group, world
        self.engine = LensEngine(w1, w2_full[r0:r1], roi_dim, dims, timebin, device=device, max_batch=max_batch)

    def local_topk(self, frames, carry=False):
        from .engine import seqmatch_topk
        counts = self.engine.forward_counts(frames, carry=carry)
        return seqmatch_topk(counts, self.L, self.K, n_places=self.n_places, place_offset=self.offset)

    def match(self, frames, carry=False):
        """frames uint8 [Q,H,W] (same on every rank) -> global top-K idx/val [Q-L+1, K] on every rank."""
        from .engine import topk_merge
        idx, val = self.local_topk(frames, carry=carry)
        if self.world == 1:
            return idx, val
        idx_all, val_all = gather_topk(idx, val, self.group)
        return topk_merge(idx_all, val_all)


class ShardedStep:
    """Batches of the sharded path with the collective off the critical path.

    submit(frames) enqueues feature spikes -> shard counts -> shard top-K on the current stream, then the
    all-gather of the packed top-K and the merge kernel on a second stream, so that they run beside the next
    batch's kernels (the gather is ~0.8 MB per rank and latency-bound).  Two sets of buffers alternate; a set is
    reused only after the second stream has finished with it.  Results are identical to ShardedMatcher.match.
    """

    def __init__(self, engine, L, K, n_places, offset, max_queries, world, group=None, depth=2):
        self.eng, self.L, self.K, self.n_places, self.offset = engine, L, K, n_places, offset
        self.world, self.group = world, group
        dev = engine.device
        qs = max_queries - L + 1
        self.comm = torch.cuda.Stream(device=dev)
        self.sets = []
        for _ in range(depth):
            self.sets.append(dict(
                pack=torch.empty((2, qs, K), dtype=torch.int32, device=dev),
                allp=torch.empty((world, 2, qs, K), dtype=torch.int32, device=dev),
                idx_all=torch.empty((world, qs, K), dtype=torch.int32, device=dev),
                val_all=torch.empty((world, qs, K), dtype=torch.float32, device=dev),
                idx=torch.empty((qs, K), dtype=torch.int32, device=dev),
                val=torch.empty((qs, K), dtype=torch.float32, device=dev),
                ready=torch.cuda.Event(), done=torch.cuda.Event(), used=False))
        self.qs = qs
        self._i = 0
        self._timed = []

    def submit(self, frames, events=None):
        """Returns (idx, val) [Q-L+1, K]: valid once the second stream has been waited for (synchronize())."""
        from .engine import seqmatch_topk, topk_merge
        Q = frames.shape[0]
        assert Q - self.L + 1 == self.qs, "ShardedStep is sized for one batch shape"
        s = self.sets[self._i % len(self.sets)]
        self._i += 1
        cur = torch.cuda.current_stream(self.eng.device)
        if s["used"]:
            cur.wait_event(s["done"])                      # the previous user's gather + merge have read / written it
        if events is not None: events[0].record()
        self.eng.feature_events(frames, carry=False)
        if events is not None: events[1].record()
        counts = self.eng.output_counts(Q, carry=False)
        if events is not None: events[2].record()
        seqmatch_topk(counts, self.L, self.K, n_places=self.n_places, place_offset=self.offset,
                      out=(s["pack"][0], s["pack"][1].view(torch.float32)))
        if events is not None: events[3].record()
        s["ready"].record(cur)
        with torch.cuda.stream(self.comm):
            self.comm.wait_event(s["ready"])
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
            if _use_host_staging(s["pack"], self.group):
                host = s["pack"].cpu()                     # synchronous copies: gloo takes CPU tensors only
                allh = torch.empty((self.world * 2,) + tuple(host.shape[1:]), dtype=torch.int32)
                dist.all_gather_into_tensor(allh, host, group=self.group)
                s["allp"].copy_(allh.view(s["allp"].shape))
            else:
                dist.all_gather_into_tensor(s["allp"].view(self.world * 2, self.qs, self.K), s["pack"], group=self.group)
            s["idx_all"].copy_(s["allp"][:, 0])
            s["val_all"].copy_(s["allp"][:, 1].view(torch.float32))
            topk_merge(s["idx_all"], s["val_all"], out=(s["idx"], s["val"]))
            t1.record()
            s["done"].record(self.comm)
        s["used"] = True
        if len(self._timed) < 64:
            self._timed.append((t0, t1))
        return s["idx"], s["val"]

    def synchronize(self):
        self.comm.synchronize()

    def gather_merge_ms(self):
        """Mean duration of all-gather + merge on the second stream over the recorded batches."""
        self.synchronize()
        ms = [a.elapsed_time(b) for a, b in self._timed]
        return float(sum(ms) / max(1, len(ms)))
