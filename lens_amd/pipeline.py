"""Batch pipelining over two HIP streams (host orchestration of the same kernels).

`LensEngine.forward_counts` runs frames -> spike streams -> counts on one stream; consecutive batches
therefore serialise, and every kernel ends with a tail in which most CUs idle (the feature kernel's slowest
query, the output kernel's last workgroups, the latency-bound top-k).  `BatchPipeline` keeps two engines
(two spike-stream workspaces, shared read-only weights) and alternates batches between two streams, so that
one batch's tails overlap the next batch's kernels.  Results are identical to the single-stream path: the
batches are independent in `reset` mode, which is the only mode offered here (carried state is sequential in
the query index by definition, lens/run_model.py:234-242).

Measured on MI355X (scripts/experiments/pipeline_experiment.py): 0.365 -> 0.328 ms per 4096-query batch at 1000 places,
no gain at 100 000 places (one kernel dominates there).
"""
import torch

from .engine import LensEngine, seqmatch_topk


class BatchPipeline:
    def __init__(self, w1, w2, roi_dim, dims, timebin, device="cuda", max_batch=4096, depth=2):
        dev = torch.device(device)
        self.engines = [LensEngine(w1, w2, roi_dim, dims, timebin, device=dev, max_batch=max_batch)]
        for _ in range(depth - 1):
            e = LensEngine(w1, w2, roi_dim, dims, timebin, device=dev, max_batch=max_batch)
            # the weights and tables are read-only: share the first engine's device copies
            e.w1t, e.w2t, e.sel, e.codes = (self.engines[0].w1t, self.engines[0].w2t, self.engines[0].sel,
                                            self.engines[0].codes)
            self.engines.append(e)
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(depth)]
        self.device = dev
        self._i = 0

    def submit(self, frames, seq_len=None, topk=None, events=None):
        """Enqueue one batch (uint8 [Q,H,W] on the device, Q <= max_batch) on the next stream.  Returns the
        counts [Q,P], or (idx, val) of `seqmatch_topk` when seq_len and topk are given, plus the stream; the
        tensors are valid once that stream (or `synchronize()`) has been waited for.
        events: optional 4 torch.cuda.Event, recorded on that stream before / between / after the three kernels."""
        k = self._i % len(self.engines)
        self._i += 1
        eng, st = self.engines[k], self.streams[k]
        Q = frames.shape[0]
        if Q > eng.max_batch:
            raise ValueError("BatchPipeline.submit takes at most max_batch queries per call")
        st.wait_stream(torch.cuda.current_stream(self.device))     # frames may have been produced there
        with torch.cuda.stream(st):
            if events is not None: events[0].record()
            eng.feature_events(frames, carry=False)
            if events is not None: events[1].record()
            out = eng.output_counts(Q, carry=False)
            if events is not None: events[2].record()
            res = seqmatch_topk(out, seq_len, topk) if (seq_len and topk) else out
            if events is not None: events[3].record()
        frames.record_stream(st)
        return res, st

    def run(self, batches, seq_len=None, topk=None):
        """All batches, results in order, synchronised."""
        results = [self.submit(b, seq_len, topk)[0] for b in batches]
        self.synchronize()
        return results

    def synchronize(self):
        for st in self.streams:
            st.synchronize()
