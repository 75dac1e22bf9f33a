"""Histogram of events per executed time step in the benchmark's spike streams."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import bench
from lens_amd.engine import LensEngine
from lens_amd.synthetic import synthetic_frames, synthetic_database
w1, w2p = bench.pretrained()
Q = 1024
eng = LensEngine(w1, synthetic_database(w2p, 1000), bench.ROI, bench.DIMS, bench.T_STEPS, device="cuda", max_batch=Q)
fr = synthetic_frames(Q, bench.ROI, seed=1234).cuda()
eng.forward_counts(fr, carry=False)
nw = eng.nwords[:Q].long().cpu()
ev = eng.events[:Q].view(Q, -1, 2)[:, :int(nw.max()) // 2].cpu()
hist = torch.zeros(64, dtype=torch.long)
firsts_at_chunk_start = 0; steps = 0
for q in range(Q):
    n = int(nw[q]) // 2
    last = (ev[q, :n, 1] < 0).nonzero().flatten()          # int32 view: sign bit set = LAST
    if len(last) == 0: continue
    sizes = torch.diff(torch.cat([torch.tensor([-1]), last]))
    hist += torch.bincount(sizes.clamp(max=63), minlength=64)
    steps += len(last)
tot = int(hist.sum())
print("executed steps per query", steps / Q)
print("events per step: " + ", ".join(f"{k}:{100.0 * int(hist[k]) / tot:.1f}%" for k in range(1, 16)))
print("mean", float((hist * torch.arange(64)).sum()) / tot)
