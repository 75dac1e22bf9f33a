"""How many of the F feature neurons ever spike over a benchmark batch (and per query)?"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import bench
from lens_amd.engine import LensEngine
from lens_amd.synthetic import synthetic_frames, synthetic_database
w1, w2p = bench.pretrained()
Q = 4096
eng = LensEngine(w1, synthetic_database(w2p, 1000), bench.ROI, bench.DIMS, bench.T_STEPS, device="cuda", max_batch=Q)
fr = synthetic_frames(Q, bench.ROI, seed=1234).cuda()
eng.forward_counts(fr, carry=False)
nw = eng.nwords[:Q].long()
ev = eng.events[:Q].view(Q, -1, 2)[:, :int(nw.max()) // 2]
valid = torch.arange(ev.shape[1], device="cuda")[None, :] < (nw[:, None] // 2)
f = ((ev[..., 0].long() & 0x7fffffff) // eng.ev_scale)
cnt = ev[..., 1]
real = valid & ~((cnt == 0x3f800000) & (f == 0) & (torch.arange(ev.shape[1], device="cuda")[None, :] >= (nw[:, None] // 2 - 16)))
hist = torch.bincount(f[valid].flatten(), minlength=eng.F)
print("F", eng.F, "features with any event over the batch:", int((hist > 0).sum()))
print("sorted counts (top 20, bottom 20):", sorted(hist.tolist(), reverse=True)[:20], sorted(hist.tolist())[:20])
per_q = torch.stack([(torch.bincount(f[q][valid[q]], minlength=eng.F) > 0).sum() for q in range(64)])
print("active features per query (first 64): mean", float(per_q.float().mean()), "max", int(per_q.max()))
cs = torch.sort(hist, descending=True).values.cumsum(0).float() / hist.sum()
for k in (50, 100, 128, 150, 156, 180):
    print(f"top {k} features carry {float(cs[k-1])*100:.2f} % of the events")
