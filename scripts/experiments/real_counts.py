"""Counts in the spike streams of the bundled example's real frames: share of chunks with a count other than 1, and the
share of single-spike events inside those chunks (what a per-event test for "count is 1" could save)."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import bench
from lens_amd.dataset import CustomImageDataset
from lens_amd.engine import LensEngine
from lens_amd.synthetic import synthetic_database
w1, w2p = bench.pretrained()
base = os.path.join(bench.ROOT, "lens", "dataset")
ds = CustomImageDataset(os.path.join(base, "example-query.csv"), os.path.join(base, "example", "davis128", "example-query"))
fr = ds.frames().cuda()
Q = len(fr)
eng = LensEngine(w1, synthetic_database(w2p, 1000), fr.shape[1], bench.DIMS, bench.T_STEPS, device="cuda", max_batch=Q)
eng.forward_counts(fr, carry=False)
nw = eng.nwords[:Q].long().cpu()
ev = eng.events[:Q].view(Q, -1, 2)[:, :int(nw.max()) // 2].cpu()
chunks = mixed = ev_in_mixed = ones_in_mixed = events = steps = 0
hist = torch.zeros(8, dtype=torch.long)
for q in range(Q):
    n = int(nw[q]) // 2
    off, cnt = ev[q, :n, 0], ev[q, :n, 1]
    c = (cnt & 0x7ffffffe).view(torch.float32)           # LAST and EMPTY_NEXT cleared
    is_mixed = (off.view(-1, 16)[:, 0] < 0)
    chunks += n // 16; mixed += int(is_mixed.sum()); events += n; steps += int((cnt < 0).sum())
    cm = c.view(-1, 16)[is_mixed]
    ev_in_mixed += cm.numel(); ones_in_mixed += int((cm == 1.0).sum())
    hist += torch.bincount(c.clamp(max=7).long(), minlength=8)
print(f"queries {Q}: events/query {events / Q:.0f}, steps/query {steps / Q:.0f}, chunks mixed {100.0 * mixed / chunks:.1f} %")
print(f"events in mixed chunks with count 1: {100.0 * ones_in_mixed / max(1, ev_in_mixed):.1f} %")
print("count histogram (0..7+):", [round(100.0 * int(h) / events, 2) for h in hist])
