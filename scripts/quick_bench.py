"""Quick per-kernel timing (development aid; bench.py is the contract benchmark)."""
import argparse, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import pretrained_example
from lens_amd.engine import LensEngine, seqmatch_topk
from lens_amd.synthetic import synthetic_frames, synthetic_database

ap = argparse.ArgumentParser()
ap.add_argument("--places", type=int, default=1000)
ap.add_argument("--queries", type=int, default=4096)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--groups", type=int, default=0)
a = ap.parse_args()
w1, w2p = pretrained_example()
w2 = synthetic_database(w2p, a.places)
frames = synthetic_frames(a.queries, 128).cuda()
eng = LensEngine(w1, w2, 128, 10, 250, max_batch=a.queries)
def ev(): return torch.cuda.Event(enable_timing=True)
for it in range(a.iters):
    e = [ev() for _ in range(4)]
    e[0].record(); eng.feature_events(frames, carry=False)
    e[1].record(); out = eng.output_counts(a.queries, carry=False, n_groups=a.groups)
    e[2].record(); idx, val = seqmatch_topk(out, 2, 25)
    e[3].record(); torch.cuda.synchronize()
    tf, to, tk = e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2]), e[2].elapsed_time(e[3])
    tot = tf + to + tk
    print(f"P={a.places} Q={a.queries} feature {tf:.3f} ms  output {to:.3f} ms  topk {tk:.3f} ms  total {tot:.3f} ms  -> {a.queries/tot*1e3:.0f} q/s")
nw = eng.nwords[:a.queries].float()
print("mean words/query", nw.mean().item(), "events/step", nw.mean().item()/250)
