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
ap.add_argument("--real", action="store_true", help="bundled 80x80 example query frames tiled to --queries (real spike density)")
ap.add_argument("--dense", action="store_true", help="also time the dense f32-MFMA variant of the output layer")
a = ap.parse_args()
w1, w2p = pretrained_example()
w2 = synthetic_database(w2p, a.places)
if a.real:
    from helpers import load_config
    c, fr, *_ = load_config("example")
    reps = (a.queries + len(fr) - 1) // len(fr)
    frames = torch.from_numpy(fr).repeat(reps, 1, 1)[:a.queries].contiguous().cuda()
    roi = 80
else:
    frames = synthetic_frames(a.queries, 128).cuda()
    roi = 128
eng = LensEngine(w1, w2, roi, 10, 250, max_batch=a.queries)
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
if a.dense:
    spk = eng.feature_events(frames, carry=False, want_spikes=True)
    from lens_amd import _lib
    from lens_amd.engine import _ptr, _stream
    out2 = torch.empty((a.queries, eng.P), dtype=torch.float32, device="cuda")
    for it in range(3):
        e0, e1 = ev(), ev()
        e0.record()
        st = _lib.lib().lens_output_counts_dense(_ptr(spk), a.queries, eng.T, eng.F, eng.F, _ptr(eng.w2t), eng.P, eng.ldw2, _ptr(out2), eng.P, _stream())
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        fl = 2.0 * 250 * eng.F * eng.P * a.queries
        print(f"dense f32-MFMA output layer: {ms:.3f} ms -> {a.queries/ms*1e3:.0f} q/s ({fl/ms/1e9:.1f} TFLOP/s dense, {fl/ms/1e9/157.3*100:.1f}% of the 157.3 TF f32 MFMA peak); identical: {bool(torch.equal(out, out2))}")
nw = eng.nwords[:a.queries].float()
print("mean words/query", nw.mean().item(), "events/step", nw.mean().item()/250)
