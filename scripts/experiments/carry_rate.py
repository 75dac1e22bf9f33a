"""Carried-state rates (one stream and 16 streams, 256 queries) for the given database sizes, as bench.py reports them,
plus the two layers of one stream timed alone.  python carry_rate.py [P ...]"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import bench
from lens_amd.engine import LensEngine
from lens_amd.synthetic import synthetic_frames, synthetic_database
w1, w2p = bench.pretrained()
fr = synthetic_frames(256, bench.ROI, seed=1234).cuda()
def timed(f, n=3):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for P in [int(x) for x in sys.argv[1:]] or [1000]:
    eng = LensEngine(w1, synthetic_database(w2p, P), bench.ROI, bench.DIMS, bench.T_STEPS, device="cuda", max_batch=256)
    r = bench.carried_rates(eng, fr, f"{P}-place DB")
    print(P, {k: (round(v["value"]), round(v["ms_per_batch"], 3)) for k, v in r.items() if k != "note"}, flush=True)
    eng.reset_state()
    print("   feature layer alone, 256 queries: %.3f ms   output layer alone: %.3f ms" % (
        timed(lambda: eng.feature_events(fr, carry=True)), timed(lambda: eng.output_counts(256, carry=True))), flush=True)
