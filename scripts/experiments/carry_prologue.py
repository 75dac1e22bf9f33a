"""Fixed cost per query of the carried output layer's lone wavefronts: the layer alone on 256 queries of the synthetic
frames and of EMPTY frames (streams of one chunk).  Measured: 0.69 us per query on LDS tiles (of 14.1), 1.38 us through L2
(of 18.1) -- what overlapping a query's first loads with the previous query's last chunks could at most return."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
import bench
from lens_amd.engine import LensEngine
from lens_amd.synthetic import synthetic_frames, synthetic_database
w1, w2p = bench.pretrained()
Q = 256
for P in (1000, 100000):
    eng = LensEngine(w1, synthetic_database(w2p, P), bench.ROI, bench.DIMS, bench.T_STEPS, device="cuda", max_batch=Q)
    for name, fr in (("synthetic", synthetic_frames(Q, bench.ROI, seed=1234).cuda()), ("empty frames", torch.zeros((Q, bench.ROI, bench.ROI), dtype=torch.uint8, device="cuda"))):
        eng.reset_state(); eng.feature_events(fr, carry=True)
        def f(): eng.output_counts(Q, carry=True)
        f(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): f()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        print(f"P={P} {name}: output layer alone {dt*1e3:.3f} ms = {dt/Q*1e6:.2f} us per query; words/query {float(eng.nwords[:Q].float().mean()):.0f}")
