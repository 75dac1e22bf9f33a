"""Carried state, 256 queries as S streams: the layout the library picks (lens_output_layout) and the time of the output
layer alone and of the whole forward pass.  Run once per forced choice to calibrate the cost model of output_plan:
    LENS_AMD_CARRY_SHARED=0 / 1, LENS_AMD_CARRY_L2=0 / 1 (read once per process).   python carry_layouts.py [P ...]"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import bench
from lens_amd.engine import LensEngine
from lens_amd.synthetic import synthetic_frames, synthetic_database
w1, w2p = bench.pretrained()
Q = 256
fr = synthetic_frames(Q, bench.ROI, seed=1234).cuda()
def timed(f, n=3):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
names = {0: "batch", 1: "lone-LDS", 2: "lone-L2", 3: "shared"}
for P in [int(x) for x in sys.argv[1:]] or [1000]:
    eng = LensEngine(w1, synthetic_database(w2p, P), bench.ROI, bench.DIMS, bench.T_STEPS, device="cuda", max_batch=Q)
    for S in (1, 2, 4, 8, 16, 32, 64, 256):
        eng.reset_state(n_streams=S)
        eng.feature_events(fr, carry=True)
        out_ms = timed(lambda: eng.output_counts(Q, carry=True))
        all_ms = timed(lambda: eng.forward_counts(fr, carry=True))
        print(f"P={P:6d} S={S:3d} {names[eng.layout(True)]:9s} tile {eng.tile_places(True):3d}  output {out_ms:7.3f} ms  forward {all_ms:7.3f} ms", flush=True)
