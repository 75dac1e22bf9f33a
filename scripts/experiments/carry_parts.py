"""Times of the two layers of one carried stream (256 queries; places = argv[1], default 1000), each launched alone."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
import bench
from lens_amd.engine import LensEngine
from lens_amd.synthetic import synthetic_frames, synthetic_database
w1, w2p = bench.pretrained()
Q = 256
P = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
eng = LensEngine(w1, synthetic_database(w2p, P), bench.ROI, bench.DIMS, bench.T_STEPS, device="cuda", max_batch=Q)
fr = synthetic_frames(Q, bench.ROI, seed=1234).cuda()
def timed(f, n=3):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("feature layer, one pass :", timed(lambda: eng.feature_events(fr, carry=True)), "ms")
print("output layer            :", timed(lambda: eng.output_counts(Q, carry=True)), "ms")
