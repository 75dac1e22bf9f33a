"""k_output_counts time at config 3 (P=100000, Q=4096, synthetic frames) for explicit query-group counts.
Usage: python scripts/experiments/groups_sweep.py [P]"""
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import bench
from lens_amd import _lib
from lens_amd.engine import LensEngine, _ptr, _stream
from lens_amd.synthetic import synthetic_frames, synthetic_database

P = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
w1, w2p = bench.pretrained()
Q = 4096
eng = LensEngine(w1, synthetic_database(w2p, P), bench.ROI, bench.DIMS, bench.T_STEPS, device="cuda", max_batch=Q)
fr = synthetic_frames(Q, bench.ROI, seed=1234).cuda()
eng.forward_counts(fr, carry=False)
out = torch.empty(Q, P, device="cuda")
for g in ([int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 4, 8, 12, 16, 24, 32, 64]):
    def run():
        _lib.check(_lib.lib().lens_output_counts(_ptr(eng.events), eng.cap, _ptr(eng.nwords), Q, _ptr(eng.w2t), eng.F, eng.P,
                                                  eng.ldw2, None, 0, g, 0, _ptr(out), P, _stream()), "lens_output_counts")
    run(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): run()
    b.record(); torch.cuda.synchronize()
    print(f"P={P} n_groups={g}: {a.elapsed_time(b) / 5:.3f} ms", flush=True)
