"""Time of one carried stream of 256 queries (reference semantics) for several chunk sizes of the pipelined form; places = argv[1] (default 1000)."""
import sys, time, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
import bench
from lens_amd.engine import LensEngine, seqmatch_topk
from lens_amd.synthetic import synthetic_frames, synthetic_database
w1, w2p = bench.pretrained()
eng = LensEngine(w1, synthetic_database(w2p, int(sys.argv[1]) if len(sys.argv) > 1 else 1000), bench.ROI, bench.DIMS, bench.T_STEPS, device="cuda", max_batch=256)
fr = synthetic_frames(256, bench.ROI, seed=1234).cuda()
for first, ch in ((16, 16), (8, 16), (8, 32), (4, 32), (8, 64), (16, 64), (4, 16), (32, 32), (1 << 30, 1 << 30)):
    eng.CARRY_FIRST, eng.CARRY_CHUNK = first, ch
    eng.reset_state(); eng.forward_counts(fr, carry=True); eng.reset_state(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        c = eng.forward_counts(fr, carry=True); seqmatch_topk(c, 2, 25)
    torch.cuda.synchronize()
    print(first, ch, (time.perf_counter() - t0) / 3 * 1e3, "ms per 256 queries", flush=True)
