"""How much do HIP event records between the kernels of a step cost? (development aid)"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import pretrained_example
from lens_amd.engine import LensEngine, seqmatch_topk
from lens_amd.synthetic import synthetic_frames, synthetic_database
Q = 4096
w1, w2p = pretrained_example()
eng = LensEngine(w1, synthetic_database(w2p, 1000), 128, 10, 250, max_batch=Q)
frames = synthetic_frames(Q, 128).cuda()
def step(evs, mode):
    if mode == 4: evs[0].record()
    eng.feature_events(frames, carry=False)
    if mode >= 2: evs[1].record()
    counts = eng.output_counts(Q, carry=False)
    if mode >= 2: evs[2].record()
    r = seqmatch_topk(counts, 2, 25)
    if mode == 4: evs[3].record()
    return r
for mode in (0, 2, 4, 0, 2, 4):
    n = 40
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(n)]
    for _ in range(3): step(evs[0], 0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(n): step(evs[k], mode)
    torch.cuda.synchronize()
    print("events per step %d: %.4f ms/step" % (mode, (time.perf_counter() - t0) / n * 1e3))
