"""Batch pipelining: throughput vs number of streams (development aid; 2-3 streams are equal, 4 collapses)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import pretrained_example
from lens_amd.pipeline import BatchPipeline
from lens_amd.synthetic import synthetic_frames, synthetic_database
w1, w2p = pretrained_example()
w2 = synthetic_database(w2p, 1000)
frames = synthetic_frames(4096, 128).cuda()
for depth in (1, 2, 3, 4, 2, 3):
    p = BatchPipeline(w1, w2, 128, 10, 250, max_batch=4096, depth=depth)
    for _ in range(4): p.submit(frames, 2, 25)
    p.synchronize(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(60): p.submit(frames, 2, 25)
    torch.cuda.synchronize()
    print("depth %d: %.4f ms/batch" % (depth, (time.perf_counter() - t0) / 60 * 1e3))
    del p
