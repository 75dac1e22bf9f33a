"""Experiment (development aid): alternate batches over two HIP streams with separate workspaces, so that one batch's
kernel tails overlap the next batch's kernels.  Prints sequential vs pipelined batch rate."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import pretrained_example
from lens_amd.engine import LensEngine, seqmatch_topk
from lens_amd.synthetic import synthetic_frames, synthetic_database
P = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
Q = 4096
w1, w2p = pretrained_example()
w2 = synthetic_database(w2p, P)
frames = synthetic_frames(Q, 128).cuda()
engs = [LensEngine(w1, w2, 128, 10, 250, max_batch=Q) for _ in range(2)]
streams = [torch.cuda.Stream() for _ in range(2)]
def step(e):
    out = e.forward_counts(frames, carry=False, check_status=False)
    return seqmatch_topk(out, 2, 25)
for e in engs: step(e)
torch.cuda.synchronize()
def run(n, pipelined):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        if pipelined:
            with torch.cuda.stream(streams[i & 1]):
                step(engs[i & 1])
        else:
            step(engs[0])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for _ in range(2):
    print("P=%d sequential %.4f ms/batch   two streams %.4f ms/batch" % (P, run(40, False), run(40, True)))
