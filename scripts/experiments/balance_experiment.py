"""Experiment (development aid): does dealing the queries to SIMDs by predicted cost shorten k_feature_events?
The frames are permuted on the host so that the kernel's natural slot order realises the deal."""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import pretrained_example
from lens_amd.engine import LensEngine, centre_pixel_index
from lens_amd.synthetic import synthetic_frames, synthetic_database
Q = 4096
w1, w2p = pretrained_example()
frames = synthetic_frames(Q, 128)
eng = LensEngine(w1, synthetic_database(w2p, 1000), 128, 10, 250, max_batch=Q)
codes = eng.codes.cpu().numpy()                      # [T][N] uint8
sel = centre_pixel_index(128, 10)
px = frames.numpy().reshape(Q, -1)[:, sel]           # [Q][N]
table = (codes[:, :, None] <= np.arange(256)[None, None, :]).sum(0)     # [N][256]: active steps for pixel value p
key = table[np.arange(px.shape[1])[None, :], px].sum(1)                  # active (t, n) pairs per query
def time_feature(fr):
    fr = fr.cuda(); eng.feature_events(fr, carry=False); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); eng.feature_events(fr, carry=False); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return min(ts), eng.nwords[:Q].cpu().numpy().astype(np.float64)
t_nat, nw = time_feature(frames)
print("natural order: %.4f ms   corr(key, words) = %.3f" % (t_nat, np.corrcoef(key, nw)[0, 1]))
order = np.argsort(-key, kind="stable")
# slot (wg b, wave w): SIMD g = 4 b + w % 4, round r = w // 4; snake deal of the sorted list over the 1024 SIMDs
perm = np.empty(Q, dtype=np.int64)
for b in range(256):
    for w in range(16):
        g, r = 4 * b + (w % 4), w // 4
        pos = r * 1024 + (g if r % 2 == 0 else 1023 - g)
        perm[b * 16 + w] = order[pos]
t_snake, _ = time_feature(frames[torch.from_numpy(perm)])
# alternative: equal-cost waves share a SIMD (sorted blocks of 4)
perm2 = np.empty(Q, dtype=np.int64)
for b in range(256):
    for w in range(16):
        g, r = 4 * b + (w % 4), w // 4
        perm2[b * 16 + w] = order[4 * g + r]
t_block, _ = time_feature(frames[torch.from_numpy(perm2)])
# oracle predictor: the measured stream length itself
order3 = np.argsort(-nw, kind="stable")
perm3 = np.empty(Q, dtype=np.int64)
for b in range(256):
    for w in range(16):
        g, r = 4 * b + (w % 4), w // 4
        pos = r * 1024 + (g if r % 2 == 0 else 1023 - g)
        perm3[b * 16 + w] = order3[pos]
t_oracle, _ = time_feature(frames[torch.from_numpy(perm3)])
print("snake by active-input count: %.4f ms   sorted blocks: %.4f ms   snake by measured stream length: %.4f ms" % (t_snake, t_block, t_oracle))
