"""Distribution of the spike-stream lengths (development aid): load balance of the output kernel's waves."""
import sys, os, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import pretrained_example
from lens_amd.engine import LensEngine
from lens_amd.synthetic import synthetic_frames, synthetic_database
w1, w2p = pretrained_example()
frames = synthetic_frames(4096, 128).cuda()
eng = LensEngine(w1, synthetic_database(w2p, 1000), 128, 10, 250, max_batch=4096)
eng.feature_events(frames, carry=False)
nw = eng.nwords[:4096].cpu().numpy().astype(np.float64)
print("words/query mean %.1f std %.1f min %d max %d  (std/mean %.3f)" % (nw.mean(), nw.std(), nw.min(), nw.max(), nw.std() / nw.mean()))
# static assignment of the output kernel: group g = 128 consecutive queries, wave w takes q = g*128 + w + 16*i
per_wave = nw.reshape(32, 8, 16).sum(1)            # [group][wave]
print("static: max wave / mean wave per group: mean %.3f  worst %.3f" % ((per_wave.max(1) / per_wave.mean(1)).mean(), (per_wave.max(1) / per_wave.mean(1)).max()))
print("whole kernel: slowest group / mean group %.3f" % (per_wave.max(1).max() / per_wave.mean()))
