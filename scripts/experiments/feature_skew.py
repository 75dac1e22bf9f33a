"""How skewed is feature activity? (development aid) fraction of events covered by the top-k features."""
import sys, os, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import pretrained_example, load_config
from lens_amd.engine import LensEngine
from lens_amd.synthetic import synthetic_frames, synthetic_database
w1, w2p = pretrained_example()
w1n = np.asarray(w1, np.float32)
static_rank = np.argsort(-np.maximum(w1n, 0).sum(1))          # data-independent heuristic
for name in ("synthetic", "real"):
    if name == "real":
        c, fr, *_ = load_config("example"); frames = torch.from_numpy(fr).cuda(); roi = 80
    else:
        frames = synthetic_frames(1024, 128).cuda(); roi = 128
    eng = LensEngine(w1, synthetic_database(w2p, 1000), roi, 10, 250, max_batch=frames.shape[0])
    spk = eng.feature_events(frames, carry=False, want_spikes=True)      # [Q,T,F]
    act = (spk > 0).sum((0, 1)).cpu().numpy().astype(np.float64)
    tot = act.sum()
    order = np.argsort(-act)
    for k in (100, 128, 152, 176):
        print(name, "top-%d by true activity: %.4f   by static heuristic: %.4f   silent features: %d" % (
            k, act[order[:k]].sum() / tot, act[static_rank[:k]].sum() / tot, int((act == 0).sum())))
