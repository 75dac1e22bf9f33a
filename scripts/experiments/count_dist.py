import sys, os, torch, numpy as np
import os; ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import pretrained_example, load_config
from lens_amd.engine import LensEngine
from lens_amd.synthetic import synthetic_frames, synthetic_database
w1, w2p = pretrained_example()
for name in ("synthetic", "real"):
    if name == "real":
        c, fr, *_ = load_config("example"); frames = torch.from_numpy(fr).cuda(); roi = 80
    else:
        frames = synthetic_frames(512, 128).cuda(); roi = 128
    eng = LensEngine(w1, synthetic_database(w2p, 1000), roi, 10, 250, max_batch=frames.shape[0])
    eng.feature_events(frames, carry=False)
    out = eng.output_counts(frames.shape[0], carry=False)
    ev = eng.events[:frames.shape[0]].cpu().numpy().view(np.uint32); nw = eng.nwords[:frames.shape[0]].cpu().numpy()
    cnts = []
    for q in range(frames.shape[0]):
        w = ev[q, :nw[q]].reshape(-1, 32)[:, 16:].reshape(-1)
        cnts.append(w)
    c = np.concatenate(cnts); mag = (c & 0x7ffffffe).view(np.float32)
    last = (c >> 31) & 1; flag = c & 1
    print(name, "events", c.size, "frac count==1", float((mag == 1.0).mean()), "count==0", float((mag == 0).mean()), ">1", float((mag > 1).mean()), "max", float(mag.max()),
          "LAST frac", float(last.mean()), "flagged frac of LAST", float(flag.sum() / max(last.sum(), 1)))
    o = out.cpu().numpy(); print("  output spikes per place per query: mean", o.mean(), "max", o.max(), " frac of (q,p) with any spike", (o > 0).mean())
