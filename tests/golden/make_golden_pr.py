#!/usr/bin/env python3
"""Golden vectors for the `--PR_curve` / `--sad` paths, from the reference's own
lens/src/metrics.py::createPR (imported as is) and lens/src/sad.py::run_sad (skimage.io.imread -> PIL stub,
IO only).  Inputs: the similarity matrix / GT of the bundled example run (tests/golden/example.npz) and the
bundled example PNG folders.   Usage: python tests/golden/make_golden_pr.py -> tests/golden/pr_sad.npz"""
import os
import sys
import types
import tempfile

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
os.environ.setdefault("MPLBACKEND", "Agg")


def main():
    from PIL import Image
    sk = types.ModuleType("skimage"); skio = types.ModuleType("skimage.io"); sk.io = skio
    skio.imread = lambda p: np.array(Image.open(p))
    pt = types.ModuleType("prettytable")
    class PrettyTable:
        def __init__(self): self.field_names = []
        def add_row(self, r): pass
        def __str__(self): return ""
    pt.PrettyTable = PrettyTable
    sys.modules.update({"skimage": sk, "skimage.io": skio, "prettytable": pt})
    sys.path.insert(0, REF)
    from lens.src.metrics import createPR
    import lens.src.sad as sad

    G = np.load(os.path.join(HERE, "example.npz"))
    S, GT = G["S_carry"], G["GTtol"].astype(int)
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        P, R = createPR(S.T, GT.T, tmp, matching="single", n_thresh=100)        # run_model.py:321
        out["lens_P"], out["lens_R"] = np.array(P, np.float64), np.array(R, np.float64)
        ref = os.path.join(REF, "lens/dataset/example/davis128/example-reference")
        qry = os.path.join(REF, "lens/dataset/example/davis128/example-query")
        captured = {}
        real_cdist = sad.torch.cdist
        def spy(a, b, p):
            d = real_cdist(a, b, p)
            captured["dist"] = d[0].numpy().copy()
            return d
        sad.torch.cdist = spy
        try:
            PR, recall = sad.run_sad(ref, qry, GT, tmp, 2)                      # run_model.py:330
        finally:
            sad.torch.cdist = real_cdist
        out["sad_dist"] = captured["dist"].astype(np.float32)
        out["sad_P"], out["sad_R"] = np.array(PR["Precision"], np.float64), np.array(PR["Recall"], np.float64)
        out["sad_recall"] = np.array(recall, np.float64)
        out["files_query"] = np.array([f for f in sorted(os.listdir(qry), key=sad.natural_sort_key) if f.endswith(".png")])
        out["files_reference"] = np.array([f for f in sorted(os.listdir(ref), key=sad.natural_sort_key) if f.endswith(".png")])
    np.savez_compressed(os.path.join(HERE, "pr_sad.npz"), **out)
    print("lens PR points", len(out["lens_P"]), "sad recall", out["sad_recall"], "dist range", out["sad_dist"].min(), out["sad_dist"].max())


if __name__ == "__main__":
    main()
