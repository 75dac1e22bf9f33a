#!/usr/bin/env python3
"""
Generate the golden vectors under tests/golden/ by running the REFERENCE's own Python modules
(imported from /root/reference -- available in the build container only; nothing here runs on
the GPU box, only the .npz outputs travel).

What is real and what is a stand-in
-----------------------------------
Real reference code executed unmodified:
    main.py (argparse, generate_model_name, initialize_and_run_model), lens/run_model.py
    (LENS, evaluate, run_inference), lens/src/dataset.py (CustomImageDataset, ProcessImage,
    SetImageAsSpikes -- the rate-coding raster), lens/src/blitnet.py (SNNLayer),
    lens/src/metrics.py (recallAtK), lens/src/loggers.py, lens/tools/checker.py,
    torch's own nn.Conv2d / nn.Linear (MKL) / conv2d, scipy's binary_dilation.
IO / plotting stand-ins (no arithmetic): torchvision.io.read_image (PIL), torchvision.transforms
    .Compose, seaborn.heatmap, prettytable.PrettyTable, skimage.io.imread.
Arithmetic stand-in: `sinabs` (absent, cannot be installed).  The stand-in below restates
    sinabs.from_torch.from_model / IAFSqueeze / FlattenTime in torch ops.  Hence the IAF rule is
    NOT pinned by these vectors ("parity unpinned" at that boundary); everything around it is.

Usage:  python tests/golden/make_golden.py            (writes tests/golden/*.npz)
"""
import os
import sys
import types
import tempfile
import hashlib

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
os.environ.setdefault("MPLBACKEND", "Agg")


# ----------------------------------------------------------------------------- stubs
def _install_stubs(capture):
    import torch
    import torch.nn as nn
    from PIL import Image

    def mod(name):
        m = types.ModuleType(name)
        sys.modules[name] = m
        return m

    tv = mod("torchvision"); tvio = mod("torchvision.io"); tvt = mod("torchvision.transforms")
    tv.io = tvio; tv.transforms = tvt

    def read_image(path):
        a = np.array(Image.open(path))
        assert a.ndim == 2 and a.dtype == np.uint8
        return torch.from_numpy(a).unsqueeze(0)
    tvio.read_image = read_image

    class Compose:
        def __init__(self, ts): self.ts = ts
        def __call__(self, x):
            for t in self.ts: x = t(x)
            return x
    tvt.Compose = Compose

    sns = mod("seaborn"); sns.heatmap = lambda *a, **k: None

    pt = mod("prettytable")
    class PrettyTable:
        def __init__(self): self.field_names = []; self.rows = []
        def add_row(self, r): self.rows.append(r)
        def __str__(self): return " | ".join(map(str, self.field_names)) + "\n" + "\n".join(" | ".join(map(str, r)) for r in self.rows)
    pt.PrettyTable = PrettyTable

    sk = mod("skimage"); skio = mod("skimage.io"); sk.io = skio
    skio.imread = lambda p: np.array(Image.open(p))

    # ---- sinabs stand-in (torch ops, mirrors sinabs 2.x defaults chosen by from_model) ----
    sin = mod("sinabs"); sl = mod("sinabs.layers"); sft = mod("sinabs.from_torch")
    sb = mod("sinabs.backend"); sbd = mod("sinabs.backend.dynapcnn"); sbc = mod("sinabs.backend.dynapcnn.chip_factory")
    sin.layers = sl; sin.from_torch = sft; sin.backend = sb; sb.dynapcnn = sbd; sbd.chip_factory = sbc

    class FlattenTime(nn.Flatten):
        def __init__(self): super().__init__(0, 1)
    sl.FlattenTime = FlattenTime

    class IAFSqueeze(nn.Module):
        def __init__(self, num_timesteps, name=""):
            super().__init__()
            self.T = num_timesteps; self.v = None; self.tag = name
            self.spike_threshold = torch.tensor(1.0); self.min_v_mem = torch.tensor(-1.0)
        def forward(self, x):
            T = self.T
            x = x.reshape(-1, T, *x.shape[1:])
            if capture.get("reset") or self.v is None or self.v.shape != x[:, 0].shape:
                self.v = torch.zeros_like(x[:, 0])
            v = self.v
            outs = []
            for t in range(T):
                v = 1.0 * v + x[:, t]
                s = (v > 0) * torch.div(v, self.spike_threshold, rounding_mode="trunc").float()
                v = v - s * self.spike_threshold
                v = torch.nn.functional.relu(v - self.min_v_mem) + self.min_v_mem
                outs.append(s)
            self.v = v
            out = torch.stack(outs, 1).flatten(0, 1)
            capture.setdefault("layers", {}).setdefault(self.tag, []).append(out.detach().clone())
            return out
    sl.IAFSqueeze = IAFSqueeze

    class Network(nn.Module):
        def __init__(self, spiking_model):
            super().__init__(); self.spiking_model = spiking_model
        def forward(self, x):
            y = self.spiking_model(x)
            capture.setdefault("rows", []).append(y.sum(dim=0).squeeze().detach().clone())
            return y

    def from_model(model, input_shape=None, num_timesteps=None, add_spiking_output=False, **kw):
        import copy
        m = copy.deepcopy(model)
        layers = list(m.children())
        if add_spiking_output and isinstance(layers[-1], (nn.Linear, nn.Conv2d)):
            layers.append(nn.ReLU())
        k = 0
        for i, l in enumerate(layers):
            if isinstance(l, nn.ReLU):
                layers[i] = IAFSqueeze(num_timesteps, name=f"iaf{k}"); k += 1
        return Network(nn.Sequential(*layers))
    sft.from_model = from_model
    sbd.DynapcnnNetwork = type("DynapcnnNetwork", (), {})
    sbc.ChipFactory = type("ChipFactory", (), {})


# ----------------------------------------------------------------------------- run
def run_reference(argv, capture):
    """Run the reference CLI in a scratch cwd that exposes ./lens/models and ./lens/dataset."""
    import importlib
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.makedirs(os.path.join(tmp, "lens"))
        os.symlink(os.path.join(REF, "lens", "models"), os.path.join(tmp, "lens", "models"))
        os.symlink(os.path.join(REF, "lens", "dataset"), os.path.join(tmp, "lens", "dataset"))
        os.chdir(tmp)
        try:
            rm = importlib.import_module("lens.run_model")
            real_recall = rm.recallAtK
            def spy(S, GT, GTsoft=None, K=1):
                capture["S"] = np.array(S); capture["GTtol"] = np.array(GT)
                r = real_recall(S, GT, GTsoft, K) if GTsoft is not None else real_recall(S, GT, K=K)
                capture.setdefault("recall_raw", []).append(float(r))
                return r
            rm.recallAtK = spy
            # num_workers=8 persistent workers are an IO detail; run the same dataset in-process
            real_DL = rm.DataLoader
            rm.DataLoader = lambda ds, **kw: real_DL(ds, batch_size=1, shuffle=False, num_workers=0)
            main = importlib.import_module("main")
            old = sys.argv
            sys.argv = ["main.py"] + argv
            try:
                main.parse_network()
            finally:
                sys.argv = old
                rm.recallAtK = real_recall
                rm.DataLoader = real_DL
        finally:
            os.chdir(cwd)


CAPTURE = {}


def one_config(tag, argv):
    res = {}
    for mode in ("carry", "reset"):
        capture = CAPTURE                      # the stubs installed once close over this dict
        capture.clear()
        capture["reset"] = mode == "reset"
        run_reference(argv, capture)
        out = np.stack([r.numpy() for r in capture["rows"]])
        assert np.all(out == np.round(out)) and out.max() < 32767
        res[f"out_{mode}"] = out.astype(np.int16)
        res[f"S_{mode}"] = capture["S"].astype(np.float32)
        res[f"recall_{mode}"] = np.array(capture["recall_raw"][:6], np.float64)
        if mode == "carry":
            res["GTtol"] = capture["GTtol"].astype(np.int8)
            # feature-layer spikes of queries 0, 1 and the last one (bit-exact kernel debugging)
            feat = capture["layers"]["iaf1"]
            pick = [0, 1, len(feat) - 1]
            res["feat_queries"] = np.array(pick, np.int32)
            res["feat_spk"] = np.stack([feat[i].numpy() for i in pick]).astype(np.int8)
            # IAF#1 must be the identity on the binary conv output
            res["iaf0_is_binary"] = np.array([int(all(bool(((x == 0) | (x == 1)).all()) for x in capture["layers"]["iaf0"]))])
        print(tag, mode, "recall", np.round(res[f"recall_{mode}"], 4), "max count", out.max())
    np.savez_compressed(os.path.join(HERE, f"{tag}.npz"), **res)


def raster_golden():
    """Real lens/src/dataset.py items -> the N centre pixels' spike raster, plus U fingerprints."""
    import torch
    from lens.src.dataset import CustomImageDataset, ProcessImage
    import torchvision.transforms as transforms
    out = {}
    for tag, csv, d, k, dims, T in (("example", "example-query.csv", "example/davis128/example-query", 8, 10, 250),
                                   ("sunset", "sunset1.csv", "brisevent/davis346/sunset1", 1, 7, 250)):
        ds = CustomImageDataset(annotations_file=os.path.join(REF, "lens/dataset", csv),
                                img_dir=os.path.join(REF, "lens/dataset", d),
                                transform=transforms.Compose([ProcessImage()]), kernel_size=k, skip=1,
                                max_samples=3, is_spiking=True, time_window=T)
        c0 = (k // 2 - 1) % k          # k == 1: the reference's kernel[-1, -1] is Python negative indexing
        sel = []
        for i in range(3):
            spikes = ds[i][0]                      # [T,1,H,W] float {0,1}
            assert spikes.shape[0] == T
            sel.append(spikes[:, 0, c0::k, c0::k][:, :dims, :dims].reshape(T, -1).numpy().astype(np.uint8))
        out[f"{tag}_sel_raster"] = np.packbits(np.stack(sel), axis=-1)
        out[f"{tag}_sel_shape"] = np.array(np.stack(sel).shape)
    # fingerprints of the global-seed rand tensor the reference draws (dataset.py:120-121)
    for (T, HW) in ((250, 6400), (250, 16384), (250, 49)):
        torch.manual_seed(50)
        U = torch.rand(T, HW).numpy()
        out[f"U_{T}_{HW}_sha256"] = np.frombuffer(hashlib.sha256(U.tobytes()).digest(), np.uint8)
        out[f"U_{T}_{HW}_head"] = U[:2, :8].copy()
    np.savez_compressed(os.path.join(HERE, "raster.npz"), **out)
    print("raster golden written")


if __name__ == "__main__":
    sys.path.insert(0, REF)
    os.environ["TQDM_DISABLE"] = "1"
    _install_stubs(CAPTURE)
    raster_golden()
    if "--raster-only" in sys.argv:
        sys.exit(0)
    one_config("example", ["--matching", "--nocuda"])
    one_config("sunset", ["--matching", "--nocuda", "--dataset", "brisevent", "--camera", "davis346",
                          "--reference", "sunset2", "--query", "sunset1", "--reference_places", "641",
                          "--query_places", "724", "--dims", "7", "--roi_dim", "7", "--feature_multiplier", "1.3"])
