#!/usr/bin/env python3
"""
Golden vectors for the training path, produced by the REFERENCE's
own code: lens/train_model.py (LENS_Trainer, train_layer, _anneal_learning_rate), lens/src/blitnet.py
(SNNLayer training-mode init, clamp_spikes, calc_stdp) and lens/src/dataset.py (CustomImageDataset with
test=False + ProcessImage(is_train=True)), imported from /root/reference in the build container.
No arithmetic is stubbed (this path does not use sinabs): only torchvision's PNG reader / Compose.
The only change to the run is the sample ORDER: the reference shuffles with an unseeded DataLoader;
here a recorded, seeded order is fed so that the run is reproducible.

Usage: python tests/golden/make_golden_train.py      -> tests/golden/train_example.npz
"""
import os
import sys
import types
import tempfile
import argparse

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
os.environ["TQDM_DISABLE"] = "1"
os.environ.setdefault("MPLBACKEND", "Agg")


def install_io_stubs():
    import torch
    from PIL import Image
    tv = types.ModuleType("torchvision"); tvio = types.ModuleType("torchvision.io"); tvt = types.ModuleType("torchvision.transforms")
    sys.modules.update({"torchvision": tv, "torchvision.io": tvio, "torchvision.transforms": tvt})
    tv.io, tv.transforms = tvio, tvt
    tvio.read_image = lambda p: torch.from_numpy(np.array(Image.open(p))).unsqueeze(0)

    class Compose:
        def __init__(self, ts): self.ts = ts
        def __call__(self, x):
            for t in self.ts: x = t(x)
            return x
    tvt.Compose = Compose


def main(places=40, epochs=3, seed=1234, chain_linear=False, tag="train_example"):
    """chain_linear=False: the reference exactly as it is (torch's nn.Linear, MKL summation order).
    chain_linear=True: identical run, except that LENS_Trainer.forward's `layer.w(spikes)` is evaluated as an
    ascending-index f32 fmaf chain (the summation order the build fixes, oracle/lens_oracle.c) -- every other
    operation is still the reference's own torch code.  The first variant bounds the effect of torch's
    unspecified GEMV order (compared with a tolerance), the second pins the restatement bit for bit."""
    import torch
    install_io_stubs()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import lens.src.blitnet as bn
    import lens.train_model as tm
    from lens.src.dataset import CustomImageDataset, ProcessImage
    import torchvision.transforms as transforms

    a = argparse.Namespace(
        dataset="example", camera="davis128", data_name="experiment001", reference="example-reference",
        query="example-query", data_dir=os.path.join(REF, "lens/dataset/"), reference_places=places, query_places=places,
        sequence_length=2, feature_multiplier=2.0, filter=1, epoch_feat=epochs, epoch_out=epochs,
        thr_l_feat=0, thr_h_feat=0.75, fire_l_feat=0.4, fire_h_feat=0.6, ip_rate_feat=0.02, stdp_rate_feat=0.01,
        thr_l_out=0, thr_h_out=0.5, fire_l_out=0.5, fire_h_out=0.5, ip_rate_out=0.02, stdp_rate_out=0.01,
        f_exc=0.35, f_inh=0.75, o_exc=1.0, o_inh=1.0, dims=10, roi_dim=80, train_model=True, GT_tolerance=3,
        sim_mat=False, PR_curve=False, matching=False, sad=False, timebin=250, nocuda=True, event_driven=False,
        simulated_speck=False, collect_data=False, headless=False, save_input=False)
    cwd = os.getcwd()
    out = {}
    real_forward = tm.LENS_Trainer.forward
    if chain_linear:
        sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
        from oracle import oracle as O

        def forward_chain(self, spikes, layer):
            x = np.ascontiguousarray(spikes.detach().numpy().reshape(-1), np.float32)
            w = np.ascontiguousarray(layer.w.weight.detach().numpy(), np.float32)
            y = np.zeros(w.shape[0], np.float32)
            O.lib().lens_oracle_linear(x.ctypes.data, w.ctypes.data, y.ctypes.data, w.shape[1], w.shape[0])
            return torch.from_numpy(y).reshape(*spikes.shape[:-1], -1)
        tm.LENS_Trainer.forward = forward_chain
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            torch.manual_seed(seed); np.random.seed(seed)
            model = tm.LENS_Trainer(a)
            ds = CustomImageDataset(annotations_file=model.dataset_file, img_dir=model.reference_dir,
                                    kernel_size=a.roi_dim // a.dims, transform=transforms.Compose([ProcessImage(is_train=True)]),
                                    skip=a.filter, max_samples=a.reference_places, test=False)
            items = [ds[i] for i in range(len(ds))]
            X = torch.stack([it[0] for it in items])                 # [S, N] float32 in [0,1]
            labels = np.array([int(it[1]) for it in items], np.int64)
            out["x_table"] = X.numpy()
            out["labels"] = labels
            for lname in ("feature_layer", "output_layer"):
                L = getattr(model, lname)
                out[f"{lname}_w0"] = L.w.weight.detach().numpy().copy()
                out[f"{lname}_thr0"] = L.thr.detach().numpy().copy()
                out[f"{lname}_fire_rate"] = L.fire_rate.numpy().copy()
                out[f"{lname}_exc"] = L.havconnCombinedExc.numpy().copy()
                out[f"{lname}_inh"] = L.havconnCombinedInh.numpy().copy()
                out[f"{lname}_eta0"] = np.array([float(L.eta_ip), float(L.eta_stdp)], np.float32)

            g = torch.Generator(); g.manual_seed(seed + 1)
            prev = []
            for lname, ep in (("feature_layer", a.epoch_feat), ("output_layer", a.epoch_out)):
                L = getattr(model, lname)
                orders = [torch.randperm(len(items), generator=g).tolist() for _ in range(ep)]
                out[f"{lname}_order"] = np.array(orders, np.int32)

                class Loader:
                    def __init__(self): self.e = 0
                    def __iter__(self):
                        o = orders[self.e]; self.e += 1
                        for i in o:
                            yield X[i].unsqueeze(0), torch.tensor([labels[i]]), None, None
                snaps, etas, step = {}, [], [0]
                real = bn.calc_stdp

                def spy(pre, spk, noclp, layer, idx, prev_layer=None):
                    etas.append((float(layer.eta_ip), float(layer.eta_stdp)))
                    r = real(pre, spk, noclp, layer, idx, prev_layer=prev_layer)
                    step[0] += 1
                    if step[0] in (1, 2, 10, 11, 37):
                        snaps[step[0]] = (layer.w.weight.detach().numpy().copy(), layer.thr.detach().numpy().copy())
                    return r
                bn.calc_stdp = spy
                try:
                    model.train_layer(Loader(), L, prev_layers=list(prev))
                finally:
                    bn.calc_stdp = real
                out[f"{lname}_eta_used"] = np.array(etas, np.float32)       # (eta_ip, eta_stdp) in effect at each step
                for k, (w, t) in snaps.items():
                    out[f"{lname}_w_step{k}"] = w
                    out[f"{lname}_thr_step{k}"] = t
                out[f"{lname}_w_final"] = L.w.weight.detach().numpy().copy()
                out[f"{lname}_thr_final"] = L.thr.detach().numpy().copy()
                prev.append(lname)
                print(lname, "steps", step[0], "w range", float(L.w.weight.min()), float(L.w.weight.max()),
                      "thr max", float(L.thr.max()), "nonzero", float((L.w.weight != 0).float().mean()))
        finally:
            os.chdir(cwd)
            tm.LENS_Trainer.forward = real_forward
    out["meta"] = np.array([places, epochs, seed, int(chain_linear)], np.int64)
    np.savez_compressed(os.path.join(HERE, f"{tag}.npz"), **out)
    print("written", os.path.join(HERE, f"{tag}.npz"))


if __name__ == "__main__":
    main(chain_linear=False, tag="train_example")
    main(chain_linear=True, tag="train_example_chain")
