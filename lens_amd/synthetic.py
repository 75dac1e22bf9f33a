"""Synthetic inputs for the benchmark configurations: event-count frames and a
large place database with the statistics of the bundled pretrained model.  Everything is drawn
from seeded CPU generators so that the GPU path and the CPU oracle see identical inputs."""
import os
import tempfile

import numpy as np
import torch


def synthetic_frames(n_queries, roi_dim=128, seed=1234):
    """uint8 [Q,roi,roi]: per-pixel rate ~ Gamma(0.5, scale 6) (sparse, mean ~3 like the bundled
    frames), frame = min(Poisson(rate), 255)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    shape = (n_queries, roi_dim, roi_dim)
    rate = torch._standard_gamma(torch.full(shape, 0.5), generator=g) * 6.0
    return torch.poisson(rate, generator=g).clamp_(max=255).to(torch.uint8)


def synthetic_events(frames, seed=4321):
    """Expand count frames into a shuffled (x, y) event list: int16 [n,2] (1-based) + int64 offsets."""
    fr = frames.numpy() if isinstance(frames, torch.Tensor) else np.asarray(frames)
    rng = np.random.default_rng(seed)
    n_frames, H, W = fr.shape
    xs, offs = [], [0]
    for i in range(n_frames):
        r, c = np.nonzero(fr[i])
        k = fr[i][r, c].astype(np.int64)
        xy = np.stack([np.repeat(c, k) + 1, np.repeat(r, k) + 1], axis=1).astype(np.int16)
        rng.shuffle(xy, axis=0)
        xs.append(xy)
        offs.append(offs[-1] + len(xy))
    xy = np.concatenate(xs, axis=0) if xs else np.zeros((0, 2), np.int16)
    return torch.from_numpy(np.ascontiguousarray(xy)), torch.tensor(offs, dtype=torch.int64)


def synthetic_database(w2_pretrained, n_places, seed=7, jitter=0.005):
    """float32 [P,F]: row p = pretrained row (p mod P0) with its features permuted by a per-block
    permutation, plus N(0, jitter) noise; keeps the bundled weight statistics."""
    w = torch.as_tensor(w2_pretrained, dtype=torch.float32)
    P0, F = w.shape
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    n_blocks = (n_places + P0 - 1) // P0
    rows = []
    for b in range(n_blocks):
        perm = torch.randperm(F, generator=g)
        rows.append(w[:, perm])
    db = torch.cat(rows, 0)[:n_places].clone()
    db += torch.randn(db.shape, generator=g) * jitter
    return db.contiguous()


def synthetic_traverses(n_places, roi_dim=128, seed=99):
    """Two traverses of the same n_places synthetic places: every place has a rate image (Gamma(0.5) * 6 per pixel, the
    statistics of synthetic_frames); the reference traverse is one Poisson draw of it, the query traverse another.
    uint8 [P,roi,roi] each; ground truth: query i shows place i."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    rate = torch._standard_gamma(torch.full((n_places, roi_dim, roi_dim), 0.5), generator=g) * 6.0
    ref = torch.poisson(rate, generator=g).clamp_(max=255).to(torch.uint8)
    qry = torch.poisson(rate, generator=g).clamp_(max=255).to(torch.uint8)
    return ref, qry


def train_on_traverse(args, ref_frames, seed=0):
    """Both layers trained on an in-memory reference traverse with the GPU trainer (lens_amd.train_model, the schedule
    of lens/train_model.py:213-252): `args` is the parsed command line (reference_places = number of frames, epochs,
    dims, roi_dim).  Returns (w1 [F,N], w2 [P,F]) float32 on the host.  The run folder the trainer's logger creates
    goes to a temporary directory."""
    from .train_model import LENS_Trainer, training_inputs
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            torch.manual_seed(seed)
            np.random.seed(seed)
            model = LENS_Trainer(args)
            x = training_inputs(ref_frames, args.roi_dim // args.dims)
            labels = torch.arange(ref_frames.shape[0])
            model.train_layer(x, labels, model.feature_layer, prev_layers=[])
            model.train_layer(x, labels, model.output_layer, prev_layers=["feature_layer"])
            for h in list(model.logger.handlers):
                h.close()
                model.logger.removeHandler(h)
        finally:
            os.chdir(cwd)
    return model.feature_layer.w.weight.detach().contiguous(), model.output_layer.w.weight.detach().contiguous()
