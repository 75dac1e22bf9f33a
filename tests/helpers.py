"""Shared test helpers: bundled configurations and oracle access (oracle = checker only)."""
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
DATA = os.path.join(ROOT, "lens", "dataset")
MODELS = os.path.join(ROOT, "lens", "models")

CONFIGS = {
    "example": dict(query_csv="example-query.csv", query_dir="example/davis128/example-query",
                    model="example-reference_LENS_IN100_FN200_DB100.pth", dims=10, roi=80, Q=100, P=100,
                    gt="example/davis128/example-reference_example-query_GT.npy"),
    "sunset": dict(query_csv="sunset1.csv", query_dir="brisevent/davis346/sunset1",
                   model="sunset2_LENS_IN49_FN63_DB641.pth", dims=7, roi=7, Q=724, P=641,
                   gt="brisevent/davis346/sunset2_sunset1_GT.npy"),
}


def load_config(tag):
    from oracle import oracle as O
    c = CONFIGS[tag]
    frames = O.load_frames(os.path.join(DATA, c["query_csv"]), os.path.join(DATA, c["query_dir"]), 1, c["Q"])
    sd = torch.load(os.path.join(MODELS, c["model"]), map_location="cpu", weights_only=True)
    w1 = sd["feature_layer.w.weight"].numpy()
    w2 = sd["output_layer.w.weight"].numpy()
    gt = np.load(os.path.join(DATA, c["gt"]))
    golden = np.load(os.path.join(GOLDEN, f"{tag}.npz"))
    return c, frames, w1, w2, gt, golden


def pretrained_example():
    sd = torch.load(os.path.join(MODELS, CONFIGS["example"]["model"]), map_location="cpu", weights_only=True)
    return sd["feature_layer.w.weight"].numpy(), sd["output_layer.w.weight"].numpy()
