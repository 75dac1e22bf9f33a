"""Sum-of-absolute-differences baseline (`--sad`): host-side mirror of lens/src/sad.py:14-68.

All PNGs of the query / reference folders in natural sort order, L1 distance matrix on the GPU
(lens_sad_matrix), the same diagonal sequence filter as LENS, similarity = 1/distance, PR curve and Recall@N.
"""
import os
import re

import numpy as np
import torch
from PIL import Image

from .metrics import recallAtK, createPR


def natural_sort_key(s):
    return [int(t) if t.isdigit() else t.lower() for t in re.split("([0-9]+)", s)]


def load_and_preprocess_images(folder_path):
    files = sorted(os.listdir(folder_path), key=natural_sort_key)
    return np.array([np.array(Image.open(os.path.join(folder_path, f))).flatten() for f in files if f.endswith(".png")])


def run_sad(reference, query, GT, outputdir, sequence_length, logger=None):
    from . import engine as E
    import matplotlib.pyplot as plt
    from .run_model import format_table, RECALL_N
    q = torch.from_numpy(load_and_preprocess_images(query)).cuda()
    r = torch.from_numpy(load_and_preprocess_images(reference)).cuda()
    dist = E.sad_matrix(q, r)                                        # [Q,P], torch.cdist(a, b, 1)
    dist_matrix_seq = E.seqmatch(dist, sequence_length).cpu().numpy().T   # conv2d(eye(L)) / L, transposed
    plt.imshow(dist_matrix_seq)
    plt.colorbar()
    plt.savefig(os.path.join(outputdir, "distance_matrix_SAD.pdf"))
    plt.close()
    with np.errstate(divide="ignore"):
        sim = 1 / dist_matrix_seq
    P, R = createPR(sim, GT, outputdir, datatype="SAD", matching="single", n_thresh=100)
    recallatn = [round(recallAtK(sim, GT, K=n), 2) for n in RECALL_N]
    table = format_table(["N"] + [str(n) for n in RECALL_N], ["Recall"] + recallatn)
    (logger.info if logger else print)(table)
    return {"Precision": P, "Recall": R}, recallatn
