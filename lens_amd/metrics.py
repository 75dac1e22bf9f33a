"""Host-side mirror of the two evaluation entry points of LENS, `recallAtK` and `createPR`
(lens/src/metrics.py:183-226 and :21-139).

What is kept from the reference on purpose, because callers and the parity tests see it: the function and
argument names, the assertion messages, the returned values (Recall@K as a float; the P and R lists starting
with the conventional point P=1, R=0) and the name of the figure file.  What is this build's own: the
arithmetic is restated in vectorised numpy (the precision/recall sweep counts hits with a sort and a binary
search instead of one pass over the matrix per threshold) and the figure is drawn by `_draw_matches` below.

S and GT are [places, queries].  Recall@K: queries (columns) without any positive in GT are dropped; for the
rest, the K most similar places are the last K rows of an ascending argsort of the column, and the query counts
as recalled if any of them is a GT positive.  `sort_kind=None` is numpy's default (unstable) sort, which is what
the reference calls -- with integer spike counts ties are the norm, so the tie order is part of the result;
"stable" gives the explicit tie policy of the GPU top-k path (ties -> larger place index).
"""
import os

import numpy as np


def _checked(S_in, GThard, GTsoft, what):
    assert S_in.shape == GThard.shape, f"{what} must have the same shape"
    if GTsoft is not None:
        assert S_in.shape == GTsoft.shape, "S_in and GTsoft must have the same shape"
    gt = GThard.astype(bool)
    S = S_in.copy()
    if GTsoft is not None:
        S[GTsoft.astype(bool) & ~gt] = S.min()          # soft-only matches are ignored during evaluation
    return S, gt


def recallAtK(S_in, GThard, GTsoft=None, K=1, sort_kind=None):
    S, gt = _checked(S_in, GThard, GTsoft, "S_in and GThard")
    assert S_in.ndim == 2, "S_in, GThard and GTsoft must be two-dimensional"
    assert K >= 1, "K must be >=1"
    keep = gt.any(axis=0)
    S, gt = S[:, keep], gt[:, keep]
    order = S.argsort(axis=0) if sort_kind is None else S.argsort(axis=0, kind=sort_kind)
    recalled = np.take_along_axis(gt, order[-K:, :], axis=0).any(axis=0)
    return np.count_nonzero(recalled) / gt.shape[1]


def recall_from_topk(topk_idx, GThard, K):
    """Recall@K from device top-k lists: topk_idx int [queries, >=K] best first (-1 = none)."""
    gt = np.asarray(GThard).astype(bool)
    cols = np.nonzero(gt.sum(axis=0) > 0)[0]
    hits = 0
    for i in cols:
        cand = topk_idx[i, :K]
        cand = cand[cand >= 0]
        hits += bool(gt[cand, i].any())
    return hits / len(cols)


def _count_at_least(values, thresholds):
    """For every threshold, how many of `values` are >= it."""
    v = np.sort(np.asarray(values).ravel())
    return v.size - np.searchsorted(v, thresholds, side="left")


def _draw_matches(S_in, GThard, best_rows, is_tp, is_fp, path, datatype):
    """The similarity matrix with the ground truth and, per query, its best match marked as a true or a false
    positive at the lowest threshold."""
    import matplotlib.pyplot as plt
    fig, ax = plt.subplots(figsize=(10, 8))
    fig.colorbar(ax.imshow(S_in, cmap="viridis", aspect="auto"), ax=ax, label="similarity")
    rows, cols = np.nonzero(GThard)
    ax.scatter(cols, rows, s=4, c="white", marker=".", label="ground truth")
    q = np.arange(S_in.shape[1])
    ax.scatter(q[is_tp], best_rows[is_tp], s=30, facecolors="none", edgecolors="green", label="best match: true positive")
    ax.scatter(q[is_fp], best_rows[is_fp], s=30, c="red", marker="x", label="best match: false positive")
    ax.set(title=f"{datatype}: best match per query", xlabel="query", ylabel="reference place")
    ax.legend(loc="upper right")
    fig.tight_layout()
    fig.savefig(path, dpi=300)
    plt.close(fig)


def createPR(S_in, GThard, outputdir=None, datatype="LENS", GTsoft=None, matching="multi", n_thresh=100):
    """Precision / recall at n_thresh equally spaced similarity thresholds from max(S) down to min(S).
    'single': every query contributes its best match only (the first maximum of its column); 'multi': every
    matrix entry counts.  Returns the lists P, R, each starting with the conventional point (P=1, R=0).  With
    'single' and an outputdir the figure `similarity_matrix_<datatype>.pdf` is written there."""
    S, gt = _checked(S_in, GThard, GTsoft, "S_in, GThard and GTsoft")
    assert S_in.ndim == 2, "S_in, GThard and GTsoft must be two-dimensional"
    assert matching in ["single", "multi"], "matching should contain one of the following strings: [single, multi]"
    assert n_thresh > 1, "n_thresh must be >1"
    best_rows = None
    if matching == "single":
        n_positive = np.count_nonzero(gt.any(axis=0))
        best_rows = np.argmax(S, axis=0)
        q = np.arange(S.shape[1])
        gt, S = gt[best_rows, q], S[best_rows, q]
    else:
        n_positive = np.count_nonzero(gt)
    thresholds = np.linspace(S.max(), S.min(), n_thresh)
    tp = _count_at_least(S[gt], thresholds)
    fp = _count_at_least(S[~gt], thresholds)
    P = [1] + [int(a) / int(a + b) for a, b in zip(tp, fp)]       # the highest threshold always admits max(S)
    R = [0] + [int(a) / n_positive for a in tp]
    if matching == "single" and outputdir is not None:
        admitted = S >= thresholds[-1]
        _draw_matches(S_in, GThard, best_rows, gt & admitted, ~gt & admitted,
                      os.path.join(outputdir, f"similarity_matrix_{datatype}.pdf"), datatype)
    return P, R
