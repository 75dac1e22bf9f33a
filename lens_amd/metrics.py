"""Recall@K as LENS evaluates it (behavioural restatement of `recallAtK`,
lens/src/metrics.py:183-226; written from its behaviour, no text taken from that file).

S and GT are [places, queries].  Queries (columns) without any positive in GT are dropped.  For the
rest, the K most similar places are the last K rows of an ascending argsort of the column; the
query counts as recalled if any of them is a GT positive.  `sort_kind=None` is numpy's default
(unstable) sort, which is what the reference calls; "stable" gives the explicit tie policy of the
GPU top-k path (ties -> larger place index).
"""
import os

import numpy as np


def recallAtK(S_in, GThard, GTsoft=None, K=1, sort_kind=None):
    assert S_in.shape == GThard.shape, "S_in and GThard must have the same shape"
    if GTsoft is not None:
        assert S_in.shape == GTsoft.shape, "S_in and GTsoft must have the same shape"
    assert S_in.ndim == 2, "S_in, GThard and GTsoft must be two-dimensional"
    assert K >= 1, "K must be >=1"
    gt = GThard.astype(bool)
    S = S_in.copy()
    if GTsoft is not None:
        soft = GTsoft.astype(bool)
        S[soft & ~gt] = S.min()          # soft-only matches are ignored during evaluation
    has_match = gt.sum(axis=0) > 0
    S = S[:, has_match]
    gt = gt[:, has_match]
    order = S.argsort(axis=0) if sort_kind is None else S.argsort(axis=0, kind=sort_kind)
    best = order[-K:, :]
    col = np.tile(np.arange(best.shape[1]), [K, 1])
    found = gt[best, col]
    return np.sum(found.sum(axis=0) > 0) / gt.shape[1]


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


def createPR(S_in, GThard, outputdir=None, datatype="LENS", GTsoft=None, matching="multi", n_thresh=100):
    """Precision / recall over n_thresh equally spaced similarity thresholds (behavioural restatement of
    `createPR`, lens/src/metrics.py:21-139).  'single': every query contributes its best match only (first
    maximum of its column); 'multi': every matrix entry counts.  Returns the lists P, R starting with the
    conventional point (P=1, R=0).  With 'single' and an outputdir, the annotated similarity-matrix figure
    `similarity_matrix_<datatype>.pdf` is written like the reference does."""
    assert S_in.shape == GThard.shape, "S_in, GThard and GTsoft must have the same shape"
    assert S_in.ndim == 2, "S_in, GThard and GTsoft must be two-dimensional"
    assert matching in ["single", "multi"], "matching should contain one of the following strings: [single, multi]"
    assert n_thresh > 1, "n_thresh must be >1"
    gt_full = GThard.astype(bool)
    gt = gt_full
    S = S_in.copy()
    if GTsoft is not None:
        S[GTsoft.astype(bool) & ~gt] = S.min()
    best_rows = None
    if matching == "single":
        n_positive = np.count_nonzero(gt.any(0))
        gt = gt[np.argmax(S, axis=0), np.arange(gt.shape[1])]
        best_rows = np.nanargmax(S, axis=0)
        S = np.max(S, axis=0)
    else:
        n_positive = np.count_nonzero(gt)
    R, P = [0], [1]
    hit = None
    for thr in np.linspace(S.max(), S.min(), n_thresh):
        hit = S >= thr
        tp = np.count_nonzero(gt & hit)
        fp = np.count_nonzero(~gt & hit)
        P.append(tp / (tp + fp))
        R.append(tp / n_positive)
    if matching == "single" and outputdir is not None:
        import matplotlib.pyplot as plt
        tp_cols, fp_cols = np.where(gt & hit)[0], np.where(~gt & hit)[0]
        fig, ax = plt.subplots(figsize=(10, 8))
        im = ax.imshow(S_in, cmap="viridis", aspect="auto")
        fig.colorbar(im, ax=ax, label="Similarity Score")
        ax.set_title(f"{datatype} Similarity Matrix with Ground Truth, TP, and FP")
        gy, gx = np.where(GThard)
        ax.scatter(gx, gy, facecolors="white", edgecolors="white", marker=".", label="Ground Truth", linewidths=0.5)
        ax.scatter(tp_cols, best_rows[tp_cols], facecolors="none", edgecolors="green", marker="o",
                   label="True Positives", linewidths=1.0)
        ax.scatter(fp_cols, best_rows[fp_cols], marker="x", color="red", label="False Positives", linewidths=1.0)
        ax.legend(loc="upper right", bbox_to_anchor=(1.0, 1))
        ax.set_xlabel("Query Index")
        ax.set_ylabel("Database Index")
        plt.tight_layout()
        plt.savefig(os.path.join(outputdir, f"similarity_matrix_{datatype}.pdf"), dpi=300)
        plt.close()
    return P, R
