"""Recall@K as LENS evaluates it (behavioural restatement of `recallAtK`,
lens/src/metrics.py:183-226; written from its behaviour, no text taken from that file).

S and GT are [places, queries].  Queries (columns) without any positive in GT are dropped.  For the
rest, the K most similar places are the last K rows of an ascending argsort of the column; the
query counts as recalled if any of them is a GT positive.  `sort_kind=None` is numpy's default
(unstable) sort, which is what the reference calls; "stable" gives the explicit tie policy of the
GPU top-k path (ties -> larger place index).
"""
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
