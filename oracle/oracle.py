"""
Python face of the CPU oracle (TEST INFRASTRUCTURE ONLY -- see oracle/lens_oracle.c header).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The numeric core is the C restatement in lens_oracle.c (ctypes); the small host-side pieces
that the reference itself does in numpy/scipy (GT tolerance, Recall@K) are restated here in
numpy, each citing the reference lines it follows.

Parity status: "parity unpinned" at the sinabs IAF boundary (lens_oracle.c header); everything
else is pinned by tests/golden (vectors generated from the reference's own modules).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    """Compile liblens_oracle.so with gcc (idempotent)."""
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liblens_oracle.so")
        if not os.path.exists(path):
            build()
        L = ctypes.CDLL(path)
        c_p = ctypes.c_void_p
        L.lens_oracle_accumulate_events.restype = ctypes.c_int64
        L.lens_oracle_accumulate_events.argtypes = [c_p, c_p, c_p, ctypes.c_int, ctypes.c_int, c_p]
        L.lens_oracle_iaf_step.restype = None
        L.lens_oracle_iaf_step.argtypes = [c_p, c_p, c_p, ctypes.c_int]
        L.lens_oracle_linear.restype = None
        L.lens_oracle_linear.argtypes = [c_p, c_p, c_p, ctypes.c_int, ctypes.c_int]
        L.lens_oracle_forward.restype = None
        L.lens_oracle_forward.argtypes = [c_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                          c_p, ctypes.c_int, c_p, ctypes.c_int, c_p, ctypes.c_int,
                                          c_p, c_p, c_p, ctypes.c_int, ctypes.c_int, c_p, c_p]
        L.lens_oracle_seqmatch.restype = None
        L.lens_oracle_seqmatch.argtypes = [c_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_p]
        L.lens_oracle_topk.restype = None
        L.lens_oracle_topk.argtypes = [c_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_p, c_p]
        L.lens_oracle_layer_forward_clamped.restype = None
        L.lens_oracle_layer_forward_clamped.argtypes = [c_p, c_p, c_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_p]
        L.lens_oracle_train_layer.restype = None
        L.lens_oracle_train_layer.argtypes = [c_p] * 12 + [ctypes.c_int] * 4
        L.lens_oracle_sad.restype = None
        L.lens_oracle_sad.argtypes = [c_p, ctypes.c_int, c_p, ctypes.c_int, ctypes.c_int, c_p]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


# --------------------------------------------------------------------------------------------
def rand_U(T, HW):
    """lens/src/dataset.py:120-121: torch.manual_seed(50); torch.rand(time_window, *image.shape).
    The seed is reset for every item, so this tensor is a constant of (T, H*W)."""
    import torch
    g = torch.Generator(device="cpu")
    g.manual_seed(50)
    return torch.rand(T, HW, generator=g).numpy()


def accumulate_events(x, y, frame_off, roi):
    """lens/collect_data.py:193-199. Raises IndexError like torch indexing would."""
    x = np.ascontiguousarray(x, dtype=np.int16)
    y = np.ascontiguousarray(y, dtype=np.int16)
    frame_off = np.ascontiguousarray(frame_off, dtype=np.int64)
    n = len(frame_off) - 1
    frames = np.zeros((n, roi, roi), dtype=np.uint8)
    bad = lib().lens_oracle_accumulate_events(_p(x), _p(y), _p(frame_off), n, roi, _p(frames))
    if bad >= 0:
        raise IndexError(f"event {bad} (x={x[bad]}, y={y[bad]}) is out of bounds for a {roi}x{roi} frame")
    return frames


class State:
    """Membrane potentials of the three IAF layers (persist across queries in carry mode)."""

    def __init__(self, N, F, P):
        self.v1 = np.zeros(N, np.float32)
        self.v2 = np.zeros(F, np.float32)
        self.v3 = np.zeros(P, np.float32)


def forward(frames, dims, U, w1, w2, carry=True, state=None, skip_zero=False, want_feat=False):
    """Query loop of LENS.evaluate (lens/run_model.py:229-246). frames uint8 [Q,H,W];
    U float32 [T,H*W]; w1 [F,N]; w2 [P,F]. Returns out float32 [Q,P] (and feat spikes [Q,T,F])."""
    frames = np.ascontiguousarray(frames, dtype=np.uint8)
    Q, H, W = frames.shape
    U = np.ascontiguousarray(U, dtype=np.float32)
    T = U.shape[0]
    assert U.shape[1] == H * W
    w1 = np.ascontiguousarray(w1, dtype=np.float32)
    w2 = np.ascontiguousarray(w2, dtype=np.float32)
    F, N = w1.shape
    P = w2.shape[0]
    assert N == dims * dims and w2.shape[1] == F
    if state is None:
        state = State(N, F, P)
    out = np.zeros((Q, P), np.float32)
    feat = np.zeros((Q, T, F), np.float32) if want_feat else None
    lib().lens_oracle_forward(_p(frames), Q, H, W, dims, _p(U), T, _p(w1), F, _p(w2), P,
                              _p(state.v1), _p(state.v2), _p(state.v3), int(bool(carry)), int(bool(skip_zero)),
                              _p(out), _p(feat) if want_feat else None)
    return (out, feat) if want_feat else out


def seqmatch(out, L):
    """lens/run_model.py:248-254. out [Q,P] -> S float32 [P-L+1, Q-L+1]; L == 0 -> out unchanged."""
    if L == 0:
        return np.asarray(out)
    o = np.ascontiguousarray(out, dtype=np.float32)
    Q, P = o.shape
    S = np.zeros((P - L + 1, Q - L + 1), np.float32)
    lib().lens_oracle_seqmatch(_p(o), Q, P, L, _p(S))
    return S


def topk(S, K):
    """Top-K places per query column of S [P',Q'] under (score desc, index desc). -> idx, val [Q',K]."""
    S = np.ascontiguousarray(S, dtype=np.float32)
    Ps, Qs = S.shape
    idx = np.zeros((Qs, K), np.int32)
    val = np.zeros((Qs, K), np.float32)
    lib().lens_oracle_topk(_p(S), Ps, Qs, K, _p(idx), _p(val))
    return idx, val


def gt_tolerance(GT, L, g):
    """lens/run_model.py:268-294: slice GT[L-2:-1, L-2:-1] (only if L != 0), binary dilation with a
    (2g+1)x(2g+1) square, cast int, transpose."""
    from scipy.ndimage import binary_dilation
    GT = np.asarray(GT)
    if L != 0:
        GT = GT[L - 2:-1, L - 2:-1]
    se = np.ones((2 * g + 1, 2 * g + 1), dtype=int)
    return binary_dilation(GT, structure=se).astype(int).T


def recall_at_k(S, GT, K, kind=None):
    """lens/src/metrics.py:183-226 (GTsoft=None branch): drop query columns without any positive,
    take the K largest similarities per column (argsort ascending, last K rows) and count a hit if
    any of them is a positive. `kind=None` is numpy's default sort, as the reference uses."""
    S = np.asarray(S)
    G = np.asarray(GT).astype(bool)
    assert S.shape == G.shape and S.ndim == 2 and K >= 1
    keep = G.sum(0) > 0
    S = S[:, keep]
    G = G[:, keep]
    order = S.argsort(0) if kind is None else S.argsort(0, kind=kind)
    top = order[-K:, :]
    cols = np.tile(np.arange(top.shape[1]), [K, 1])
    hits = G[top, cols]
    return np.sum(hits.sum(0) > 0) / G.shape[1]


def recall_from_topk(idx, GT, K):
    """Recall@K from top-k index lists idx [Q',>=K] (policy: score desc, index desc) and GT [P',Q']."""
    G = np.asarray(GT).astype(bool)
    keep = np.nonzero(G.sum(0) > 0)[0]
    hit = 0
    for i in keep:
        cand = idx[i, :K]
        cand = cand[cand >= 0]
        hit += bool(G[cand, i].any())
    return hit / len(keep)


def load_frames(csv_path, img_dir, skip=1, max_samples=None):
    """lens/src/dataset.py:74-84,99-104: CSV (first column = file name), every `skip`-th row,
    first `max_samples`; 8-bit grayscale PNG -> uint8 [H,W]."""
    import csv
    from PIL import Image
    with open(csv_path) as fh:
        rows = list(csv.reader(fh))[1:]
    rows = rows[::skip]
    if max_samples is not None:
        rows = rows[:max_samples]
    frames = []
    for r in rows:
        p = os.path.join(img_dir, r[0])
        if not os.path.exists(p):
            raise FileNotFoundError(p)
        im = np.array(Image.open(p))
        assert im.ndim == 2 and im.dtype == np.uint8
        frames.append(im)
    return np.stack(frames)


# --------------------------------------------------------------------------------------------
# training path (pinned by tests/golden/train_example.npz, made with the reference's own code)
# --------------------------------------------------------------------------------------------
def layer_forward_clamped(x_table, w, thr):
    """clamp(w(x) - thr, 0, 1) for every sample (lens/train_model.py:163-168)."""
    x = np.ascontiguousarray(x_table, np.float32); w = np.ascontiguousarray(w, np.float32)
    thr = np.ascontiguousarray(thr, np.float32).reshape(-1)
    S, n_in = x.shape; n_out = w.shape[0]
    y = np.zeros((S, n_out), np.float32)
    lib().lens_oracle_layer_forward_clamped(_p(x), _p(w), _p(thr), S, n_in, n_out, _p(y))
    return y


def anneal_schedule(eta_ip0, eta_stdp0, T):
    """Learning rates in effect at each of the T steps of train_layer (lens/train_model.py:150-152,
    120-128, 179-181): the constructor values at step 0, then every 10 steps
    eta = init * ((T - mod)/T)**2 with init_itp = 2*eta_stdp0 (torch.mul of a float32 tensor with a
    Python float)."""
    import torch
    eip, est = torch.tensor(eta_ip0, dtype=torch.float32), torch.tensor(eta_stdp0, dtype=torch.float32)
    init_itp, init_stdp = est * 2, est.clone()
    out_ip, out_st = np.zeros(T, np.float32), np.zeros(T, np.float32)
    for mod in range(T):
        out_ip[mod], out_st[mod] = float(eip), float(est)
        if mod % 10 == 0:
            pt = pow(float(T - mod) / T, 2)
            eip, est = torch.mul(init_itp, pt), torch.mul(init_stdp, pt)
    return out_ip, out_st


def train_layer(w, thr, exc, inh, fire_rate, x_table, order, eta_ip, eta_stdp, force_idx=None, prev_fire_rate=None):
    """T = len(order) training steps in place on (w, thr). Returns (w, thr)."""
    w = np.ascontiguousarray(w, np.float32).copy(); thr = np.ascontiguousarray(thr, np.float32).reshape(-1).copy()
    exc = np.ascontiguousarray(exc, np.uint8); inh = np.ascontiguousarray(inh, np.uint8)
    fr = np.ascontiguousarray(fire_rate, np.float32).reshape(-1)
    x = np.ascontiguousarray(x_table, np.float32)
    order = np.ascontiguousarray(order, np.int32).reshape(-1)
    eip = np.ascontiguousarray(eta_ip, np.float32); est = np.ascontiguousarray(eta_stdp, np.float32)
    n_out, n_in = w.shape
    spk = force_idx is not None
    fi = np.ascontiguousarray(force_idx, np.int32).reshape(-1) if spk else None
    pfr = np.ascontiguousarray(prev_fire_rate, np.float32).reshape(-1) if spk else None
    have_rate = int(fr[0] > 0)
    lib().lens_oracle_train_layer(_p(w), _p(thr), _p(exc), _p(inh), _p(fr), have_rate, _p(x), _p(order),
                                  _p(fi) if spk else None, _p(pfr) if spk else None, _p(eip), _p(est),
                                  n_out, n_in, len(order), int(spk))
    return w, thr


def sad_matrix(query, reference):
    """lens/src/sad.py:34-38: L1 distance between every query and reference frame. uint8 [Q,...], [P,...] -> [Q,P]."""
    q = np.ascontiguousarray(query, np.uint8).reshape(len(query), -1)
    r = np.ascontiguousarray(reference, np.uint8).reshape(len(reference), -1)
    d = np.zeros((len(q), len(r)), np.float32)
    lib().lens_oracle_sad(_p(q), len(q), _p(r), len(r), q.shape[1], _p(d))
    return d
