"""CPU tests: the oracle against the golden vectors produced by the reference's own modules
(tests/golden/make_golden.py), plus its edge-case behaviour.  No GPU needed."""
import hashlib

import numpy as np
import pytest

from helpers import GOLDEN, load_config
from oracle import oracle as O


@pytest.fixture(scope="module", autouse=True)
def _built():
    O.build()
    O.lib()


@pytest.mark.parametrize("tag,nq", [("example", None), ("sunset", None)])
@pytest.mark.parametrize("mode", ["carry", "reset"])
def test_forward_matches_reference_plumbing(tag, nq, mode):
    """out[Q,P] of the restatement == out of lens/run_model.py driven with torch's own Linear/Conv."""
    c, frames, w1, w2, gt, G = load_config(tag)
    U = O.rand_U(250, c["roi"] ** 2)
    out, feat = O.forward(frames, c["dims"], U, w1, w2, carry=(mode == "carry"), want_feat=True)
    assert np.array_equal(out.astype(np.int16), G[f"out_{mode}"])
    if mode == "carry":
        assert np.array_equal(feat[G["feat_queries"]].astype(np.int8), G["feat_spk"])
        assert int(G["iaf0_is_binary"][0]) == 1
    # event-driven evaluation of the same chains is bit-identical
    out2 = O.forward(frames[:40], c["dims"], U, w1, w2, carry=(mode == "carry"), skip_zero=True)
    assert np.array_equal(out2, out[:40])


@pytest.mark.parametrize("tag", ["example", "sunset"])
def test_matching_matches_reference(tag):
    """Sequence matcher, GT tolerance and Recall@N against what run_model.py / metrics.py produced."""
    c, frames, w1, w2, gt, G = load_config(tag)
    for mode in ("carry", "reset"):
        S = O.seqmatch(G[f"out_{mode}"].astype(np.float32), 2)
        assert S.dtype == np.float32 and np.array_equal(S, G[f"S_{mode}"])
        GT = O.gt_tolerance(gt, 2, 3)
        assert np.array_equal(GT, G["GTtol"])
        R = [O.recall_at_k(S, GT, K) for K in (1, 5, 10, 15, 20, 25)]
        assert np.allclose(R, G[f"recall_{mode}"], rtol=0, atol=0)


def test_rand_tensor_is_the_reference_draw():
    """A private generator seeded 50 == the global `torch.manual_seed(50); torch.rand(...)` of dataset.py:120-121."""
    R = np.load(f"{GOLDEN}/raster.npz")
    for T, HW in ((250, 6400), (250, 16384), (250, 49)):
        U = O.rand_U(T, HW)
        assert hashlib.sha256(U.tobytes()).digest() == R[f"U_{T}_{HW}_sha256"].tobytes()
        assert np.array_equal(U[:2, :8], R[f"U_{T}_{HW}_head"])


@pytest.mark.parametrize("tag,k,dims", [("example", 8, 10), ("sunset", 1, 7)])
def test_raster_matches_dataset_py(tag, k, dims):
    """(U < px/255) at the centre pixels == the spikes CustomImageDataset.__getitem__ produced."""
    c, frames, *_ = load_config(tag)
    R = np.load(f"{GOLDEN}/raster.npz")
    shape = tuple(R[f"{tag}_sel_shape"])
    want = np.unpackbits(R[f"{tag}_sel_raster"], axis=-1)[..., :shape[-1]]
    roi = c["roi"]
    U = O.rand_U(250, roi * roi)
    c0 = (k // 2 - 1) % k
    pix = (np.arange(dims) * k + c0)
    sel = (pix[:, None] * roi + pix[None, :]).reshape(-1)
    for i in range(3):
        img = frames[i].reshape(-1).astype(np.float32) / np.float32(255)
        got = (U[:, sel] < img[sel][None, :]).astype(np.uint8)
        assert np.array_equal(got, want[i])


def test_iaf_rule_edge_cases():
    """The literal sinabs update, including the relu(v+1)-1 rounding and multi-spike behaviour."""
    v = np.array([0.0, 0.0, 0.0, -0.5, 0.999999940395], np.float32)
    cur = np.array([2.75, -3.0, 0.3, -0.6, 0.0], np.float32)
    spk = np.zeros(5, np.float32)
    O.lib().lens_oracle_iaf_step(v.ctypes.data, cur.ctypes.data, spk.ctypes.data, 5)
    assert spk.tolist() == [2.0, 0.0, 0.0, 0.0, 0.0]
    assert v[0] == np.float32(0.75) and v[1] == np.float32(-1.0)           # multi-spike; clip at v_min
    assert v[2] == np.float32(np.float32(0.3) + np.float32(1)) - np.float32(1)   # rounding by the clip
    assert v[3] == np.float32(-1.0)
    # v = 1 - 2^-24 rounds up to 1.0 through (v + 1) - 1 and fires on the NEXT step without input
    assert v[4] == np.float32(1.0)
    O.lib().lens_oracle_iaf_step(v.ctypes.data, np.zeros(5, np.float32).ctypes.data, spk.ctypes.data, 5)
    assert spk[4] == 1.0 and v[4] == 0.0


def test_linear_is_an_ascending_fma_chain():
    rng = np.random.default_rng(0)
    x = rng.integers(0, 3, 37).astype(np.float32)
    w = rng.standard_normal((5, 37)).astype(np.float32)
    y = np.zeros(5, np.float32)
    O.lib().lens_oracle_linear(x.ctypes.data, w.ctypes.data, y.ctypes.data, 37, 5)
    ref = np.zeros(5, np.float32)
    for o in range(5):
        acc = np.float32(0)
        for i in range(37):   # fma: exact product and sum in float64 (exact for these magnitudes), one rounding
            acc = np.float32(np.float64(x[i]) * np.float64(w[o, i]) + np.float64(acc))
        ref[o] = acc
    assert np.array_equal(y, ref)


def test_topk_policy_is_stable_argsort():
    rng = np.random.default_rng(4)
    S = (rng.integers(0, 4, size=(50, 7)) / 2).astype(np.float32)   # many ties
    for K in (1, 5, 25):
        idx, val = O.topk(S, K)
        order = S.argsort(0, kind="stable")[-K:][::-1].T           # [Q,K], best first
        assert np.array_equal(idx, order)
        assert np.array_equal(val, np.take_along_axis(S.T, order, 1))
    idx, val = O.topk(S[:3], 5)
    assert (idx[:, 3:] == -1).all() and np.isneginf(val[:, 3:]).all()


def test_accumulate_events_follows_torch_indexing():
    import torch
    rng = np.random.default_rng(1)
    roi, n = 12, 4000
    x = rng.integers(0, roi + 1, n).astype(np.int16)     # 0 wraps to the last column like frame[.., -1]
    y = rng.integers(0, roi + 1, n).astype(np.int16)
    off = np.array([0, 1500, 1500, n], np.int64)
    got = O.accumulate_events(x, y, off, roi)
    for f in range(3):
        frame = torch.zeros((roi, roi), dtype=int)
        for e in range(off[f], off[f + 1]):
            frame[int(y[e]) - 1, int(x[e]) - 1] += 1              # lens/collect_data.py:197
        assert np.array_equal(got[f], frame.numpy().astype(np.uint8))
    assert got[1].sum() == 0
    with pytest.raises(IndexError):
        O.accumulate_events(np.array([roi + 1], np.int16), np.array([1], np.int16), np.array([0, 1]), roi)
    with pytest.raises(IndexError):
        torch.zeros((roi, roi), dtype=int)[0, roi]


# ---------------------------------------------------------------------------------------------
# training path: goldens come from the reference's own lens/train_model.py + blitnet.py
# ---------------------------------------------------------------------------------------------
def _train_replay(G, upto=None):
    """Replay the golden run with the oracle; yields (layer, step, w, thr) at the golden snapshot steps."""
    X, labels = G["x_table"], G["labels"]
    table = X
    for lname, spk in (("feature_layer", False), ("output_layer", True)):
        order = G[f"{lname}_order"].reshape(-1)
        T = len(order)
        eip, est = O.anneal_schedule(float(G[f"{lname}_eta0"][0]), float(G[f"{lname}_eta0"][1]), T)
        assert np.array_equal(np.stack([eip, est], 1), G[f"{lname}_eta_used"])     # annealing schedule restated exactly
        w, thr = G[f"{lname}_w0"], G[f"{lname}_thr0"]
        done = 0
        for k in (1, 2, 10, 11, 37, T):
            kw = dict(force_idx=labels[order[done:k]], prev_fire_rate=G["feature_layer_fire_rate"]) if spk else {}
            w, thr = O.train_layer(w, thr, G[f"{lname}_exc"], G[f"{lname}_inh"], G[f"{lname}_fire_rate"], table,
                                   order[done:k], eip[done:k], est[done:k], **kw)
            done = k
            yield lname, ("final" if k == T else f"step{k}"), w, thr
        if not spk:
            table = O.layer_forward_clamped(X, w, thr)


def test_training_restatement_is_bit_exact_with_pinned_gemv_order():
    """Reference STDP / clamps / ITP / homeostasis / annealing (its own torch code) with only the Linear's
    summation order pinned to the ascending fmaf chain: the oracle must agree bit for bit at every snapshot."""
    G = np.load(f"{GOLDEN}/train_example_chain.npz")
    n = 0
    for lname, tag, w, thr in _train_replay(G):
        assert np.array_equal(w, G[f"{lname}_w_{tag}"]), (lname, tag)
        assert np.array_equal(thr, G[f"{lname}_thr_{tag}"].reshape(-1)), (lname, tag)
        n += 1
    assert n == 12


def test_training_matches_unmodified_reference_within_gemv_order_noise():
    """Against the completely unmodified reference run (torch's MKL GEMV order, which is unspecified) the
    weights agree to 1e-6 absolute (float32 scores tolerance of the north star is 1e-5); thresholds exactly."""
    G = np.load(f"{GOLDEN}/train_example.npz")
    for lname, tag, w, thr in _train_replay(G):
        assert np.allclose(w, G[f"{lname}_w_{tag}"], rtol=0, atol=1e-6), (lname, tag)
        assert np.array_equal(thr, G[f"{lname}_thr_{tag}"].reshape(-1))
        assert np.array_equal(w == 0, G[f"{lname}_w_{tag}"] == 0)          # sparsity pattern is preserved
