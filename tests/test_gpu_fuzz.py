"""Fixed-seed randomised parity of the HIP path against the CPU oracle, bit for bit (integer-valued spike counts,
membrane states compared as float32 bit patterns, tolerance 0), plus sampled parity at sizes the oracle cannot
run whole: real spike density against a 20 000-place database, and training of a 100 000-row output layer.
"""
import os

import numpy as np
import pytest
import torch

from helpers import load_config, pretrained_example

pytestmark = pytest.mark.gpu

PRED1 = np.nextafter(np.float32(1), np.float32(0))       # 1 - 2^-24: the clip rounds it up to exactly 1.0


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    oracle.lib()
    return oracle


def _diff(a, b):
    """Number of differing elements (NaNs in the same place count as equal); keeps assertion messages short."""
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape:
        return -1
    return int((~((a == b) | ((a != a) & (b != b)))).sum())


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _case(rng, hot):
    k = int(rng.integers(1, 4))
    if hot:      # the instantiation with the hand-scheduled step loop: 65..128 inputs, 129..256 features
        dims = int(rng.integers(9, 12))
        F = int(rng.choice([129, 130, 156, 157, 192, 193, 200, 255, 256]))
        T = int(rng.choice([3, 16, 40, 97]))
        P = int(rng.choice([1, 64, 127, 128, 129, 300, 700]))
    else:
        dims = int(rng.integers(1, 12))
        F = int(rng.choice([1, 3, 17, 64, 65, 130, 156, 157, 200, 208, 209, 312, 313, 400, 624, 640, 900]))
        T = int(rng.choice([1, 2, 7, 16, 33, 100]))
        P = int(rng.choice([1, 2, 63, 64, 65, 127, 128, 129, 255, 256, 257, 500, 1000, 1300]))
    roi, N, Q = dims * k, dims * dims, int(rng.integers(1, 14))
    w1s, w2s = rng.choice([0.05, 0.3, 1.0, 4.0]), rng.choice([0.01, 0.1, 0.5, 2.0])   # 4.0 / 2.0: counts of 2 and more
    w1 = (rng.standard_normal((F, N)) * w1s).astype(np.float32)
    w1[rng.random((F, N)) < rng.choice([0.0, 0.5, 0.9])] = 0
    w2 = (rng.standard_normal((P, F)) * w2s).astype(np.float32)
    if rng.random() < 0.4:
        w2[rng.integers(0, P)] = PRED1                      # membranes that land on 1 - 2^-24
    if rng.random() < 0.15:
        w1[rng.integers(0, F)] = 1.0e7                      # spike counts >= 2^23
    dens = rng.choice([1, 3, 30, 255])
    frames = rng.integers(0, dens + 1, size=(Q, roi, roi)).astype(np.uint8)
    frames[rng.random((Q, roi, roi)) < rng.choice([0.0, 0.7, 0.97])] = 0
    if rng.random() < 0.3:
        frames[rng.integers(0, Q)] = 0
    return dims, roi, T, w1, w2, frames


def _odd_state(rng, n):
    """A carried-in membrane state that is not a clip output (values at and beyond the threshold, below the floor)."""
    v = (rng.standard_normal(n) * 0.7).astype(np.float32)
    v[rng.random(n) < 0.2] = rng.choice(np.array([1.0, 2.5, -3.0, PRED1, 1e-30, -1.0, 0.5], np.float32))
    return v


@pytest.mark.parametrize("hot,n_cases,seed", [(True, 150, 20241), (False, 200, 20242)])
def test_fixed_seed_fuzz_against_the_oracle(O, hot, n_cases, seed):
    from lens_amd.engine import LensEngine, seqmatch_topk
    from lens_amd import _lib
    L_ = _lib.lib()
    rng = np.random.default_rng(seed)
    for case in range(n_cases):
        dims, roi, T, w1, w2, frames = _case(rng, hot)
        (F, N), P = w1.shape, w2.shape[0]
        eng = LensEngine(w1, w2, roi, dims, T, device="cuda")
        U = O.rand_U(T, roi * roi)
        fr = torch.from_numpy(frames).cuda()
        tag = dict(case=case, dims=dims, roi=roi, F=F, P=P, T=T, Q=len(frames))
        ref = O.forward(frames, dims, U, w1, w2, carry=False)
        out = eng.forward_counts(fr, carry=False)
        bad = _diff(out.cpu().numpy(), ref)
        assert bad == 0, ("reset", tag)
        # carried state, starting from membranes that no earlier step could have left behind
        st = O.State(N, F, P)
        st.v2[:], st.v3[:] = _odd_state(rng, F), _odd_state(rng, P)
        v2_0, v3_0 = st.v2.copy(), st.v3.copy()
        eng.reset_state()
        eng.v_feat.copy_(torch.from_numpy(st.v2).cuda()[None])
        eng.v_out.copy_(torch.from_numpy(st.v3).cuda()[None])
        refc = O.forward(frames, dims, U, w1, w2, carry=True, state=st)
        outc = eng.forward_counts(fr, carry=True)
        bad_f = _diff(_bits(eng.v_feat.cpu().numpy()[0]), _bits(st.v2))
        assert bad_f == 0, ("v_feat", tag)
        bad = _diff(outc.cpu().numpy(), refc)
        assert bad == 0, ("carry", tag)
        bad_o = _diff(_bits(eng.v_out.cpu().numpy()[0]), _bits(st.v3))
        assert bad_o == 0, ("v_out", tag)
        if case % 3 == 0:
            # the same run as S identical carried streams, enough of them that the output layer reads its 64-place tiles
            # through L2 (k_output_carry_l2) instead of LDS: every stream must repeat the single stream's result
            S = next((n for n in (130, 260, 520) if L_.lens_output_reads_through_l2(F, P, 1, n) and
                      L_.lens_output_tile_places(F, P, 1, n) == 64), 0)
            if S:
                Qc = len(frames)
                engs = LensEngine(w1, w2, roi, dims, T, device="cuda", max_batch=Qc * S)
                engs.reset_state(n_streams=S)
                engs.v_feat.copy_(torch.from_numpy(v2_0).cuda()[None].expand(S, -1))
                engs.v_out.copy_(torch.from_numpy(v3_0).cuda()[None].expand(S, -1))
                outs = engs.forward_counts(fr.repeat(S, 1, 1), carry=True).view(S, Qc, P)
                assert bool((outs == outc[None]).all() if np.isfinite(refc).all() else
                            _diff(outs.cpu().numpy(), np.broadcast_to(refc, (S, Qc, P))) == 0), ("carry through L2", tag, S)
                assert _diff(_bits(engs.v_out.cpu().numpy()), _bits(np.broadcast_to(st.v3, (S, P)))) == 0, ("v_out through L2", tag)
                del engs
        if case % 3 == 1:
            # ... and as 20 identical streams on the wide LDS tiles that carried streams share 16 at a time (forced: the
            # cost model picks that layout only for many streams on large databases)
            os.environ["LENS_AMD_CARRY_SHARED"] = "1"
            try:
                if L_.lens_output_layout(F, P, 1, 20) == 3:
                    S, Qc = 20, len(frames)
                    engs = LensEngine(w1, w2, roi, dims, T, device="cuda", max_batch=Qc * S)
                    engs.reset_state(n_streams=S)
                    engs.v_feat.copy_(torch.from_numpy(v2_0).cuda()[None].expand(S, -1))
                    engs.v_out.copy_(torch.from_numpy(v3_0).cuda()[None].expand(S, -1))
                    outs = engs.forward_counts(fr.repeat(S, 1, 1), carry=True).view(S, Qc, P)
                    assert bool((outs == outc[None]).all() if np.isfinite(refc).all() else
                                _diff(outs.cpu().numpy(), np.broadcast_to(refc, (S, Qc, P))) == 0), ("carry, shared tiles", tag)
                    assert _diff(_bits(engs.v_out.cpu().numpy()), _bits(np.broadcast_to(st.v3, (S, P)))) == 0, ("v_out, shared tiles", tag)
                    del engs
            finally:
                os.environ.pop("LENS_AMD_CARRY_SHARED", None)
        L = int(rng.choice([1, 2, 3]))
        if len(frames) >= L and P >= L and np.isfinite(refc).all():
            gi, gv = seqmatch_topk(outc, L, 25)
            oi, ov = O.topk(O.seqmatch(refc, L), 25)
            bad_k = _diff(gi.cpu().numpy(), oi) + _diff(gv.cpu().numpy(), ov)
            assert bad_k == 0, ("topk", tag, L)


def test_real_density_sampled_parity_against_20000_places(O):
    """The bundled example's query frames (8 feature spikes per step instead of the synthetic 2.6, so chunks with
    counts of 2 are common) tiled to 512 queries against a 20 000-place database: sampled queries x sampled places
    equal the oracle run on exactly those database rows (output neurons are independent given the streams)."""
    from lens_amd.engine import LensEngine
    from lens_amd.synthetic import synthetic_database
    c, fr, w1, w2p, gt, G = load_config("example")
    P, Q = 20_000, 512
    w2 = synthetic_database(w2p, P).numpy()
    frames = np.ascontiguousarray(np.tile(fr, (6, 1, 1))[:Q])
    frames[300:] = np.roll(frames[300:], 7, axis=2)          # not just repeats
    eng = LensEngine(w1, w2, c["roi"], c["dims"], 250, device="cuda", max_batch=Q)
    out = eng.forward_counts(torch.from_numpy(frames).cuda(), carry=False)
    nw = eng.nwords[:Q]
    w0 = eng.events[:Q].view(Q, -1, 2)[:, :int(nw.max().item()) // 2, 0]
    valid = torch.arange(w0.shape[1], device="cuda")[None, :] < (nw[:, None] // 2)
    assert ((w0 < 0) & valid).sum().item() > 1000                          # chunks flagged "some count is not 1"
    rng = np.random.default_rng(4)
    places = np.unique(np.concatenate([[0, 127, 128, P - 129, P - 1], rng.integers(0, P, 250)]))
    queries = np.unique(np.concatenate([[0, 15, 16, 99, 100, 299, 300, Q - 1], rng.integers(0, Q, 56)]))
    ref = O.forward(frames[queries], c["dims"], O.rand_U(250, c["roi"] ** 2), w1, np.ascontiguousarray(w2[places]),
                    carry=False, skip_zero=True)
    got = out[torch.from_numpy(queries).cuda()][:, torch.from_numpy(places).cuda()].cpu().numpy()
    bad = _diff(got, ref)
    assert bad == 0 and ref.sum() > 1000


def test_training_100000_rows_sampled_against_the_oracle(O):
    """Benchmark training size: 300 spike-forcing steps of a [100 000 x 200] output layer on the GPU; weight rows
    evolve independently given the sample sequence, so 64 scattered rows (and the forced ones among them) must equal
    the oracle run on exactly those rows."""
    from lens_amd.engine import train_layer, train_in_pad, pad_table
    rng = np.random.default_rng(77)
    n_out, n_in, S, T = 100_000, 200, 64, 300
    w0 = (rng.standard_normal((n_out, n_in)) * 0.3).astype(np.float32)
    w0[rng.random((n_out, n_in)) < 0.3] = 0
    exc, inh = (w0 > 0).astype(np.uint8), (w0 < 0).astype(np.uint8)
    thr0 = (rng.random(n_out) * 0.3).astype(np.float32)
    fr = np.full(n_out, 0.5, np.float32)
    x = (rng.random((S, n_in)) * (rng.random((S, n_in)) < 0.4)).astype(np.float32)
    pfr = (0.3 + rng.random(n_in) * 0.4).astype(np.float32)
    order = rng.integers(0, S, T).astype(np.int32)
    rows = np.unique(np.concatenate([[0, 63, 64, n_out - 1], rng.integers(0, n_out, 60)]))
    fi = rng.integers(0, n_out, T).astype(np.int32)
    fi[::3] = rng.choice(rows, size=len(fi[::3]))            # a third of the steps force one of the sampled rows
    eip, est = O.anneal_schedule(0.02, 0.01, T)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    ld = train_in_pad(n_in)
    w, thr = dev(w0), dev(thr0)
    train_layer(w, thr, dev(exc), dev(inh), dev(fr), True, pad_table(dev(x), ld), dev(order), dev(eip), dev(est),
                force_idx=dev(fi), mpre_table=pad_table((torch.from_numpy(x) / torch.from_numpy(pfr)).cuda(), ld))
    # oracle on the sampled rows plus one spare row that takes the forcing of every row outside the sample
    sub = np.concatenate([rows, [1]])
    pos = {int(r): i for i, r in enumerate(rows)}
    fi_sub = np.array([pos.get(int(f), len(rows)) for f in fi], np.int32)
    rw, rthr = O.train_layer(w0[sub], thr0[sub], exc[sub], inh[sub], fr[sub], x, order, eip, est,
                             force_idx=fi_sub, prev_fire_rate=pfr)
    gw, gthr = w.cpu().numpy(), thr.cpu().numpy()
    bad_w, bad_t = _diff(gw[rows], rw[:len(rows)]), _diff(gthr[rows], rthr[:len(rows)])
    assert bad_w == 0 and bad_t == 0
    changed = _diff(gw[rows], w0[rows])
    assert changed > 0
