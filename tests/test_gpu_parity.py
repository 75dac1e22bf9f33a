"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and the golden vectors.

Bar: bit-exact spike counts / similarity matrices / top-k indices (integer-valued work); the only
floating-point outputs, S = n/L, must be equal to the oracle's float32 (tolerance 0, stated here).
"""
import numpy as np
import pytest
import torch

from helpers import load_config, pretrained_example

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    oracle.lib()
    return oracle


def _engine(w1, w2, roi, dims, T=250, **kw):
    from lens_amd.engine import LensEngine
    return LensEngine(w1, w2, roi, dims, T, device="cuda", **kw)


@pytest.mark.parametrize("tag", ["example", "sunset"])
@pytest.mark.parametrize("mode", ["carry", "reset"])
def test_bundled_counts_match_golden(tag, mode):
    """Config 1 inputs: out[Q,P] must equal the vectors produced by the reference's own plumbing."""
    c, frames, w1, w2, gt, G = load_config(tag)
    eng = _engine(w1, w2, c["roi"], c["dims"])
    out = eng.forward_counts(torch.from_numpy(frames).cuda(), carry=(mode == "carry")).cpu().numpy()
    assert np.array_equal(out, np.round(out))
    assert np.array_equal(out.astype(np.int16), G[f"out_{mode}"])


@pytest.mark.parametrize("tag", ["example", "sunset"])
def test_feature_spikes_match_golden(tag):
    c, frames, w1, w2, gt, G = load_config(tag)
    eng = _engine(w1, w2, c["roi"], c["dims"])
    spk = eng.feature_events(torch.from_numpy(frames).cuda(), carry=True, want_spikes=True).cpu().numpy()
    fq = G["feat_queries"]
    assert np.array_equal(spk[fq].astype(np.int8), G["feat_spk"])


def test_carry_state_persists_across_calls():
    """Two half-batches with carried state == one full batch (state lives in the engine)."""
    c, frames, w1, w2, gt, G = load_config("example")
    eng = _engine(w1, w2, c["roi"], c["dims"])
    f = torch.from_numpy(frames).cuda()
    a = eng.forward_counts(f[:37], carry=True)
    b = eng.forward_counts(f[37:], carry=True)
    out = torch.cat([a, b]).cpu().numpy()
    assert np.array_equal(out.astype(np.int16), G["out_carry"])
    eng.reset_state()
    again = eng.forward_counts(f, carry=True).cpu().numpy()
    assert np.array_equal(again.astype(np.int16), G["out_carry"])


@pytest.mark.parametrize("chunk", [7, 32, 1000])
def test_pipelined_carried_stream_is_identical(chunk):
    """A single carried stream runs its feature layer a chunk of queries ahead of the output layer on a second HIP
    stream (chunks of 7 and 32 queries; 1000 = one launch per layer, the unpipelined form): same counts, same carried
    state, also when the call is repeated on the carried state."""
    c, frames, w1, w2, gt, G = load_config("example")
    f = torch.from_numpy(frames).cuda()
    ref = _engine(w1, w2, c["roi"], c["dims"])
    ref.CARRY_CHUNK = 1 << 30
    eng = _engine(w1, w2, c["roi"], c["dims"])
    eng.CARRY_CHUNK = chunk
    for _ in range(2):
        a, b = ref.forward_counts(f, carry=True), eng.forward_counts(f, carry=True)
        assert torch.equal(a, b)
        assert torch.equal(ref.v_feat, eng.v_feat) and torch.equal(ref.v_out, eng.v_out)
    assert np.array_equal(eng.forward_counts(f[:5], carry=True).cpu().numpy(), ref.forward_counts(f[:5], carry=True).cpu().numpy())


@pytest.mark.parametrize("tag", ["example", "sunset"])
def test_seqmatch_and_recall_match_golden(tag, O):
    c, frames, w1, w2, gt, G = load_config(tag)
    from lens_amd.engine import seqmatch, seqmatch_topk
    out = torch.from_numpy(G["out_carry"].astype(np.float32)).cuda()
    S = seqmatch(out, 2).cpu().numpy().T            # reference orientation [P',Q']
    assert np.array_equal(S, G["S_carry"])           # float32 equality, tolerance 0
    GT = O.gt_tolerance(gt, 2, 3)
    for K in (1, 5, 10, 15, 20, 25):
        idx, val = seqmatch_topk(out, 2, K)
        oi, ov = O.topk(G["S_carry"], K)
        assert np.array_equal(idx.cpu().numpy(), oi)
        assert np.array_equal(val.cpu().numpy(), ov)
        # Recall from the GPU top-k == the reference formula evaluated with a stable argsort
        assert O.recall_from_topk(idx.cpu().numpy(), GT, K) == O.recall_at_k(G["S_carry"], GT, K, kind="stable")


@pytest.mark.parametrize("P,Q,mode", [(1000, 48, "reset"), (1000, 24, "carry"), (300, 40, "reset"), (2500, 16, "reset")])
def test_synthetic_counts_match_oracle(P, Q, mode, O):
    """Config 2 shape (128x128 frames, N=100, F=200) at sizes the oracle finishes in seconds."""
    from lens_amd.synthetic import synthetic_frames, synthetic_database
    w1, w2p = pretrained_example()
    w2 = synthetic_database(w2p, P).numpy()
    frames = synthetic_frames(Q, 128, seed=99 + P)
    eng = _engine(w1, w2, 128, 10)
    out = eng.forward_counts(frames.cuda(), carry=(mode == "carry")).cpu().numpy()
    U = O.rand_U(250, 128 * 128)
    ref = O.forward(frames.numpy(), 10, U, w1, w2, carry=(mode == "carry"), skip_zero=True)
    assert np.array_equal(out, ref)
    assert out.sum() > 0


def test_multi_stream_carry_matches_oracle(O):
    """n_streams carried runs == the oracle run separately on each contiguous slice."""
    from lens_amd.synthetic import synthetic_frames, synthetic_database
    w1, w2p = pretrained_example()
    w2 = synthetic_database(w2p, 500).numpy()
    frames = synthetic_frames(30, 128, seed=5)
    eng = _engine(w1, w2, 128, 10)
    eng.reset_state(n_streams=3)
    out = eng.forward_counts(frames.cuda(), carry=True).cpu().numpy()
    U = O.rand_U(250, 128 * 128)
    ref = np.concatenate([O.forward(frames.numpy()[s * 10:(s + 1) * 10], 10, U, w1, w2, carry=True, skip_zero=True)
                          for s in range(3)])
    assert np.array_equal(out, ref)


def test_carried_tiles_through_l2_match_oracle(O):
    """Carried state with more (64-place tile, stream) wavefronts than the CUs' LDS holds tiles: the output layer reads the
    database through L2 (every wavefront resident at once).  51 tiles x 16 streams = 816 wavefronts > 512; counts equal
    the oracle's run on each stream, and the carried state continues over a second call."""
    from lens_amd.synthetic import synthetic_frames, synthetic_database
    w1, w2p = pretrained_example()
    P, S = 3230, 16
    w2 = synthetic_database(w2p, P).numpy()
    frames = synthetic_frames(3 * S, 128, seed=17)
    eng = _engine(w1, w2, 128, 10)
    assert not eng.reads_through_l2(carry=True)
    eng.reset_state(n_streams=S)
    assert eng.reads_through_l2(carry=True) and eng.tile_places(carry=True) == 64
    out = eng.forward_counts(frames[:2 * S].cuda(), carry=True).cpu().numpy()
    U = O.rand_U(250, 128 * 128)
    f = frames.numpy()
    ref = [O.forward(f[s * 2:(s + 1) * 2], 10, U, w1, w2, carry=True, skip_zero=True) for s in range(S)]
    assert np.array_equal(out, np.concatenate(ref)) and out.sum() > 0
    # second call: one more query per stream, on the carried state
    out2 = eng.forward_counts(frames[2 * S:].cuda(), carry=True).cpu().numpy()
    for s in range(S):
        both = np.concatenate([f[s * 2:(s + 1) * 2], f[2 * S + s:2 * S + s + 1]])
        assert np.array_equal(out2[s], O.forward(both, 10, U, w1, w2, carry=True, skip_zero=True)[2]), s


def test_carried_streams_on_shared_tiles_match_oracle(O):
    """Many carried streams on a database of many tiles: the batch mode's wide LDS tile is shared by 16 streams, one
    wavefront each walking its stream in order (12 tiles x 13 groups of streams here; the last group holds 8).  Counts
    equal the oracle's run on each stream, and the carried state continues over a second call."""
    from lens_amd.synthetic import synthetic_frames, synthetic_database
    w1, w2p = pretrained_example()
    P, S = 1500, 200
    w2 = synthetic_database(w2p, P).numpy()
    frames = synthetic_frames(3 * S, 128, seed=23)
    eng = _engine(w1, w2, 128, 10)
    eng.reset_state(n_streams=S)
    assert eng.layout(carry=True) == 3 and eng.tile_places(carry=True) == 128 and not eng.reads_through_l2(carry=True)
    out = eng.forward_counts(frames[:2 * S].cuda(), carry=True).cpu().numpy()
    U = O.rand_U(250, 128 * 128)
    f = frames.numpy()
    ref = [O.forward(f[s * 2:(s + 1) * 2], 10, U, w1, w2, carry=True, skip_zero=True) for s in range(S)]
    assert np.array_equal(out, np.concatenate(ref)) and out.sum() > 0
    out2 = eng.forward_counts(frames[2 * S:].cuda(), carry=True).cpu().numpy()
    for s in range(0, S, 7):
        both = np.concatenate([f[s * 2:(s + 1) * 2], f[2 * S + s:2 * S + s + 1]])
        assert np.array_equal(out2[s], O.forward(both, 10, U, w1, w2, carry=True, skip_zero=True)[2]), s


def test_shared_tiles_in_whole_rounds_and_a_tail_of_lone_wavefronts_match_oracle(O, monkeypatch):
    """16 carried streams on a database of two wide tiles more than one round of the CUs: the 256 tiles of the round
    are shared by the streams, the last 300 places run as a second launch of lone wavefronts on 128-place tiles (same
    streams).  Counts equal the oracle's run on each stream over two calls, and the run without the tail."""
    from lens_amd.synthetic import synthetic_frames, synthetic_database
    w1, w2p = pretrained_example()
    P, S, T = 256 * 192 + 300, 16, 60
    w2 = synthetic_database(w2p, P).numpy()
    frames = synthetic_frames(3 * S, 128, seed=29)
    monkeypatch.setenv("LENS_AMD_CARRY_SHARED", "3")       # (the cost model would take two rounds of 128-place tiles here)
    eng = _engine(w1, w2, 128, 10, T=T)
    eng.reset_state(n_streams=S)
    assert eng.layout(carry=True) == 3 and eng.tile_places(carry=True) == 192 and eng.tail_places(carry=True) == 300
    out = eng.forward_counts(frames[:2 * S].cuda(), carry=True).cpu().numpy()
    out2 = eng.forward_counts(frames[2 * S:].cuda(), carry=True).cpu().numpy()
    U = O.rand_U(T, 128 * 128)
    f = frames.numpy()
    for s in range(S):
        both = np.concatenate([f[s * 2:(s + 1) * 2], f[2 * S + s:2 * S + s + 1]])
        ref = O.forward(both, 10, U, w1, w2, carry=True, skip_zero=True)
        assert np.array_equal(out[2 * s:2 * s + 2], ref[:2]) and np.array_equal(out2[s], ref[2]), s
    assert out[:, -300:].sum() > 0 and out[:, :-300].sum() > 0
    monkeypatch.setenv("LENS_AMD_CARRY_TAIL", "0")
    eng.reset_state(n_streams=S)
    assert eng.layout(carry=True) == 3 and eng.tail_places(carry=True) == 0
    assert np.array_equal(eng.forward_counts(frames[:2 * S].cuda(), carry=True).cpu().numpy(), out)


def test_odd_sizes_and_wide_layers(O):
    """Ragged sizes: N=49 (one partial register), F=63, P=641 already covered by sunset; here a wide
    feature layer (F=520 -> 16 registers/lane rounds to the 32 variant) and N=144."""
    rng = np.random.default_rng(3)
    dims, roi, T = 12, 36, 40
    N, F, P, Q = dims * dims, 520, 77, 6
    w1 = (rng.standard_normal((F, N)) * 0.6).astype(np.float32)
    w2 = (rng.standard_normal((P, F)) * 0.05).astype(np.float32)
    frames = rng.integers(0, 60, size=(Q, roi, roi)).astype(np.uint8)
    eng = _engine(w1, w2, roi, dims, T=T)
    out = eng.forward_counts(torch.from_numpy(frames).cuda(), carry=True).cpu().numpy()
    U = O.rand_U(T, roi * roi)
    ref = O.forward(frames, dims, U, w1, w2, carry=True)
    assert np.array_equal(out, ref)


def test_empty_and_saturated_frames(O):
    """All-zero frames never spike; all-255 frames spike every step (px >= code always)."""
    w1, w2 = pretrained_example()
    frames = np.zeros((4, 80, 80), np.uint8)
    frames[1] = 255
    frames[3] = 255
    eng = _engine(w1, w2, 80, 10)
    out = eng.forward_counts(torch.from_numpy(frames).cuda(), carry=True).cpu().numpy()
    U = O.rand_U(250, 6400)
    ref = O.forward(frames, 10, U, w1, w2, carry=True)
    assert np.array_equal(out, ref)
    assert out[0].sum() == 0


def test_large_multi_spike_counts_are_exact(O):
    """Hundreds of spikes per step (multi-spike) travel through the stream as full float32 counts."""
    dims, roi, T = 2, 2, 6
    w1 = np.full((3, 4), 64.25, np.float32)      # 4 inputs * 64.25 -> 257 spikes in one step
    w1[1] = 0.3
    w2 = (np.arange(15, dtype=np.float32).reshape(5, 3) - 7) * 0.013
    eng = _engine(w1, w2, roi, dims, T=T)
    frames = np.array([[[255, 255], [255, 255]], [[200, 3], [90, 255]]], np.uint8)
    out = eng.forward_counts(torch.from_numpy(frames).cuda(), carry=True).cpu().numpy()
    ref = O.forward(frames, dims, O.rand_U(T, 4), w1, w2, carry=True)
    assert np.array_equal(out, ref) and out.max() > 100


def test_topk_ties_and_short_lists(O):
    """Heavy ties (scores in {0,1,2}) and K larger than the number of places."""
    from lens_amd.engine import seqmatch_topk
    rng = np.random.default_rng(11)
    out = rng.integers(0, 3, size=(9, 700)).astype(np.float32)
    for L in (1, 2, 3, 5):
        S = O.seqmatch(out, L)
        for K in (1, 25, 64):
            idx, val = seqmatch_topk(torch.from_numpy(out).cuda(), L, K)
            oi, ov = O.topk(S, K)
            assert np.array_equal(idx.cpu().numpy(), oi)
            assert np.array_equal(val.cpu().numpy(), ov)
    small = rng.integers(0, 3, size=(4, 10)).astype(np.float32)
    idx, val = seqmatch_topk(torch.from_numpy(small).cuda(), 2, 25)
    oi, ov = O.topk(O.seqmatch(small, 2), 25)
    assert np.array_equal(idx.cpu().numpy(), oi) and np.array_equal(val.cpu().numpy(), ov)


def test_sharded_topk_merge_equals_unsharded(O):
    """Place shards with an (L-1)-column halo, local top-k with global indices, merge == global."""
    from lens_amd.engine import seqmatch_topk, topk_merge
    rng = np.random.default_rng(2)
    Q, P, L, K, R = 33, 1000, 2, 25, 8
    out = rng.integers(0, 6, size=(Q, P)).astype(np.float32)
    full_i, full_v = seqmatch_topk(torch.from_numpy(out).cuda(), L, K)
    Ps = P - L + 1
    bounds = [r * Ps // R for r in range(R + 1)]
    parts_i, parts_v = [], []
    for r in range(R):
        p0, p1 = bounds[r], bounds[r + 1]
        shard = np.ascontiguousarray(out[:, p0:p1 + L - 1])
        i, v = seqmatch_topk(torch.from_numpy(shard).cuda(), L, K, n_places=p1 - p0, place_offset=p0)
        parts_i.append(i); parts_v.append(v)
    mi, mv = topk_merge(torch.stack(parts_i), torch.stack(parts_v))
    assert torch.equal(mi, full_i) and torch.equal(mv, full_v)


def test_accumulate_events_matches_oracle(O):
    from lens_amd.engine import accumulate_events
    from lens_amd.synthetic import synthetic_frames, synthetic_events
    frames = synthetic_frames(5, 128, seed=8)
    frames[2] = 0                                   # a window without events
    frames[3, 10, 20] = 255                         # saturate, then push past the uint8 wrap below
    xy, off = synthetic_events(frames)
    extra = torch.tensor([[21, 11]] * 3, dtype=torch.int16)   # 3 more events on pixel (10,20) of frame 4
    xy = torch.cat([xy, extra]); off[-1] += 3
    got = accumulate_events(xy.cuda(), off.cuda(), 128).cpu().numpy()
    ref = O.accumulate_events(xy.numpy()[:, 0], xy.numpy()[:, 1], off.numpy(), 128)
    assert np.array_equal(got, ref)
    assert np.array_equal(got[:4], frames.numpy()[:4])
    # wrap-around of the uint8 cast: 256 events on one pixel -> 0
    xyw = torch.tensor([[5, 7]] * 256, dtype=torch.int16)
    offw = torch.tensor([0, 256], dtype=torch.int64)
    gw = accumulate_events(xyw.cuda(), offw.cuda(), 16).cpu().numpy()
    assert gw[0, 6, 4] == 0 and gw.sum() == 0
    # python-style negative index (x = 0 -> column -1) and an out-of-range event
    xyn = torch.tensor([[0, 1]], dtype=torch.int16)
    gn = accumulate_events(xyn.cuda(), torch.tensor([0, 1]).cuda(), 16).cpu().numpy()
    assert gn[0, 0, 15] == 1
    with pytest.raises(IndexError):
        accumulate_events(torch.tensor([[17, 1]], dtype=torch.int16).cuda(), torch.tensor([0, 1]).cuda(), 16)
    with pytest.raises(IndexError):
        O.accumulate_events(np.array([17], np.int16), np.array([1], np.int16), np.array([0, 1]), 16)


def test_sad_matrix_matches_reference_and_oracle(O):
    """lens_sad_matrix == torch.cdist(p=1) of the reference's run_sad on the bundled frames, and == the oracle
    on ragged sizes (HW not a multiple of 16, Q not a multiple of the query tile)."""
    import os
    from PIL import Image
    from helpers import ROOT, GOLDEN
    from lens_amd.engine import sad_matrix
    G = np.load(os.path.join(GOLDEN, "pr_sad.npz"))
    base = os.path.join(ROOT, "lens/dataset/example/davis128")
    q = np.stack([np.array(Image.open(os.path.join(base, "example-query", f))) for f in G["files_query"]])
    r = np.stack([np.array(Image.open(os.path.join(base, "example-reference", f))) for f in G["files_reference"]])
    d = sad_matrix(torch.from_numpy(q).cuda(), torch.from_numpy(r).cuda()).cpu().numpy()
    assert np.array_equal(d, G["sad_dist"])
    rng = np.random.default_rng(9)
    q2 = rng.integers(0, 256, size=(13, 7, 7)).astype(np.uint8)
    r2 = rng.integers(0, 256, size=(301, 7, 7)).astype(np.uint8)
    d2 = sad_matrix(torch.from_numpy(q2).cuda(), torch.from_numpy(r2).cuda()).cpu().numpy()
    assert np.array_equal(d2, O.sad_matrix(q2, r2))


@pytest.mark.parametrize("P,Q", [(1000, 40), (130, 33)])
def test_dense_mfma_variant_is_bit_identical(P, Q, O):
    """The f32-MFMA dense kernel (k-ordered fmaf chain in hardware) == the event-driven kernel == the oracle."""
    from lens_amd.synthetic import synthetic_frames, synthetic_database
    w1, w2p = pretrained_example()
    w2 = synthetic_database(w2p, P).numpy()
    frames = synthetic_frames(Q, 128, seed=17)
    eng = _engine(w1, w2, 128, 10)
    a = eng.forward_counts(frames.cuda(), carry=False).cpu().numpy()
    b = eng.forward_counts_dense(frames.cuda()).cpu().numpy()
    assert np.array_equal(a, b)
    ref = O.forward(frames.numpy(), 10, O.rand_U(250, 128 * 128), w1, w2, carry=False, skip_zero=True)
    assert np.array_equal(b, ref) and b.sum() > 0
    c, fr, w1b, w2b, gt, G = load_config("sunset")          # F = 63 (k-pair padding), P = 641
    eng2 = _engine(w1b, w2b, c["roi"], c["dims"])
    out = eng2.forward_counts_dense(torch.from_numpy(fr[:50]).cuda()).cpu().numpy()
    assert np.array_equal(out.astype(np.int16), G["out_reset"][:50])


@pytest.mark.parametrize("mode", ["carry", "reset"])
def test_batch_chunking_is_transparent(mode, O):
    """Queries processed in several launches (max_batch < Q) give the same counts, in both state modes."""
    c, frames, w1, w2, gt, G = load_config("example")
    eng = _engine(w1, w2, c["roi"], c["dims"], max_batch=7)
    out = eng.forward_counts(torch.from_numpy(frames[:31]).cuda(), carry=(mode == "carry")).cpu().numpy()
    assert np.array_equal(out.astype(np.int16), G[f"out_{mode}"][:31])


def test_output_layer_without_lds_tile(O):
    """F = 700 (> 624): the W2^T tile does not fit in LDS at any width, weights are read through L2
    (`k_output_counts<4, false>`); also exercises the 16-register feature variant without LDS tables."""
    rng = np.random.default_rng(21)
    dims, roi, T = 6, 12, 30
    N, F, P, Q = dims * dims, 700, 333, 9
    w1 = (rng.standard_normal((F, N)) * 0.5).astype(np.float32)
    w2 = (rng.standard_normal((P, F)) * 0.04).astype(np.float32)
    frames = rng.integers(0, 90, size=(Q, roi, roi)).astype(np.uint8)
    eng = _engine(w1, w2, roi, dims, T=T)
    assert eng.ev_scale == 1
    U = O.rand_U(T, roi * roi)
    for carry in (True, False):
        eng.reset_state()
        out = eng.forward_counts(torch.from_numpy(frames).cuda(), carry=carry).cpu().numpy()
        assert np.array_equal(out, O.forward(frames, dims, U, w1, w2, carry=carry)) and out.sum() > 0


def test_random_shape_sweep(O):
    """A dozen random small configurations (fan-ins, feature counts on every tile width, ragged P, short and
    long T, dense and sparse frames, both state modes): counts, sequence match and top-k all bit-exact."""
    from lens_amd.engine import seqmatch, seqmatch_topk
    rng = np.random.default_rng(2024)
    shapes = [(3, 9, 40, 70, 17), (4, 8, 150, 129, 33), (5, 10, 157, 300, 16), (8, 16, 313, 64, 5), (2, 6, 1, 5, 64),
              (7, 7, 63, 641, 31), (6, 12, 100, 257, 48), (10, 20, 200, 130, 32), (3, 3, 300, 1, 20), (4, 12, 625, 100, 9)]
    for dims, roi, F, P, T in shapes:
        N = dims * dims
        Q = int(rng.integers(3, 12))
        w1 = (rng.standard_normal((F, N)) * rng.choice([0.2, 0.8])).astype(np.float32)
        w1[rng.random((F, N)) < 0.3] = 0
        w2 = (rng.standard_normal((P, F)) * rng.choice([0.03, 0.3])).astype(np.float32)
        dens = rng.choice([2, 40, 255])
        frames = rng.integers(0, dens + 1, size=(Q, roi, roi)).astype(np.uint8)
        frames[rng.integers(0, Q)] = 0
        eng = _engine(w1, w2, roi, dims, T=T)
        U = O.rand_U(T, roi * roi)
        for carry in (False, True):
            eng.reset_state()
            out = eng.forward_counts(torch.from_numpy(frames).cuda(), carry=carry)
            ref = O.forward(frames, dims, U, w1, w2, carry=carry)
            assert np.array_equal(out.cpu().numpy(), ref), (dims, roi, F, P, T, carry)
        for L in (1, 2):
            if Q >= L and P >= L:
                assert np.array_equal(seqmatch(out, L).cpu().numpy().T, O.seqmatch(ref, L))
                K = min(25, 64)
                gi, gv = seqmatch_topk(out, L, K)
                oi, ov = O.topk(O.seqmatch(ref, L), K)
                assert np.array_equal(gi.cpu().numpy(), oi) and np.array_equal(gv.cpu().numpy(), ov)


def _encode_stream(steps, ev_scale, chunk=16):
    """Spike stream of one query in the layout documented at the top of lens_kernels.hip (white-box helper):
    steps = per time step a list of (feature, count) in ascending feature order."""
    ev = []                                   # [offset word, count word]
    flagged = False
    for t, st in enumerate(steps):
        if not st:
            if flagged:
                continue
            last_mag = (ev[-1][1] & 0x7fffffff) if ev else 0
            if t > 0 and last_mag < 0x4b000000:
                ev[-1][1] |= 1                # EV_EMPTY_NEXT on the previous LAST event
                flagged = True
            else:
                ev.append([0, 0x80000000])    # explicit empty step
            continue
        for k, (f, c) in enumerate(st):
            bits = int(np.float32(c).view(np.uint32))
            if k == len(st) - 1:
                bits |= 0x80000000
            ev.append([f * ev_scale, bits])
        flagged = False
    while len(ev) % chunk:
        ev.append([0, 0x3f800000])
    for c0 in range(0, len(ev), chunk):
        if any(((w1 & 0x7ffffffe) != 0x3f800000) for _, w1 in ev[c0:c0 + chunk]):
            ev[c0][0] |= 0x80000000           # EV_CHUNK_MIXED
    return np.array(ev, dtype=np.uint32).reshape(-1)


# (F, P, places per LDS tile in batch mode): one case per generated assembly loop (64-, 128-, 192- and 256-place
# tiles) and one for the C++ loop that reads the weights through L2; the tile width follows from (F, P)
@pytest.mark.parametrize("F,P,tile", [(400, 5, 64), (200, 100, 128), (200, 150, 192), (100, 250, 256), (700, 5, 0)])
def test_stream_consumer_dropped_steps(O, F, P, tile):
    """lens_output_counts on hand-built streams against the oracle's dense Linear + IAF, step by step.  The
    streams leave out the steps without spikes; the cases where such a step is not a no-op are all here:
    a membrane that the clip leaves at exactly 1.0 (weight 1 - 2^-24) fires in the first dropped step, an
    empty step 0 on carried state, counts > 1 (mixed chunks) and >= 2^23 (no room for the flag bit).
    Batch mode runs the tile width named by the case (asserted); carried state runs the width the library picks
    for one wavefront per tile (the 64-place loop, or L2)."""
    import ctypes
    from lens_amd import _lib
    assert _lib.lib().lens_output_tile_places(F, P, 0, 0) == tile
    pred1 = float(np.nextafter(np.float32(1.0), np.float32(0.0)))
    w2 = np.zeros((P, F), dtype=np.float32)       # the places beyond the first five stay silent
    w2[:5, 0] = [pred1, 0.3, pred1, 0.45, 1e-7]   # a lone spike of feature 0 on rested membranes: nobody reaches 1.0,
    w2[:5, 1] = [-1.0, 0.2, pred1, 1.5, 1e-7]     # places 0 and 2 land on 1 - 2^-24, which the clip rounds up to 1.0
    big = float(2 ** 23)
    queries = [
        [[], [(0, 1)], [], [(1, 1)], [], [], [(0, 1), (1, 2)], [], [(0, 1)], [], [(0, 3)], [(0, 1)], [(1, 1)], []],
        [[(0, 1)], [], [], [(0, 1)], [(1, big)], [], [(1, 1)], [(0, 1), (1, 1)], [], [(0, 1)], [], [], [], [(0, 1)]],
        [[], [], [], [(1, 1)], [(0, 1)], [], [(0, 1)], [], [(0, 1)], [], [(0, 1)], [], [(0, 2), (1, 5)], []],
        [[(0, 1)]] * 3 + [[]] * 30 + [[(0, 1), (1, 1)]] * 17 + [[]],
        # odd counts >= 2^23: the mantissa LSB is part of the count there, not the "empty steps follow" flag
        [[(1, big + 1)], [(0, 1)], [(0, big + 3), (1, big + 5)], [], [(1, 1)], [(0, 2 * big + 2)], [(1, big + 7)], []],
    ]
    T = max(len(q) for q in queries)
    queries = [q + [[]] * (T - len(q)) for q in queries]
    w1 = np.zeros((F, 1), dtype=np.float32)
    eng = _engine(w1, w2, 1, 1, T=max(T, 4))
    Q = len(queries)
    eng._workspace(Q)
    L = O.lib()
    fp = ctypes.POINTER(ctypes.c_float)
    ptr = lambda a: a.ctypes.data_as(fp)
    for carry in (False, True):
        eng.reset_state()
        assert eng.tile_places(carry) == (tile if not carry else min(tile, 64))
        words = np.zeros((Q, eng.cap), dtype=np.uint32)
        nw = np.zeros(Q, dtype=np.int32)
        for q, steps in enumerate(queries):
            enc = _encode_stream(steps, eng.event_scale(carry))      # the row stride of this mode's tile
            words[q, :enc.size] = enc
            nw[q] = enc.size
        eng.events[:Q].copy_(torch.from_numpy(words.view(np.int32)))
        eng.nwords[:Q].copy_(torch.from_numpy(nw))
        if carry:   # a carried state that is not a clip output: step 0 may not be dropped
            v0 = np.zeros(P, dtype=np.float32)
            v0[:5] = [1.0, 2.5, -3.0, 0.99999994, 1e-30]
            eng.v_out.copy_(torch.from_numpy(v0).cuda()[None, :])
        got = eng.output_counts(Q, carry=carry).cpu().numpy()
        v = v0.copy() if carry else None
        ref = np.zeros((Q, P), dtype=np.float32)
        for q, steps in enumerate(queries):
            if not carry:
                v = np.zeros(P, dtype=np.float32)
            for st in steps:
                x = np.zeros(F, dtype=np.float32)
                for f, c in st:
                    x[f] = c
                cur = np.zeros(P, dtype=np.float32)
                spk = np.zeros(P, dtype=np.float32)
                L.lens_oracle_linear(ptr(x), ptr(w2), ptr(cur), F, P)
                L.lens_oracle_iaf_step(ptr(v), ptr(cur), ptr(spk), P)
                ref[q] += spk
        assert np.array_equal(got, ref), (F, P, carry, got[:, :5], ref[:, :5])
        if carry:
            assert np.array_equal(eng.v_out.cpu().numpy()[0], v)
    assert ref[0, 0] != 0 or ref[3, 2] != 0
    if not tile:
        return

    # The carried run once more in every other layout the library has for carried state, as S identical streams: each
    # must repeat the single stream's counts and final state.
    def replay(S, layout, places):
        engs = _engine(w1, w2, 1, 1, T=max(T, 4), max_batch=Q * S)
        engs.reset_state(n_streams=S)
        assert (engs.layout(True), engs.tile_places(True)) == (layout, places), (F, P, S, engs.layout(True), engs.tile_places(True))
        engs._workspace(Q * S)
        enc_w = np.zeros((Q, engs.cap), dtype=np.uint32)
        enc_n = np.zeros(Q, dtype=np.int32)
        for q, steps in enumerate(queries):
            enc = _encode_stream(steps, engs.event_scale(True))
            enc_w[q, :enc.size] = enc
            enc_n[q] = enc.size
        engs.events[:Q * S].copy_(torch.from_numpy(enc_w.view(np.int32)).repeat(S, 1))
        engs.nwords[:Q * S].copy_(torch.from_numpy(enc_n).repeat(S))
        engs.v_out.copy_(torch.from_numpy(v0).cuda()[None, :].expand(S, -1))
        gots = engs.output_counts(Q * S, carry=True).cpu().numpy().reshape(S, Q, P)
        assert np.array_equal(gots, np.broadcast_to(ref, (S, Q, P))), (F, P, S, layout, places)
        assert np.array_equal(engs.v_out.cpu().numpy(), np.broadcast_to(v, (S, P))), (F, P, S, layout, places)

    # 520 streams: more (64-place tile, stream) wavefronts than the CUs' LDS holds tiles -> read through L2 (k_output_carry_l2)
    replay(520, 2, 64)
    import os
    saved = {k: os.environ.get(k) for k in ("LENS_AMD_CARRY_L2", "LENS_AMD_CARRY_SHARED")}
    try:
        os.environ["LENS_AMD_CARRY_SHARED"], os.environ["LENS_AMD_CARRY_L2"] = "0", "1"
        replay(1, 2, 64)                                   # a single stream through L2
        os.environ["LENS_AMD_CARRY_L2"] = "0"
        replay(3, 1, 64)                                   # lone wavefronts on LDS tiles, three streams
        for ppl in (2, 3, 4):                              # wide tiles shared by up to 16 streams: every assembly loop
            if F * 64 * ppl * 4 <= 160 * 1024 - 4096:
                os.environ["LENS_AMD_CARRY_SHARED"] = str(ppl)
                replay(1, 3, 64 * ppl)
                replay(20, 3, 64 * ppl)                    # two groups of streams: 16 wavefronts + 4 (12 idle)
    finally:
        for k, val in saved.items():
            if val is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = val


def test_event_list_ingest_matches_oracle(O):
    """Timestamped events -> time windows -> count frames in one launch; checked against the oracle's
    restatement of create_images (lens/collect_data.py:193-199) window by window."""
    from lens_amd.events import frames_from_events, window_events
    rng = np.random.default_rng(5)
    n, roi = 200_000, 64
    t = np.sort(rng.integers(0, 40_000_000, size=n)).astype(np.float64)        # 40 s of microsecond stamps
    t[(t > 7e6) & (t < 12.5e6)] = 7e6                                          # a silent stretch: windows without events
    x = rng.integers(1, roi + 1, size=n)
    y = rng.integers(1, roi + 1, size=n)
    x[:3000] = 5; y[:3000] = 9                                                 # > 255 hits on one pixel: uint8 wrap
    frames, win = frames_from_events(t, x, y, timebin_ms=1000.0, roi_dim=roi)
    off, win2 = window_events(t, 1000.0)
    assert np.array_equal(win, win2) and len(win) < 40 and win[-1] == 39
    ref = O.accumulate_events(x, y, off, roi)
    assert np.array_equal(frames.cpu().numpy(), ref)
    with pytest.raises(IndexError):
        frames_from_events(t[:10], np.full(10, roi + 70), y[:10], 1000.0, roi)


def test_batch_pipeline_equals_single_stream(O):
    """Two-stream batch pipelining is host orchestration only: same counts and top-k as one engine, batch by batch."""
    from lens_amd.pipeline import BatchPipeline
    from lens_amd.engine import seqmatch_topk
    from lens_amd.synthetic import synthetic_frames, synthetic_database
    w1, w2p = pretrained_example()
    w2 = synthetic_database(w2p, 700)
    batches = [synthetic_frames(q, 128, seed=50 + i).cuda() for i, q in enumerate((96, 64, 33, 96, 1, 80))]
    eng = _engine(w1, w2, 128, 10, max_batch=96)
    pipe = BatchPipeline(w1, w2, 128, 10, 250, max_batch=96)
    counts = pipe.run(batches)
    tops = pipe.run([b for b in batches if b.shape[0] >= 2], seq_len=2, topk=25)
    k = 0
    for b, c in zip(batches, counts):
        ref = eng.forward_counts(b, carry=False)
        assert torch.equal(c, ref)
        if b.shape[0] >= 2:
            ri, rv = seqmatch_topk(ref, 2, 25)
            assert torch.equal(tops[k][0], ri) and torch.equal(tops[k][1], rv)
            k += 1
    assert np.array_equal(counts[2].cpu().numpy(),
                          O.forward(batches[2].cpu().numpy(), 10, O.rand_U(250, 128 * 128), np.asarray(w1, np.float32),
                                    w2.numpy(), carry=False, skip_zero=True))


def test_two_pass_carried_feature_layer_writes_identical_streams(O):
    """Carried state: lens_feature_currents + lens_feature_fire (input currents of all queries in parallel, then the
    integrate-and-fire walk with one wavefront per 64 features and a merge of their event lists) against the
    one-wavefront lens_feature_events: the same words in every stream (flags, explicit empty steps, padding), the
    same lengths and the same carried membranes -- on the bundled sets and on shapes that exercise counts above 1,
    silent steps, a ragged last wavefront, T not a multiple of the prefetch depth and several streams."""
    from lens_amd import _lib
    from lens_amd.engine import _ptr, _stream
    rng = np.random.default_rng(77)
    cases = []
    for tag in ("example", "sunset"):
        c, frames, w1, w2, gt, G = load_config(tag)
        cases.append((w1, w2[:64], c["roi"], c["dims"], 250, frames[:40], 1))
    for dims, roi, F, T, dens, wscale, streams in [(5, 10, 157, 33, 255, 0.8, 1), (8, 16, 313, 250, 40, 0.3, 3),
                                                   (3, 9, 40, 7, 255, 3.0, 1), (10, 20, 768, 19, 90, 0.5, 2),
                                                   (6, 12, 64, 64, 2, 0.9, 1), (4, 8, 65, 100, 255, 1e9, 1)]:
        N = dims * dims
        w1 = (rng.standard_normal((F, N)) * wscale).astype(np.float32)
        w1[rng.random((F, N)) < 0.3] = 0
        w2 = (rng.standard_normal((8, F)) * 0.1).astype(np.float32)
        frames = rng.integers(0, dens + 1, size=(11, roi, roi)).astype(np.uint8)
        frames[3] = 0
        cases.append((w1, w2, roi, dims, T, frames, streams))
    L = _lib.lib()
    for w1, w2, roi, dims, T, frames, streams in cases:
        eng = _engine(w1, w2, roi, dims, T=T)
        assert eng._two_pass
        fr = torch.from_numpy(frames).cuda()
        Q = fr.shape[0]
        got = []
        for two_pass in (False, True):
            eng.reset_state(streams)
            eng.v_feat.copy_(torch.linspace(-0.9, 0.99, eng.v_feat.numel(), device="cuda").view_as(eng.v_feat))
            eng._workspace(Q)
            eng.events.fill_(-1); eng.nwords.fill_(-1)
            for rep in range(2):                                 # the second call starts from the first one's state
                if two_pass:
                    eng.feature_events(fr, carry=True)
                else:
                    st = L.lens_feature_events(_ptr(fr), Q, eng.HW, _ptr(eng.sel), eng.N, _ptr(eng.codes), eng.T, _ptr(eng.w1t),
                                               eng.F, eng.F, _ptr(eng.v_feat), streams, 1, _ptr(eng.events), eng.cap,
                                               _ptr(eng.nwords), eng.event_scale(True), None, None, _stream())
                    assert st == 0
            torch.cuda.synchronize()
            nw = eng.nwords[:Q].cpu().numpy()
            ev = eng.events[:Q].cpu().numpy()
            got.append((nw, [ev[q, :nw[q]].copy() for q in range(Q)], eng.v_feat.cpu().numpy().copy()))
        (nw0, ev0, v0), (nw1, ev1, v1) = got
        assert np.array_equal(nw0, nw1), (w1.shape, T, nw0, nw1)
        for q in range(Q):
            assert np.array_equal(ev0[q], ev1[q]), (w1.shape, T, q)
        assert np.array_equal(v0, v1)
        assert nw0.max() > 32 or w1.shape[0] == 64


def test_dealing_queries_to_wavefronts_changes_no_stream(O):
    """Batch mode with the load-balancing table: the 16 queries of a workgroup are dealt to its wavefronts by work.
    Which wavefront integrates a query must not show: same stream lengths and words as without the table, for a batch
    that ends in a partial workgroup, and the counts equal the oracle's."""
    from lens_amd import _lib
    from lens_amd.engine import _ptr, _stream
    from lens_amd.synthetic import synthetic_frames
    w1, w2p = pretrained_example()
    eng = _engine(w1, w2p, 128, 10)
    Q = 16 * 5 + 7
    frames = synthetic_frames(Q, 128, seed=5)
    frames[3] = 0
    frames[20:24] = 200            # very heavy queries in one workgroup
    fr = frames.cuda()
    eng._workspace(Q)
    got = []
    for table in (None, eng.active_steps):
        eng.events.fill_(-1); eng.nwords.fill_(-1)
        st = _lib.lib().lens_feature_events(_ptr(fr), Q, eng.HW, _ptr(eng.sel), eng.N, _ptr(eng.codes), eng.T, _ptr(eng.w1t),
                                            eng.F, eng.F, None, 0, 0, _ptr(eng.events), eng.cap, _ptr(eng.nwords),
                                            eng.event_scale(False), None, _ptr(table), _stream())
        assert st == 0
        torch.cuda.synchronize()
        nw = eng.nwords[:Q].cpu().numpy()
        ev = eng.events[:Q].cpu().numpy()
        got.append((nw, [ev[q, :nw[q]].copy() for q in range(Q)]))
    assert np.array_equal(got[0][0], got[1][0])
    for q in range(Q):
        assert np.array_equal(got[0][1][q], got[1][1][q]), q
    out = eng.output_counts(Q, carry=False).cpu().numpy()
    ref = O.forward(frames.numpy(), 10, O.rand_U(250, 128 * 128), np.asarray(w1, np.float32), np.asarray(w2p, np.float32), carry=False)
    assert np.array_equal(out, ref)
