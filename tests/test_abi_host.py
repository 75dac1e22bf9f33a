"""CPU tests (no GPU): the C-ABI library loads and exports every symbol the header declares, its host
helpers and argument checks work, and the host-side logic (pixel selection, threshold codes, CLI
surface, sharding arithmetic) matches the reference behaviour."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from helpers import ROOT, GOLDEN, load_config


@pytest.fixture(scope="module")
def L():
    from lens_amd import _lib
    _lib.build()
    return _lib.lib()


def test_every_declared_symbol_is_exported_and_bound(L):
    from lens_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "lens_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(lens_[a-z_0-9]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    raw = ctypes.CDLL(_lib.SO_PATH)
    for name in declared:
        assert getattr(raw, name) is not None
    assert L.lens_amd_abi_version() == _lib.ABI_VERSION


def test_argument_errors_are_negative_statuses_without_touching_the_gpu(L):
    assert L.lens_feature_events(None, 1, 1, None, 1, None, 1, None, 1, 1, None, 1, 0, None, 32, None, 1, None, None, None) == -1
    assert L.lens_output_counts(None, 16, None, 1, None, 1, 1, 256, None, 1, 1, 0, None, 1, None) == -1
    assert L.lens_seqmatch_topk(None, 1, 1, 1, 1, 1, 0, None, None, None) == -1
    assert L.lens_accumulate_events(None, None, 0, 8, None, None, None) == -1
    buf = (ctypes.c_char * 64)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert L.lens_accumulate_events(p, p, 1, 500, p, p, None) == -3          # roi too large for the LDS histogram
    assert L.lens_seqmatch_topk(p, 4, 8, 4, 2, 65, 0, p, p, None) == -2       # K > 64
    assert L.lens_seqmatch_topk(p, 1, 8, 4, 2, 5, 0, p, p, None) == -2        # fewer queries than L
    assert L.lens_output_counts(p, 32, p, 1, p, 4, 10, 100, None, 10, 1, 0, p, 10, None) == -4   # ldw2 % 256
    assert L.lens_output_counts(p, 16, p, 1, p, 4, 10, 256, None, 10, 1, 0, p, 10, None) == -2   # streams hold whole 32-word chunks
    assert L.lens_output_counts(p, 48, p, 1, p, 4, 10, 256, None, 10, 1, 0, p, 10, None) == -2
    # carried state read through L2: the streams carry byte offsets of rows at the minimal ldw2, any other is refused
    assert L.lens_output_counts(p, 32, p, 1, p, 200, 100000, 100352, p, 100000, 1, 1, p, 100000, None) == -2
    assert L.lens_amd_strerror(-2).decode().startswith("a size")
    assert L.lens_event_capacity_words(250, 200) == 100000
    assert L.lens_event_capacity_words(3, 0) == 32
    # tile width by (F, P): 256 places while they fit (F <= 156), else 192 or 128 -- whichever pads P less, times its rate
    assert L.lens_event_scale(63, 1000, 0, 0) == 1024 and L.lens_event_scale(5000, 1000, 0, 0) == 1
    assert L.lens_event_scale(200, 100000, 0, 0) == 512 and L.lens_event_scale(200, 1000, 0, 0) == 512
    assert L.lens_event_scale(200, 64, 0, 0) == 256 and L.lens_event_scale(300, 100000, 0, 0) == 512
    # tile widths: batch mode by cost per place; carried state (one wavefront per tile) by tiles resident at once
    assert [L.lens_output_tile_places(200, p, 0, 0) for p in (64, 1000, 100000)] == [64, 128, 192]
    assert L.lens_output_tile_places(100, 100000, 0, 0) == 256 and L.lens_output_tile_places(700, 5, 0, 0) == 0
    assert L.lens_output_tile_places(200, 100000, 1, 1) == 64 and L.lens_event_scale(200, 1000, 1, 1) == 256
    assert L.lens_event_scale(200, 100000, 1, 1) == 4 * 100096      # read through L2: an offset word is the row's byte offset
    # carried state reads the database through L2 once its (64-place tile, stream) wavefronts outnumber the tiles the
    # CUs' LDS surely holds (one less than fits per CU: 2 x 256 at F = 200); the batch mode never does for layers that fit a tile
    rd = L.lens_output_reads_through_l2
    assert [rd(200, p, 1, 1) for p in (1000, 32768, 32769, 100000)] == [0, 0, 1, 1]
    assert [rd(200, 3000, 1, g) for g in (1, 10, 11)] == [0, 0, 1]
    # the three carried layouts (1 lone wavefronts on LDS tiles, 2 through L2, 3 wide tiles shared by 16 streams)
    lay = L.lens_output_layout
    assert lay(200, 100000, 0, 0) == 0 and [lay(200, 100000, 1, g) for g in (1, 2, 4, 16)] == [2, 2, 3, 3] and lay(200, 12501, 1, 4) == 2
    assert [lay(200, 1000, 1, g) for g in (1, 16, 32, 64)] == [1, 1, 1, 2] and lay(200, 12501, 1, 32) == 3
    assert L.lens_output_tile_places(200, 100000, 1, 16) == 192 and L.lens_event_scale(200, 100000, 1, 16) == 512
    assert L.lens_output_tile_places(200, 12501, 1, 32) == 128 and lay(700, 5, 1, 4) == 1
    assert rd(200, 100000, 0, 0) == 0 and rd(700, 5, 0, 0) == 1 and rd(700, 5, 1, 1) == 1
    # shared tiles cover whole rounds of the 256 CUs; the places beyond run as lone wavefronts on 128-place tiles
    tail = L.lens_output_tail_places
    assert [tail(200, 100000, 1, g) for g in (1, 4, 16, 64)] == [0, 100000 - 512 * 192, 100000 - 512 * 192, 0]
    assert tail(200, 100000, 0, 0) == 0 and tail(200, 12501, 1, 32) == 0 and tail(200, 98304, 1, 16) == 0
    assert tail(200, 512 * 192 + 300, 1, 16) == 300 and tail(200, 256 * 192 + 300, 1, 16) == 0    # two rounds of 128-place tiles there


def test_active_step_table_counts_input_spikes(L):
    """lens_build_active_steps: table[n][p] = number of steps in which input n spikes at pixel value p, i.e. the row sums
    of the raster (the feature kernel's load-balancing hint is the exact number of input spikes of a query)."""
    rng = np.random.default_rng(3)
    T, N = 37, 11
    codes = rng.integers(1, 256, size=(T, N)).astype(np.uint8)
    table = np.zeros((N, 256), dtype=np.uint16)
    assert L.lens_build_active_steps(codes.ctypes.data_as(ctypes.c_void_p), T, N, table.ctypes.data_as(ctypes.c_void_p)) == 0
    want = (np.arange(256)[None, None, :] >= codes[:, :, None]).sum(0)
    assert np.array_equal(table, want)
    assert L.lens_build_active_steps(None, T, N, table.ctypes.data_as(ctypes.c_void_p)) == -1


def test_product_refuses_to_run_without_gpu():
    """No CPU fallback: engine construction and device ops raise when there is no ROCm device."""
    from lens_amd._lib import LensAmdError
    from lens_amd.engine import LensEngine, seqmatch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(LensAmdError):
        LensEngine(np.zeros((4, 4), np.float32), np.zeros((3, 4), np.float32), 2, 2, 5, device="cpu")
    with pytest.raises(LensAmdError):
        seqmatch(torch.zeros(4, 4), 2)


@pytest.mark.parametrize("roi,dims", [(80, 10), (128, 10), (7, 7), (36, 12), (9, 4)])
def test_centre_pixel_index_is_the_reference_conv(roi, dims):
    """sel[n] picks exactly the pixel the one-hot Conv2d of lens/run_model.py:130-137 selects."""
    from lens_amd.engine import centre_pixel_index
    k = roi // dims
    conv = torch.nn.Conv2d(1, 1, kernel_size=k, stride=k, padding=0, bias=False)
    w = torch.zeros(1, 1, k, k)
    c = (k // 2) - 1
    w[0, 0, c, c] = 1
    conv.weight = torch.nn.Parameter(w, requires_grad=False)
    img = torch.arange(roi * roi, dtype=torch.float32).reshape(1, 1, roi, roi)
    want = conv(img)[0, 0, :dims, :dims].reshape(-1).numpy().astype(np.int64)
    assert np.array_equal(centre_pixel_index(roi, dims), want[:dims * dims])


@pytest.mark.parametrize("T,roi,dims", [(250, 80, 10), (250, 7, 7), (40, 128, 10)])
def test_threshold_codes_reproduce_the_float_raster(T, roi, dims, L):
    """(px >= code)  <=>  (U < px/255 in float32) for every one of the 256 pixel values."""
    from lens_amd.engine import build_threshold_codes, centre_pixel_index, rand_tensor
    sel = centre_pixel_index(roi, dims)
    codes = build_threshold_codes(T, roi, sel).numpy()
    U = rand_tensor(T, roi * roi)[:, torch.as_tensor(sel, dtype=torch.long)]
    assert codes.min() >= 1
    for px in range(256):
        img = (torch.tensor([px], dtype=torch.uint8) / 255)            # dataset.py:23
        want = (U < img).numpy()
        assert np.array_equal(codes <= px, want), px


def test_rand_tensor_matches_golden_fingerprint():
    import hashlib
    from lens_amd.engine import rand_tensor
    R = np.load(os.path.join(GOLDEN, "raster.npz"))
    U = rand_tensor(250, 6400).numpy()
    assert hashlib.sha256(U.tobytes()).digest() == R["U_250_6400_sha256"].tobytes()


def test_cli_surface_matches_the_reference():
    """44 reference flags with their defaults (main.py:85-186) + model-name rule (main.py:27-38)."""
    import main
    p = main.build_parser()
    a = p.parse_args([])
    ref_defaults = dict(dataset="example", camera="davis128", data_name="experiment001", reference="example-reference",
                        query="example-query", data_dir="./lens/dataset/", reference_places=100, query_places=100,
                        sequence_length=2, feature_multiplier=2.0, filter=1, epoch_feat=128, epoch_out=128,
                        thr_l_feat=0, thr_h_feat=0.75, fire_l_feat=0.4, fire_h_feat=0.6, ip_rate_feat=0.02,
                        stdp_rate_feat=0.01, thr_l_out=0, thr_h_out=0.5, fire_l_out=0.5, fire_h_out=0.5,
                        ip_rate_out=0.02, stdp_rate_out=0.01, f_exc=0.35, f_inh=0.75, o_exc=1.0, o_inh=1.0, dims=10,
                        roi_dim=80, train_model=False, GT_tolerance=3, sim_mat=False, PR_curve=False, matching=False,
                        sad=False, timebin=250, nocuda=False, event_driven=False, simulated_speck=False,
                        collect_data=False, headless=False, save_input=False)
    assert len(ref_defaults) == 44
    for k, v in ref_defaults.items():
        assert getattr(a, k) == v, k

    class M:
        reference = "sunset2"; input = 49; feature = 63; reference_places = 641
    assert main.generate_model_name(M) == "sunset2_LENS_IN49_FN63_DB641.pth"
    assert os.path.exists(os.path.join(ROOT, "lens", "models", main.generate_model_name(M)))


def test_checker_assertions(tmp_path):
    import main
    from lens_amd import checker
    a = main.build_parser().parse_args([])
    cwd = os.getcwd()
    os.chdir(ROOT)
    try:
        checker.check_args(a)
        bad = main.build_parser().parse_args(["--dims", "9", "--roi_dim", "80"])    # 80//9 = 8 -> conv gives 10, not 9
        with pytest.raises(AssertionError, match="ROI dimension"):
            checker.check_args(bad)
        bad = main.build_parser().parse_args(["--query_places", "101"])
        with pytest.raises(AssertionError, match="Not enough query images"):
            checker.check_args(bad)
        bad = main.build_parser().parse_args(["--dataset", "nope"])
        with pytest.raises(AssertionError, match="Data directory does not exist"):
            checker.check_args(bad)
    finally:
        os.chdir(cwd)


def test_model_file_loads_into_the_mirror_container():
    """The reference's state_dict keys load bit-exactly into blitnet.SNNLayer containers."""
    from lens_amd.blitnet import SNNLayer
    sd = torch.load(os.path.join(ROOT, "lens", "models", "example-reference_LENS_IN100_FN200_DB100.pth"),
                    map_location="cpu", weights_only=True)

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.feature_layer = SNNLayer(dims=[100, 200], inference=True)
            self.output_layer = SNNLayer(dims=[200, 100], inference=True)
    net = Net()
    missing = net.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    assert torch.equal(net.feature_layer.w.weight, sd["feature_layer.w.weight"])
    assert net.output_layer.thr.shape == (1, 100)


def test_dataset_loader_and_metrics_mirror():
    from lens_amd.dataset import CustomImageDataset
    from lens_amd.metrics import recallAtK, recall_from_topk
    from oracle import oracle as O
    c, frames, w1, w2, gt, G = load_config("example")
    ds = CustomImageDataset(os.path.join(ROOT, "lens/dataset/example-query.csv"),
                            os.path.join(ROOT, "lens/dataset/example/davis128/example-query"), skip=3, max_samples=7)
    assert len(ds) == 7 and np.array_equal(ds.frames().numpy(), frames[::3][:7])
    assert np.array_equal(ds.frames(workers=1).numpy(), ds.frames(workers=4).numpy())
    assert ds[2][1] == 6
    with pytest.raises(FileNotFoundError):
        CustomImageDataset(os.path.join(ROOT, "lens/dataset/example-query.csv"), "/nonexistent")[0]
    GT = G["GTtol"]
    S = G["S_carry"]
    R = [recallAtK(S, GT, K=k) for k in (1, 5, 10, 15, 20, 25)]
    assert np.allclose(R, G["recall_carry"], rtol=0, atol=0)           # same numbers as lens/src/metrics.py
    idx, _ = O.topk(S, 25)
    for k in (1, 5, 25):
        assert recall_from_topk(idx, GT, k) == recallAtK(S, GT, K=k, sort_kind="stable")
    with pytest.raises(AssertionError):
        recallAtK(S, GT[:-1], K=1)


def test_table_format_and_shard_bounds():
    from lens_amd.run_model import format_table
    from lens_amd.dist import shard_bounds, shard_rows
    t = format_table(["N", "1", "5"], ["Recall", 0.63, 0.84]).splitlines()
    assert t[0] == "+--------+------+------+" and t[1] == "|   N    |  1   |  5   |" and t[3] == "| Recall | 0.63 | 0.84 |"
    for P, Lq, W in ((100000, 2, 8), (1000, 2, 3), (641, 3, 5), (10, 1, 4)):
        b = shard_bounds(P, Lq, W)
        assert b[0] == 0 and b[-1] == P - Lq + 1 and all(b[i] <= b[i + 1] for i in range(W))
        rows = [shard_rows(P, Lq, r, W) for r in range(W)]
        assert rows[-1][1] == P and sum(r[2] for r in rows) == P - Lq + 1
        assert all(r[1] - r[0] == r[2] + Lq - 1 for r in rows)


def test_synthetic_generators_are_deterministic():
    from lens_amd.synthetic import synthetic_frames, synthetic_database, synthetic_events
    a, b = synthetic_frames(3, 32, seed=5), synthetic_frames(3, 32, seed=5)
    assert torch.equal(a, b) and a.dtype == torch.uint8
    xy, off = synthetic_events(a)
    from oracle import oracle as O
    assert np.array_equal(O.accumulate_events(xy[:, 0].numpy(), xy[:, 1].numpy(), off.numpy(), 32), a.numpy())
    w = torch.randn(10, 6)
    db = synthetic_database(w, 25)
    assert db.shape == (25, 6) and torch.equal(db, synthetic_database(w, 25))


def test_training_mode_init_and_preprocessing_match_the_reference():
    """Same seeds -> the mirror's SNNLayer training-mode construction and the training-image preprocessing are
    bit-identical to what the reference built in the golden run (its w0, thr0, fire_rate, masks, x_table)."""
    from lens_amd.blitnet import SNNLayer
    from lens_amd.train_model import training_inputs, anneal_schedule
    from lens_amd.dataset import CustomImageDataset
    G = np.load(os.path.join(GOLDEN, "train_example.npz"))
    places, epochs, seed = (int(v) for v in G["meta"][:3])
    torch.manual_seed(seed); np.random.seed(seed)
    feat = SNNLayer(dims=[100, 200], thr_range=[0, 0.75], fire_rate=[0.4, 0.6], ip_rate=0.02, stdp_rate=0.01, p=[0.35, 0.75])
    out = SNNLayer(dims=[200, places], thr_range=[0, 0.5], fire_rate=[0.5, 0.5], ip_rate=0.02, stdp_rate=0.01, p=[1.0, 1.0],
                   spk_force=True)
    for name, L in (("feature_layer", feat), ("output_layer", out)):
        assert np.array_equal(L.w.weight.detach().numpy(), G[f"{name}_w0"]), name
        assert np.array_equal(L.thr.detach().numpy(), G[f"{name}_thr0"])
        assert np.array_equal(L.fire_rate.numpy(), G[f"{name}_fire_rate"])
        assert np.array_equal(L.havconnCombinedExc.numpy(), G[f"{name}_exc"])
        assert np.array_equal(L.havconnCombinedInh.numpy(), G[f"{name}_inh"])
        T = G[f"{name}_order"].size
        eip, est, _, _ = anneal_schedule(L, T)
        assert np.array_equal(torch.stack([eip, est], 1).numpy(), G[f"{name}_eta_used"])
    ds = CustomImageDataset(os.path.join(ROOT, "lens/dataset/example-reference.csv"),
                            os.path.join(ROOT, "lens/dataset/example/davis128/example-reference"), max_samples=places, test=False)
    x = training_inputs(ds.frames(), 8)
    assert np.array_equal(x.numpy(), G["x_table"])
    assert np.array_equal(np.array([ds[i][1] for i in range(len(ds))]), G["labels"])


def test_createPR_matches_the_reference():
    """createPR('single', 100 thresholds) on the example run's matrix == lens/src/metrics.py::createPR output."""
    from lens_amd.metrics import createPR
    G = np.load(os.path.join(GOLDEN, "example.npz"))
    PR = np.load(os.path.join(GOLDEN, "pr_sad.npz"))
    P, R = createPR(G["S_carry"].T, G["GTtol"].astype(int).T, None, matching="single", n_thresh=100)
    assert np.array_equal(np.array(P, np.float64), PR["lens_P"]) and np.array_equal(np.array(R, np.float64), PR["lens_R"])
    P2, R2 = createPR(G["S_carry"], G["GTtol"].astype(int), None, matching="multi", n_thresh=7)
    assert len(P2) == 8 and P2[0] == 1 and R2[0] == 0 and R2[-1] == 1.0


def test_event_windows_skip_empty_windows(tmp_path):
    """Host side of the event-list ingest: windows of `timebin` ms, windows without events give no frame
    (lens/collect_data.py:193-204), timestamps must be ordered."""
    from lens_amd.events import window_events, load_event_text
    t = np.array([10, 11, 999, 1000, 1000, 2999, 7000, 7001], dtype=np.float64)          # microseconds
    off, win = window_events(t, timebin_ms=1.0)                                            # 1 ms = 1000 ticks
    assert off.tolist() == [0, 5, 6, 8] and win.tolist() == [0, 2, 6]                     # t0 = first event (t = 10)
    off, win = window_events(t, timebin_ms=1.0, t0=0.0)
    assert off.tolist() == [0, 3, 5, 6, 8] and win.tolist() == [0, 1, 2, 7]
    off, win = window_events(np.zeros(0), 1.0)
    assert off.tolist() == [0] and win.size == 0
    with pytest.raises(ValueError):
        window_events(np.array([5.0, 4.0]), 1.0)
    with pytest.raises(ValueError):
        window_events(t, 1.0, t0=50.0)
    p = tmp_path / "ev.txt"
    p.write_text("128 128\n100 3 4 1\n250.5 7 8 0\n")
    tt, x, y, pol = load_event_text(str(p))
    assert tt.tolist() == [100.0, 250.5] and x.tolist() == [3, 7] and y.tolist() == [4, 8] and pol.tolist() == [1, 0]
    # the same recording inside a zip archive (member <stem>.txt, lens/tools/dvstools.py:270-283) and as parquet
    import zipfile
    from lens_amd.events import load_events
    z = tmp_path / "rec.zip"
    with zipfile.ZipFile(z, "w") as zf:
        zf.writestr("rec.txt", p.read_text())
        zf.writestr("notes.md", "x")
    for got in (load_events(str(z)), load_events(str(p))):
        assert [a.tolist() for a in got] == [[100.0, 250.5], [3, 7], [4, 8], [1, 0]]
    import pyarrow as pa
    import pyarrow.parquet as pq
    pq.write_table(pa.table({"t": [100.0, 250.5], "x": [3, 7], "y": [4, 8]}), tmp_path / "rec.parquet")
    tt, x, y, pol = load_events(str(tmp_path / "rec.parquet"))
    assert tt.tolist() == [100.0, 250.5] and x.tolist() == [3, 7] and y.tolist() == [4, 8] and pol.tolist() == [0, 0]
    with pytest.raises(ValueError):
        pq.write_table(pa.table({"t": [1.0], "x": [3]}), tmp_path / "bad.parquet")
        load_events(str(tmp_path / "bad.parquet"))


def test_training_pad_covers_wide_layers_and_short_csv_annealing():
    """Fan-ins up to 2048 have a kernel instantiation (e.g. --dims 20 => 400 inputs, 800 features); the learning-rate
    annealing horizon is reference_places * epochs (lens/train_model.py:124-125,143) even when fewer steps are taken."""
    import torch
    from lens_amd import _lib
    from lens_amd.train_model import anneal_schedule
    L = _lib.lib()
    assert [L.lens_train_in_pad(n) for n in (1, 64, 65, 200, 201, 256, 257, 800, 1024, 1025, 2048, 2049)] == \
        [64, 64, 128, 200, 208, 256, 512, 1024, 1024, 2048, 2048, 0]

    class Layer:
        eta_ip = torch.tensor(0.02)
        eta_stdp = torch.tensor(0.01)
    ip, st, _, _ = anneal_schedule(Layer, 25, horizon=100)
    assert float(st[0]) == float(torch.tensor(0.01)) and float(st[1]) == float(torch.mul(torch.tensor(0.01), pow(100 / 100, 2)))
    assert float(st[11]) == float(torch.mul(torch.tensor(0.01), pow(float(100 - 10) / 100, 2)))
    assert float(ip[21]) == float(torch.mul(torch.tensor(0.01) * 2, pow(float(100 - 20) / 100, 2)))
    full, _, _, _ = anneal_schedule(Layer, 25)
    assert float(full[21]) == float(torch.mul(torch.tensor(0.01) * 2, pow(float(25 - 20) / 25, 2)))
