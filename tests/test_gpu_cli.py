"""GPU end-to-end test of the drop-in surface: `python main.py --matching` (config 1 inputs) must
reproduce the reference run's similarity matrix and Recall@N table (tests/golden)."""
import glob
import os

import numpy as np
import pytest

from helpers import ROOT, GOLDEN

pytestmark = pytest.mark.gpu

SUNSET = ["--dataset", "brisevent", "--camera", "davis346", "--reference", "sunset2", "--query", "sunset1",
          "--reference_places", "641", "--query_places", "724", "--dims", "7", "--roi_dim", "7",
          "--feature_multiplier", "1.3"]


def _run(argv):
    import main
    cwd = os.getcwd()
    os.chdir(ROOT)
    try:
        before = set(glob.glob("lens/output/*"))
        R = main.parse_network(argv)
        new = sorted(set(glob.glob("lens/output/*")) - before)
        assert len(new) == 1
        return R, os.path.join(ROOT, new[0])
    finally:
        os.chdir(cwd)


@pytest.mark.parametrize("tag,argv", [("example", []), ("sunset", SUNSET)])
def test_cli_matching_reproduces_reference_run(tag, argv):
    G = np.load(os.path.join(GOLDEN, f"{tag}.npz"))
    R, folder = _run(["--matching"] + argv)
    assert R == [round(float(x), 2) for x in G["recall_carry"]]
    S = np.load(os.path.join(folder, "similarity_matrix.npy"))
    assert S.dtype == np.float32 and np.array_equal(S, G["S_carry"])
    for f in ("lens.log", "distance_matrix_lens.pdf", "GTtol.pdf"):
        assert os.path.exists(os.path.join(folder, f))
    log = open(os.path.join(folder, "lens.log")).read()
    assert "Recall" in log and "Succesfully completed inferencing using LENS" in log


@pytest.mark.parametrize("tag,argv", [("example", []), ("sunset", SUNSET)])
def test_cli_recall_from_gpu_topk(tag, argv):
    """`--amd_topk always`: Recall@N comes from the fused sequence matcher + top-25 on the GPU (what the CLI does by
    itself for large databases).  It equals the reference's recallAtK under a stable sort -- the GPU path's explicit
    tie policy -- and every listed place lies in the tie set of the reference's own (unstable) sort: its score is the
    score the reference's ranking has at that rank."""
    from lens_amd.metrics import recallAtK
    G = np.load(os.path.join(GOLDEN, f"{tag}.npz"))
    R, folder = _run(["--matching", "--amd_topk", "always"] + argv)
    S, GT = G["S_carry"], G["GTtol"]
    assert R == [round(recallAtK(S, GT, K=n, sort_kind="stable"), 2) for n in (1, 5, 10, 15, 20, 25)]
    idx = np.load(os.path.join(folder, "topk_places.npy"))                  # [queries, 25], best first
    assert idx.shape == (S.shape[1], 25)
    ranked = -np.sort(-S, axis=0)[:25, :]                                   # scores by rank, per query column
    assert np.array_equal(np.take_along_axis(S, idx.T.astype(np.int64), axis=0), ranked)
    assert all(len(set(col)) == 25 for col in idx)                          # 25 different places per query
    log = open(os.path.join(folder, "lens.log")).read()
    assert "ties: higher score first, then the larger place index" in log
    assert np.array_equal(np.load(os.path.join(folder, "similarity_matrix.npy")), S)   # `always` keeps the artefacts


def test_cli_reset_state_variant():
    G = np.load(os.path.join(GOLDEN, "example.npz"))
    R, folder = _run(["--matching", "--amd_state", "reset"])
    assert R == [round(float(x), 2) for x in G["recall_reset"]]
    assert np.array_equal(np.load(os.path.join(folder, "similarity_matrix.npy")), G["S_reset"])


def test_cli_refuses_cpu_and_hw_modes():
    from lens_amd._lib import LensAmdError
    with pytest.raises(LensAmdError):
        _run(["--matching", "--nocuda"])
    with pytest.raises(SystemExit):
        _run(["--event_driven"])


def test_cli_pr_curve_and_sad_reproduce_reference_outputs():
    """`--matching --PR_curve --sad`: LENS PR curve, SAD PR curve and SAD Recall@N equal what the reference's
    createPR / run_sad produced on the same data, and the comparison artefacts are written."""
    import json
    PR = np.load(os.path.join(GOLDEN, "pr_sad.npz"))
    R, folder = _run(["--matching", "--PR_curve", "--sad"])
    lens = json.load(open(os.path.join(folder, "PR_curve_LENS.json")))
    sad = json.load(open(os.path.join(folder, "PR_curve_SAD.json")))
    assert np.array_equal(np.array(lens["Precision"], np.float64), PR["lens_P"])
    assert np.array_equal(np.array(lens["Recall"], np.float64), PR["lens_R"])
    assert np.array_equal(np.array(sad["Precision"], np.float64), PR["sad_P"])
    assert np.array_equal(np.array(sad["Recall"], np.float64), PR["sad_R"])
    rows = open(os.path.join(folder, "SAD_vs_LENS_RecallatN.csv")).read().strip().splitlines()
    assert rows[0] == "N,SAD,LENS" and [float(x.split(",")[1]) for x in rows[1:]] == list(PR["sad_recall"])
    for f in ("Precision_Recall_plot.pdf", "SAD_vs_LENS_RecallatN.pdf", "distance_matrix_SAD.pdf",
              "similarity_matrix_LENS.pdf", "similarity_matrix_SAD.pdf"):
        assert os.path.exists(os.path.join(folder, f)), f
    with pytest.raises(NameError):
        _run(["--PR_curve"])
