"""Inference driver: the host-side mirror of lens/run_model.py for the GPU hot path.

Same public surface as the reference (`LENS(args)`, `.add_layer`, `.load_model`, `.evaluate`,
`run_inference`, attribute names, model-file keys, log/PDF artefacts, the returned Recall list);
the numeric core -- raster, both Linear layers, the three sinabs IAF layers, the time-sum, the
sequence matcher and (for large databases) the top-k -- runs in liblens_amd.so on the GPU.

Semantics kept from the reference, on purpose:
  * the membrane state is carried from query to query (the reference never calls reset_states(),
    lens/run_model.py:234-242); `--amd_state reset` selects the per-query-reset variant instead;
  * Recall@N uses numpy's default argsort on the full matrix, like lens/src/metrics.py:219 -- for a small
    matrix.  From 2^24 entries on (or with `--amd_topk always`) it comes from the GPU's fused sequence matcher +
    top-25, whose tie policy is explicit (score descending, then place index descending = what a STABLE ascending
    argsort + [-K:] returns) and is written to the log; the full matrix is then neither copied to the host nor sorted;
  * sequence_length == 0 skips the matcher and leaves the matrix untransposed (run_model.py:253-254).
There is no CPU execution path: without a ROCm GPU (or with --nocuda) construction fails.
"""
import os

import numpy as np
import torch
import torch.nn as nn

from . import blitnet as bn
from ._lib import LensAmdError
from .dataset import CustomImageDataset
from .loggers import model_logger
from .metrics import recallAtK, recall_from_topk, createPR

RECALL_N = [1, 5, 10, 15, 20, 25]
TOPK_AUTO_ENTRIES = 1 << 24      # places x queries from which --amd_topk auto ranks on the GPU


def format_table(field_names, row):
    """ASCII table in PrettyTable's default layout (the reference logs a PrettyTable, run_model.py:305-308)."""
    cells = [[str(x) for x in field_names], [str(x) for x in row]]
    width = [max(len(c[i]) for c in cells) + 2 for i in range(len(field_names))]
    rule = "+" + "+".join("-" * w for w in width) + "+"
    fmt = lambda c: "|" + "|".join(s.center(w) for s, w in zip(c, width)) + "|"
    return "\n".join([rule, fmt(cells[0]), rule, fmt(cells[1]), rule])


def create_GTtol(GT, distance=2):
    """Ground truth with tolerance: binary dilation with a (2d+1)^2 square (run_model.py:272-289).
    Host-side scipy, as in the reference."""
    from scipy.ndimage import binary_dilation
    return binary_dilation(GT, structure=np.ones((2 * distance + 1, 2 * distance + 1), dtype=int)).astype(int)


class LENS(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.args = args
        for name in vars(args):
            setattr(self, name, getattr(args, name))
        self.dataset_file = os.path.join(self.data_dir, self.query + ".csv")
        self.query_dir = os.path.join(self.data_dir, self.dataset, self.camera, self.query)
        self.reference_dir = os.path.join(self.data_dir, self.dataset, self.camera, self.reference)
        self.device = model_logger(self)
        if getattr(self, "nocuda", False) or self.device.type != "cuda":
            raise LensAmdError("lens_amd runs the LENS hot path on an MI355X only: no ROCm device is usable "
                               "(--nocuda was given or none is present) and there is no CPU fallback")
        self.layer_dict = {}
        self.layer_counter = 0
        self.input = int(args.dims * args.dims)
        self.feature = int(self.input * self.feature_multiplier)
        self.output = int(args.reference_places)
        self.add_layer("feature_layer", dims=[self.input, self.feature], device=self.device, inference=True)
        self.add_layer("output_layer", dims=[self.feature, self.output], device=self.device, inference=True)
        if not hasattr(self, "matrix"):
            self.matrix = None
        self.kernel_size = self.roi_dim // self.dims
        self.engine = None

    def add_layer(self, name, **kwargs):
        if name in self.layer_dict:
            raise ValueError(f"Layer with name {name} already exists.")
        setattr(self, name, bn.SNNLayer(**kwargs))
        self.layer_dict[name] = self.layer_counter
        self.layer_counter += 1

    def load_model(self, model_path):
        """torch.save(state_dict) files of the reference: keys <layer>.thr, <layer>.w.weight (f32)."""
        self.load_state_dict(torch.load(model_path, map_location="cpu", weights_only=True), strict=False)

    # ------------------------------------------------------------------------------------
    def build_engine(self):
        from .engine import LensEngine
        self.engine = LensEngine(self.feature_layer.w.weight.detach(), self.output_layer.w.weight.detach(),
                                 self.roi_dim, self.dims, self.timebin, device=self.device)
        return self.engine

    def similarity_rows(self, frames):
        """out[Q,P]: spike counts of the output layer per query (run_model.py:229-246)."""
        if self.engine is None:
            self.build_engine()
        carry = getattr(self, "amd_state", "carry") != "reset"
        self.engine.reset_state()
        counts = self.engine.forward_counts(frames.to(self.device), carry=carry)
        return counts

    def evaluate(self, test_loader, model):
        """test_loader: a CustomImageDataset (or anything with .frames() -> uint8 [Q,H,W])."""
        import matplotlib
        if os.environ.get("DISPLAY", "") == "" and not os.environ.get("MPLBACKEND"):
            matplotlib.use("Agg")
        import matplotlib.pyplot as plt
        from .engine import seqmatch

        if hasattr(test_loader, "frames_to_device"):
            frames = test_loader.frames_to_device(self.device)
        else:
            frames = test_loader.frames() if hasattr(test_loader, "frames") else test_loader
        counts = self.similarity_rows(frames)
        assert tuple(counts.shape) == (model.query_places, model.reference_places)
        L = self.sequence_length
        mode = getattr(self, "amd_topk", "auto")
        use_topk = self.matching and L != 0 and max(RECALL_N) <= model.reference_places - L + 1 and (
            mode == "always" or (mode == "auto" and counts.numel() >= TOPK_AUTO_ENTRIES))
        # the full matrix on the host: always for a small one (the reference's artefacts), for a large one only if asked
        want_matrix = not use_topk or mode == "always" or self.sim_mat or self.PR_curve
        dist_matrix_seq = None
        if want_matrix:
            if L != 0:
                S_dev = seqmatch(counts, L)                                     # query-major on the device
                dist_matrix_seq = np.ascontiguousarray(S_dev.cpu().numpy().T)   # reference orientation
            else:
                dist_matrix_seq = counts.cpu().numpy().astype(np.float64)
            np.save(os.path.join(self.output_folder, "similarity_matrix.npy"), dist_matrix_seq)
            plt.imshow(dist_matrix_seq)
            plt.colorbar()
            plt.savefig(os.path.join(self.output_folder, "distance_matrix_lens.pdf"))
            plt.close()
        else:
            model.logger.info(f"similarity matrix {model.reference_places} x {model.query_places}: kept on the GPU, "
                              "not written or plotted (--sim_mat / --PR_curve / --amd_topk always bring it to the host)")

        R = []
        if self.matching:
            GT = np.load(os.path.join(self.data_dir, self.dataset, self.camera,
                                      self.reference + "_" + self.query + "_GT.npy"))
            if L != 0:
                GT = GT[L - 2:-1, L - 2:-1]
            GTtol = create_GTtol(GT, distance=self.GT_tolerance).T
            if want_matrix:
                plt.imshow(GTtol)
                plt.colorbar()
                plt.savefig(os.path.join(self.output_folder, "GTtol.pdf"))
                plt.close()
            if use_topk:
                from .engine import seqmatch_topk
                idx, _ = seqmatch_topk(counts, L, max(RECALL_N))
                idx = idx.cpu().numpy()
                np.save(os.path.join(self.output_folder, "topk_places.npy"), idx)
                for n in RECALL_N:
                    R.append(round(recall_from_topk(idx, GTtol, n), 2))
                model.logger.info(f"Recall@N from the GPU top-{max(RECALL_N)} lists; ties: higher score first, then the "
                                  "larger place index (a stable ascending argsort + [-K:]; the reference's default "
                                  "argsort leaves the order of ties unspecified)")
            else:
                for n in RECALL_N:
                    R.append(round(recallAtK(dist_matrix_seq, GTtol, K=n), 2))
            model.logger.info(format_table(["N"] + [str(n) for n in RECALL_N], ["Recall"] + R))

        if self.sim_mat:
            plt.figure(figsize=(10, 8))
            plt.imshow(dist_matrix_seq, aspect="auto", cmap="viridis")
            plt.colorbar()
            plt.title("Similarity matrix")
            plt.xlabel("Query")
            plt.ylabel("Database")
            plt.savefig(os.path.join(self.output_folder, "similarity_matrix.pdf"))
            plt.show()
            plt.close()

        # Same flag coupling as the reference (run_model.py:319-333): --PR_curve needs --matching (GTtol),
        # --sad needs --PR_curve (lens_PR); the reference dies with a NameError there, here the message says why.
        if self.PR_curve:
            if not self.matching:
                raise NameError("GTtol is not defined: --PR_curve needs --matching (as in the reference)")
            LENS_P, LENS_R = createPR(dist_matrix_seq.T, GTtol.T, self.output_folder, matching="single", n_thresh=100)
            lens_PR = {"Precision": LENS_P, "Recall": LENS_R}
        if self.sad:
            if not (self.matching and self.PR_curve):
                raise NameError("lens_PR is not defined: --sad needs --matching and --PR_curve (as in the reference)")
            from .sad import run_sad
            from .plot_results import plot_PR, plot_recall
            sad_PR, sad_Recall = run_sad(self.reference_dir, self.query_dir, GTtol, self.output_folder,
                                         self.sequence_length, logger=model.logger)
            plot_PR(lens_PR, sad_PR, self.output_folder)
            plot_recall(R, sad_Recall, RECALL_N, self.output_folder)

        model.logger.info("")
        model.logger.info("Succesfully completed inferencing using LENS")
        return R


def run_inference(model, model_name):
    """Load the query set and the pretrained model, run `evaluate` (lens/run_model.py:360-397)."""
    test_dataset = CustomImageDataset(annotations_file=model.dataset_file, img_dir=model.query_dir,
                                      kernel_size=model.kernel_size, skip=model.filter,
                                      max_samples=model.query_places, is_spiking=True,
                                      time_window=model.timebin)
    model.eval()
    model.load_model(os.path.join("./lens/models", model_name))
    return model.evaluate(test_dataset, model)
