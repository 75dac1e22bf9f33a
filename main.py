#!/usr/bin/env python3
"""Command-line entry point: the flag set, defaults, mode dispatch and model-file naming of the
reference's main.py (lines 27-38, 40-76, 85-186), served by the MI355X hot path in lens_amd.

    python main.py --sim_mat --matching            # bundled example, GPU

Inference-and-matching and --train_model are implemented.  The neuromorphic-hardware modes
(--event_driven, --simulated_speck, --collect_data) stop with a clear message; --nocuda is accepted but
refused at model construction (this build has no CPU execution path).
"""
import argparse
import sys

# (flag, type, default, help) -- one row per reference flag, same order of groups as main.py:85-186
_VALUE_FLAGS = [
    ("dataset", str, "example", "dataset folder under data_dir"),
    ("camera", str, "davis128", "camera sub-folder of the dataset"),
    ("data_name", str, "experiment001", "name of a newly collected data set"),
    ("reference", str, "example-reference", "reference (database) traverse"),
    ("query", str, "example-query", "query traverse"),
    ("data_dir", str, "./lens/dataset/", "root of the dataset files"),
    ("reference_places", int, 100, "number of database places"),
    ("query_places", int, 100, "number of query frames"),
    ("sequence_length", int, 2, "length of the diagonal sequence matcher (0 = off)"),
    ("feature_multiplier", float, 2.0, "feature layer size = multiplier * dims^2"),
    ("filter", int, 1, "use every n-th image of the traverse"),
    ("epoch_feat", int, 128, "training epochs, feature layer"),
    ("epoch_out", int, 128, "training epochs, output layer"),
    ("thr_l_feat", float, 0, "feature layer: low threshold"),
    ("thr_h_feat", float, 0.75, "feature layer: high threshold"),
    ("fire_l_feat", float, 0.4, "feature layer: low firing rate"),
    ("fire_h_feat", float, 0.6, "feature layer: high firing rate"),
    ("ip_rate_feat", float, 0.02, "feature layer: ITP learning rate"),
    ("stdp_rate_feat", float, 0.01, "feature layer: STDP learning rate"),
    ("thr_l_out", float, 0, "output layer: low threshold"),
    ("thr_h_out", float, 0.5, "output layer: high threshold"),
    ("fire_l_out", float, 0.5, "output layer: low firing rate"),
    ("fire_h_out", float, 0.5, "output layer: high firing rate"),
    ("ip_rate_out", float, 0.02, "output layer: ITP learning rate"),
    ("stdp_rate_out", float, 0.01, "output layer: STDP learning rate"),
    ("f_exc", float, 0.35, "feature layer excitatory connection probability"),
    ("f_inh", float, 0.75, "feature layer inhibitory connection probability"),
    ("o_exc", float, 1.0, "output layer excitatory connection probability"),
    ("o_inh", float, 1.0, "output layer inhibitory connection probability"),
    ("dims", int, 10, "side of the down-sampled input (dims x dims input neurons)"),
    ("roi_dim", int, 80, "side of the input frames in pixels"),
    ("GT_tolerance", int, 3, "ground-truth tolerance in frames"),
    ("timebin", int, 250, "time steps of the rate-coded simulation per query"),
]
_SWITCHES = [
    ("train_model", "train a new network on the reference traverse"),
    ("sim_mat", "plot the similarity matrix"),
    ("PR_curve", "plot a precision-recall curve"),
    ("matching", "evaluate Recall@N against the ground truth"),
    ("sad", "run the sum-of-absolute-differences baseline"),
    ("nocuda", "force the CPU (refused: this build is GPU-only)"),
    ("event_driven", "online inference on a Speck2fDevKit (hardware mode, not supported)"),
    ("simulated_speck", "time-based simulation on a Speck2fDevKit (hardware mode, not supported)"),
    ("collect_data", "record frames from a Speck2fDevKit (hardware mode, not supported)"),
    ("headless", "run the Speck2fDevKit without GUI"),
    ("save_input", "save the input spikes as numpy arrays"),
]


def build_parser():
    p = argparse.ArgumentParser(description="LENS visual place recognition on MI355X (lens_amd)")
    for name, typ, default, text in _VALUE_FLAGS:
        p.add_argument("--" + name, type=typ, default=default, help=text)
    for name, text in _SWITCHES:
        p.add_argument("--" + name, action="store_true", help=text)
    # build-only extension (not in the reference): membrane state handling across queries
    p.add_argument("--amd_state", choices=["carry", "reset"], default="carry",
                   help="carry: reference semantics (state persists from query to query); "
                        "reset: zero the state per query (batch-parallel)")
    p.add_argument("--amd_topk", choices=["auto", "always", "never"], default="auto",
                   help="Recall@N from the GPU's fused sequence-match + top-25 instead of a host argsort of the full "
                        "matrix (auto: when places x queries exceeds 2^24; the full-matrix artefacts are then skipped "
                        "unless --sim_mat / --PR_curve ask for them)")
    return p


def generate_model_name(model):
    """<reference>_LENS_IN<inputs>_FN<features>_DB<places>.pth (main.py:27-38)."""
    return f"{''.join(model.reference)}_LENS_IN{model.input}_FN{model.feature}_DB{model.reference_places}.pth"


def initialize_and_run_model(args):
    from lens_amd import checker
    checker.check_args(args)
    for flag in ("collect_data", "event_driven", "simulated_speck"):
        if getattr(args, flag):
            sys.exit(f"--{flag} drives SynSense Speck hardware through samna; lens_amd implements the "
                     "GPU inference-and-matching path only")
    if args.train_model:
        from lens_amd.train_model import LENS_Trainer, train_model
        model = LENS_Trainer(args)
        return train_model(model, generate_model_name(model))
    from lens_amd.run_model import LENS, run_inference
    model = LENS(args)
    return run_inference(model, generate_model_name(model))


def parse_network(argv=None):
    args = build_parser().parse_args(argv)
    return initialize_and_run_model(args)


if __name__ == "__main__":
    parse_network()
