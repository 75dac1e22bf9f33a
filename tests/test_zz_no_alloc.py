"""lens_forward_counts_static (the entry point that allocates nothing: its spike-stream workspace is a static
device array of the library) against the existing pair lens_feature_events + lens_output_counts, through the
C ABI, bit for bit (integer-valued spike counts, tolerance 0) -- on two calls in a row, the second with a
smaller batch, so that the array still holds the first call's streams when the second one runs."""
import numpy as np
import pytest
import torch

from helpers import pretrained_example

pytestmark = pytest.mark.gpu


def test_static_scratch_entry_point_equals_the_allocating_path():
    from lens_amd import _lib
    from lens_amd.engine import LensEngine
    from lens_amd.synthetic import synthetic_frames, synthetic_database
    w1, w2p = pretrained_example()
    w2 = synthetic_database(w2p, 700)
    eng = LensEngine(w1, w2, 128, 10, 250, device="cuda")
    ref_eng = LensEngine(w1, w2, 128, 10, 250, device="cuda")
    big = synthetic_frames(96, 128, seed=71).cuda()
    small = synthetic_frames(17, 128, seed=72).cuda()
    for carry in (False, True):
        eng.reset_state(); ref_eng.reset_state()
        for frames in (big, small):                      # second call: smaller input over the first call's leftovers
            out = torch.full((frames.shape[0], eng.P), -7.0, dtype=torch.float32, device="cuda")
            eng.forward_counts_static(frames, out, carry=carry)
            ref = ref_eng.forward_counts(frames, carry=carry)
            assert torch.equal(out, ref), (carry, frames.shape[0])
            assert float(out.sum()) > 0
        if carry:
            assert torch.equal(eng.v_out, ref_eng.v_out) and torch.equal(eng.v_feat, ref_eng.v_feat)
    # an empty batch is the documented warm-up call; a batch beyond the array's size is refused, nothing enqueued
    eng.forward_counts_static(big[:0], torch.empty((0, eng.P), dtype=torch.float32, device="cuda"))
    too_many = 64 * 1024 * 1024 // (4 * eng.cap) + 1
    frames = synthetic_frames(too_many, 128, seed=73).cuda()
    out = torch.empty((too_many, eng.P), dtype=torch.float32, device="cuda")
    with pytest.raises(_lib.LensAmdError, match="static scratch"):
        eng.forward_counts_static(frames, out)
