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
            torch.cuda.synchronize()
            before = (torch.cuda.memory_allocated(), torch.cuda.memory_reserved())
            eng.forward_counts_static(frames, out, carry=carry)
            torch.cuda.synchronize()
            assert (torch.cuda.memory_allocated(), torch.cuda.memory_reserved()) == before   # torch's allocator saw nothing
            ref = ref_eng.forward_counts(frames, carry=carry)
            assert torch.equal(out, ref), (carry, frames.shape[0])
            assert float(out.sum()) > 0
        if carry:
            assert torch.equal(eng.v_out, ref_eng.v_out) and torch.equal(eng.v_feat, ref_eng.v_feat)
    # a call that the output layer would refuse (here: a counts matrix narrower than P) is refused before the feature
    # layer is enqueued: negative status, carried state of both layers untouched
    from lens_amd.engine import _ptr, _stream
    eng.reset_state(); eng.v_feat.fill_(0.25); eng.v_out.fill_(-0.5)
    out = torch.empty((small.shape[0], eng.P), dtype=torch.float32, device="cuda")
    st = _lib.lib().lens_forward_counts_static(
        _ptr(small), small.shape[0], eng.HW, _ptr(eng.sel), eng.N, _ptr(eng.codes), eng.T, _ptr(eng.w1t), eng.F, eng.F,
        _ptr(eng.w2t), eng.P, eng.ldw2, _ptr(eng.v_feat), _ptr(eng.v_out), eng.P, 1, 1, _ptr(out), eng.P - 1, _stream())
    torch.cuda.synchronize()
    assert st == -2 and bool((eng.v_feat == 0.25).all()) and bool((eng.v_out == -0.5).all())
    # an empty batch is the documented warm-up call; a batch beyond the array's size is refused, nothing enqueued
    eng.forward_counts_static(big[:0], torch.empty((0, eng.P), dtype=torch.float32, device="cuda"))
    too_many = 64 * 1024 * 1024 // (4 * eng.cap) + 1
    frames = synthetic_frames(too_many, 128, seed=73).cuda()
    out = torch.empty((too_many, eng.P), dtype=torch.float32, device="cuda")
    with pytest.raises(_lib.LensAmdError, match="static scratch"):
        eng.forward_counts_static(frames, out)
