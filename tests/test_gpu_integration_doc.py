"""The integration guide at the repository root shows the ctypes stub a LENS maintainer would add.  This test
executes exactly that code block (library path patched) on both bundled data sets and compares with the golden vectors of the
reference run, so the documented binding cannot rot."""
import os
import re
import types

import numpy as np
import pytest
import torch

from helpers import ROOT, load_config

pytestmark = pytest.mark.gpu


def _stub_source():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    src = [b for b in blocks if "class AmdForward" in b]
    assert len(src) == 1
    return src[0].replace('"/path/to/lens_amd/liblens_amd.so"', repr(os.path.join(ROOT, "lens_amd", "liblens_amd.so")))


@pytest.mark.parametrize("tag", ["example", "sunset"])
def test_documented_ctypes_stub_reproduces_the_reference_counts(tag):
    ns = {}
    exec(compile(_stub_source(), "INTEGRATION.md", "exec"), ns)
    c, frames, w1, w2, gt, G = load_config(tag)
    # what the stub reads from a lens.run_model.LENS instance after load_model()
    lin = lambda w: types.SimpleNamespace(w=types.SimpleNamespace(weight=torch.from_numpy(w)))
    model = types.SimpleNamespace(feature_layer=lin(w1), output_layer=lin(w2), roi_dim=c["roi"], dims=c["dims"])
    fwd = ns["AmdForward"](model, 250)
    out = fwd(torch.from_numpy(frames).cuda()).cpu().numpy()
    assert np.array_equal(out.astype(np.int16), G["out_carry"])      # the reference never resets the state
