"""Pins the one unpinned assumption of the oracle -- the integrate-and-fire rule of sinabs' IAFSqueeze as
`sinabs.from_torch.from_model` builds it (lens/run_model.py:151-156) -- against the REAL sinabs, wherever that
package is importable.  It is not installed in the build or test images (no network), so this test is skipped
there; `oracle/lens_oracle.c:iaf_one` was transcribed from the public sinabs 2.x sources (from_torch.py,
layers/iaf.py, layers/functional/lif.py, activation/{spike_generation,reset_mechanism}.py) and this is the check
to run the day a sinabs installation is at hand.
"""
import ctypes

import numpy as np
import pytest

sinabs = pytest.importorskip("sinabs")


def _oracle_network(x, w1, w2, v1, v2):
    """Linear -> IAF -> Linear -> IAF over the T rows of x with the oracle's own primitives; state in v1, v2."""
    from oracle import oracle as O
    L = O.lib()
    fp = ctypes.POINTER(ctypes.c_float)
    p = lambda a: a.ctypes.data_as(fp)
    T, n = x.shape
    m, q = w1.shape[0], w2.shape[0]
    out = np.zeros((T, q), np.float32)
    for t in range(T):
        cur1, s1 = np.zeros(m, np.float32), np.zeros(m, np.float32)
        L.lens_oracle_linear(p(np.ascontiguousarray(x[t])), p(w1), p(cur1), n, m)
        L.lens_oracle_iaf_step(p(v1), p(cur1), p(s1), m)
        cur2, s2 = np.zeros(q, np.float32), np.zeros(q, np.float32)
        L.lens_oracle_linear(p(s1), p(w2), p(cur2), m, q)
        L.lens_oracle_iaf_step(p(v2), p(cur2), p(s2), q)
        out[t] = s2
    return out


def test_iaf_rule_equals_real_sinabs_from_model():
    import torch
    import torch.nn as nn
    from sinabs.from_torch import from_model
    rng = np.random.default_rng(0)
    T, n, m, q = 64, 12, 9, 7
    pred1 = np.nextafter(np.float32(1), np.float32(0))
    w1 = (rng.standard_normal((m, n)) * 0.7).astype(np.float32)
    w1[0, :] = 0; w1[0, 0] = pred1            # membrane lands on 1 - 2^-24: clip rounds it to 1.0, fires a step later
    w1[1, :] = 0; w1[1, 1] = 2.75             # multi-spike: 2 spikes, 0.75 left
    w1[2, :] = 0; w1[2, 2] = -3.5             # driven below min_v_mem = -1: clipped
    w1[3, :] = 0; w1[3, 3] = 1.0              # exactly at threshold
    w2 = (rng.standard_normal((q, m)) * 0.6).astype(np.float32)
    x = (rng.random((T, n)) < 0.3).astype(np.float32)
    x[5] = 0; x[6] = 0                        # input-free steps (the GPU stream drops those)
    model = nn.Sequential(nn.Linear(n, m, bias=False), nn.ReLU(), nn.Linear(m, q, bias=False))
    with torch.no_grad():
        model[0].weight.copy_(torch.from_numpy(w1))
        model[2].weight.copy_(torch.from_numpy(w2))
    snn = from_model(model, input_shape=(n,), num_timesteps=T, add_spiking_output=True)
    v1, v2 = np.zeros(m, np.float32), np.zeros(q, np.float32)
    with torch.no_grad():
        for call in range(2):                 # the second call starts from the first call's membrane state
            got = snn(torch.from_numpy(x)).reshape(T, q).numpy()
            ref = _oracle_network(x, w1, w2, v1, v2)
            bad = int((got != ref).sum())
            assert bad == 0, (call, getattr(sinabs, "__version__", "?"))
