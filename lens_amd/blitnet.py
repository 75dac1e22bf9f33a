"""Weights container of one LENS layer in inference mode.

Mirror of `SNNLayer(dims, ..., inference=True)` (lens/src/blitnet.py:59-64): a bias-free
`nn.Linear` named `w` plus a threshold parameter `thr` of shape [1, out] that the inference path
loads but never uses.  Keeping the same attribute names keeps the reference's state_dict keys
(`<layer>.w.weight`, `<layer>.thr`) loadable bit-exactly.  The training-mode branch of the
reference class (STDP rules) is outside this path.
"""
import torch
import torch.nn as nn


class SNNLayer(nn.Module):
    def __init__(self, dims=(0, 0), thr_range=(0, 0), device=None, inference=True, **unused):
        super().__init__()
        if not inference:
            raise NotImplementedError("lens_amd implements the inference container only "
                                      "(training kernels are a later row, SURVEY.md 8f-1)")
        self.device = device
        self.w = nn.Linear(int(dims[0]), int(dims[1]), bias=False)
        lo, hi = float(thr_range[0]), float(thr_range[1])
        self.thr = nn.Parameter(torch.zeros([1, int(dims[-1])]).uniform_(lo, hi))
