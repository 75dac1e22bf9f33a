"""Weights container of one LENS layer (host-side mirror of `SNNLayer`, lens/src/blitnet.py:41-125).

inference=True  (blitnet.py:59-64): a bias-free `nn.Linear` named `w` plus a threshold parameter `thr`
    [1, out] that the inference path loads but never uses.
inference=False (blitnet.py:65-125): the training-mode state -- thresholds, target firing rates, learning
    rates, excitatory / inhibitory sparse initial weights (`addWeights`, blitnet.py:127-168), the combined
    weight matrix and the two sign masks.  Construction makes the same torch / numpy RNG calls in the same
    order as the reference, so equal seeds give bit-identical initial state (tests/test_abi_host.py).
The attribute names match the reference so its state_dict keys (`<layer>.w.weight`, `<layer>.thr`) load and
save unchanged.  The STDP arithmetic itself (calc_stdp) runs in the HIP kernel k_train_layer.
"""
import numpy as np
import torch
import torch.nn as nn


class SNNLayer(nn.Module):
    def __init__(self, dims=(0, 0), thr_range=(0, 0), fire_rate=(0, 0), ip_rate=0, stdp_rate=0, const_inp=(0, 0),
                 p=(1, 1), spk_force=False, device=None, inference=False, args=None):
        super().__init__()
        self.device = device
        n_in, n_out = int(dims[0]), int(dims[1])
        if inference:
            self.w = nn.Linear(n_in, n_out, bias=False)
            self.thr = nn.Parameter(torch.zeros([1, n_out]).uniform_(float(thr_range[0]), float(thr_range[1])))
            return
        if np.isscalar(thr_range): thr_range = [thr_range, thr_range]
        if np.isscalar(fire_rate): fire_rate = [fire_rate, fire_rate]
        if np.isscalar(const_inp): const_inp = [const_inp, const_inp]
        self.x = torch.zeros([1, n_out])
        self.eta_ip = torch.tensor(ip_rate)
        self.eta_stdp = torch.tensor(stdp_rate)
        self.thr = nn.Parameter(torch.zeros([1, n_out]).uniform_(thr_range[0], thr_range[1]))
        self.fire_rate = torch.zeros([1, n_out]).uniform_(fire_rate[0], fire_rate[1])
        if not torch.all(self.fire_rate == 0).item():          # evenly spaced target rates (blitnet.py:83-87)
            fstep = (fire_rate[1] - fire_rate[0]) / n_out
            for i in range(n_out):
                self.fire_rate[:, i] = fire_rate[0] + fstep * (i + 1)
        self.have_rate = torch.any(self.fire_rate[:, 0] > 0.0)
        self.const_inp = torch.zeros([1, n_out]).uniform_(const_inp[0], const_inp[1])
        self.p = p
        self.dims = [n_in, n_out]
        self.spk_force = spk_force
        exc = nn.Linear(n_in, n_out, bias=False)                # consumes RNG like the reference's placeholders
        exc_w = self.addWeights(dims=[n_in, n_out], W_range=[0, 1], p=p[0])
        inh = nn.Linear(n_in, n_out, bias=False)
        inh_w = self.addWeights(dims=[n_in, n_out], W_range=[-1, 0], p=p[-1])
        del exc, inh
        self.w = nn.Linear(n_in, n_out, bias=False)
        self.w.weight = nn.Parameter(torch.add(exc_w, inh_w))
        self.havconnCombinedExc = self.w.weight > 0
        self.havconnCombinedInh = self.w.weight < 0

    @staticmethod
    def addWeights(W_range=(0, 0), p=0, dims=(0, 0), device=None):
        """Sparse signed initial weights (blitnet.py:127-168): N(mean, sd) from the range, wrong-signed entries
        zeroed, entries dropped with probability 1-p (numpy RNG), everything divided by the L1 norm of the LAST
        row (the reference normalises by `W[len(W)-1]`)."""
        nrow, ncol = int(dims[1]), int(dims[0])
        mean = (W_range[0] + W_range[1]) / 2.0
        sd = (W_range[1] - W_range[0]) / 6.0
        W = torch.empty(nrow, ncol).normal_(mean=mean, std=sd)
        if W_range[-1] != 0:
            W[W < 0] = 0.0
        else:
            W[W > 0] = 0.0
        drop = np.random.rand(nrow, ncol) > p
        if drop.any():
            W[torch.from_numpy(drop)] = 0.0
        nrm = torch.linalg.norm(W[len(W) - 1], ord=1, axis=0)
        nrm[nrm == 0.0] = 1.0
        return nn.Parameter(W / nrm)
