"""Training-step throughput of k_train_layer: steps/s for the feature layer and for the
output layer at P places, with the algorithmic HBM/ALU figures (development benchmark)."""
import argparse, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lens_amd import engine as E
from lens_amd.blitnet import SNNLayer

ap = argparse.ArgumentParser()
ap.add_argument("--places", type=int, default=100000)
ap.add_argument("--steps", type=int, default=2000)
ap.add_argument("--samples", type=int, default=1000)
a = ap.parse_args()
torch.manual_seed(0); np.random.seed(0)
dev = torch.device("cuda")
S, T = a.samples, a.steps
for name, n_in, n_out, spk in (("feature", 100, 200, False), ("output", 200, a.places, True),
                               ("feature", 400, 800, False), ("output", 800, 10000, True)):      # --dims 20: rows over 4 / 8 lanes
    L = SNNLayer(dims=[n_in, n_out], thr_range=[0, 0.5], fire_rate=[0.4, 0.6], ip_rate=0.02, stdp_rate=0.01,
                 p=[0.35, 0.75] if not spk else [1.0, 1.0], spk_force=spk)
    ld = E.train_in_pad(n_in)
    table = E.pad_table((torch.rand(S, n_in) * (torch.rand(S, n_in) < 0.5)).float().to(dev), ld)
    mpre = E.pad_table(table[:, :n_in] / 0.5, ld) if spk else None
    order = torch.randint(0, S, (T,), dtype=torch.int32, device=dev)
    fi = torch.randint(0, n_out, (T,), dtype=torch.int32, device=dev) if spk else None
    eta_ip = torch.full((T,), 0.02, device=dev); eta_st = torch.full((T,), 0.01, device=dev)
    w = L.w.weight.detach().to(dev).contiguous(); thr = L.thr.detach().reshape(-1).to(dev).contiguous()
    args = (w, thr, L.havconnCombinedExc.to(torch.uint8).to(dev), L.havconnCombinedInh.to(torch.uint8).to(dev),
            L.fire_rate.reshape(-1).to(dev), True, table, order, eta_ip, eta_st)
    E.train_layer(*args, force_idx=fi, mpre_table=mpre)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    E.train_layer(*args, force_idx=fi, mpre_table=mpre)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    elems = n_in * n_out
    print(f"{name:8s} layer [{n_out} x {n_in}]  {T} steps in {dt*1e3:.1f} ms -> {T/dt:.0f} steps/s; "
          f"{elems*T/dt/1e9:.1f} G weight-updates/s; would be {2*4*elems*T/dt/1e12:.2f} TB/s if W were streamed from HBM each step")
