import sys, time, torch
sys.path.insert(0, '/root/repo')
from lens_amd.engine import seqmatch_topk
Q, P = 4096, 100000
c = torch.randint(0, 30, (Q, 100096), device="cuda").float()
for L in (1, 2, 3):
    seqmatch_topk(c, L, 25, n_places=P - L + 1); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): seqmatch_topk(c, L, 25, n_places=P - L + 1)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"L={L}: {dt*1e3:.3f} ms  {Q*P*4/dt/1e12:.2f} TB/s of the matrix once")
# what a plain streaming read of the same matrix reaches on this box (torch's reduction kernels), for comparison
for name, f in (("sum", lambda: c.sum()), ("max over places", lambda: c.max(dim=1)), ("copy", lambda: c.clone())):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): f()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"torch {name}: {dt*1e3:.3f} ms  {c.numel()*4/dt/1e12:.2f} TB/s read")
