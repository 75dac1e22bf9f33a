"""Time per place of the output kernel for each LDS tile width that fits F (LENS_AMD_OUTPUT_PPL forces the width;
one process per width because the library reads the variable once).  Usage: python scripts/experiments/ppl_cost.py F P"""
import os
import subprocess
import sys

if len(sys.argv) == 4:
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
    from lens_amd.engine import LensEngine
    F, P, ppl = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    g = torch.Generator().manual_seed(1)
    dims = 10
    w1 = (torch.randn(F, dims * dims, generator=g) * 0.35).numpy()
    w1[torch.rand(F, dims * dims, generator=g).numpy() < 0.5] = 0
    w2 = (torch.randn(P, F, generator=g) * 0.1).numpy()
    Q = 2048
    frames = (torch.rand(Q, 80, 80, generator=g) < 0.1).to(torch.uint8) * torch.randint(1, 255, (Q, 80, 80), generator=g, dtype=torch.uint8)
    eng = LensEngine(w1, w2, 80, dims, 250, device="cuda", max_batch=Q)
    fr = frames.cuda()
    eng.forward_counts(fr, carry=False)
    spk = None
    from lens_amd import _lib
    from lens_amd.engine import _ptr, _stream
    out = torch.empty(Q, P, device="cuda")
    def run():
        _lib.check(_lib.lib().lens_output_counts(_ptr(eng.events), eng.cap, _ptr(eng.nwords), Q, _ptr(eng.w2t), eng.F, eng.P,
                                                  eng.ldw2, None, 0, 0, 0, _ptr(out), P, _stream()), "lens_output_counts")
    run(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): run()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    words = float(eng.nwords[:Q].float().mean())
    print(f"F={F} P={P} ppl={ppl} ev_scale={eng.ev_scale} words/query={words:.0f} {ms:.3f} ms  {ms * 1e6 / (Q * P * words / 2):.5f} ns per event-place")
else:
    F, P = sys.argv[1], sys.argv[2]
    for ppl in (1, 2, 3, 4):
        if int(F) * 64 * ppl * 4 > 160 * 1024 - 4096:
            continue
        env = dict(os.environ, LENS_AMD_OUTPUT_PPL=str(ppl))
        subprocess.run([sys.executable, __file__, F, P, str(ppl)], env=env, check=True)
