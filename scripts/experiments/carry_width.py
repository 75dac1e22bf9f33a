"""Carried-state output layer alone: time per query for each LDS tile width (run once per width, the width is forced
through LENS_AMD_OUTPUT_PPL in a child process).  python carry_width.py [P] [streams]"""
import os, subprocess, sys
P = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1
if os.environ.get("CARRY_WIDTH_CHILD"):
    import time, torch
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
    import bench
    from lens_amd.engine import LensEngine
    from lens_amd.synthetic import synthetic_frames, synthetic_database
    w1, w2p = bench.pretrained()
    Q = 64 * S
    eng = LensEngine(w1, synthetic_database(w2p, P), bench.ROI, bench.DIMS, bench.T_STEPS, device="cuda", max_batch=Q)
    eng.reset_state(S)
    fr = synthetic_frames(Q, bench.ROI, seed=1234).cuda()
    eng.feature_events(fr, carry=True)
    def timed(f, n=3):
        f(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): f()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    ms = timed(lambda: eng.output_counts(Q, carry=True))
    print(f"P={P} streams={S} tile={eng.tile_places(True)} places: output layer {ms:.3f} ms per {Q} queries = {ms * 1e3 / (Q // S):.1f} us per query of a stream", flush=True)
else:
    for ppl in ("0", "1", "2", "3"):
        env = dict(os.environ, CARRY_WIDTH_CHILD="1")
        if ppl != "0":
            env["LENS_AMD_OUTPUT_PPL"] = ppl
        print("forced ppl", ppl, end=": ", flush=True)
        subprocess.run([sys.executable, __file__, str(P), str(S)], env=env, check=True, timeout=300)
