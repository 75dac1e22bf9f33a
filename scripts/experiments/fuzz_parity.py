"""Randomised parity fuzz (development aid): random layer shapes, weight scales, frame densities and state modes,
GPU path vs the CPU oracle, bit-exact.  Usage: fuzz_parity.py [n_cases] [seed]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from lens_amd.engine import LensEngine, seqmatch, seqmatch_topk
from lens_amd import _lib
from oracle import oracle as O

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
BIG = len(sys.argv) > 3 and sys.argv[3] == "big"     # larger Q / P / T: several query groups, many tiles, long streams
HOT = len(sys.argv) > 3 and sys.argv[3] == "hot"     # the shapes of the hand-scheduled loops: 65..128 inputs, 129..256 features
t0 = time.time()
n_l2 = n_shared = 0
for case in range(n_cases):
    dims = int(rng.integers(1, 12))
    k = int(rng.integers(1, 4))
    roi = dims * k
    N = dims * dims
    F = int(rng.choice([1, 3, 17, 64, 65, 130, 156, 157, 200, 256, 312, 313, 400, 624, 640, 900]))
    P = int(rng.choice([1, 2, 63, 64, 65, 127, 128, 129, 255, 256, 257, 500, 1000, 1300]))
    T = int(rng.choice([1, 2, 7, 16, 33, 100, 250]))
    Q = int(rng.integers(1, 20))
    if HOT:
        dims = int(rng.integers(9, 12)); roi = dims * k; N = dims * dims
        F = int(rng.choice([129, 130, 156, 157, 192, 193, 200, 255, 256]))
    if BIG:
        Q = int(rng.integers(20, 400)); P = int(rng.choice([300, 1000, 2500, 5000])); T = int(rng.choice([100, 250]))
        F = int(rng.choice([64, 156, 200, 312, 400])); dims = int(rng.integers(4, 11)); roi = dims * k; N = dims * dims
    w1s, w2s = rng.choice([0.05, 0.3, 1.0, 4.0]), rng.choice([0.01, 0.1, 0.5, 2.0])
    w1 = (rng.standard_normal((F, N)) * w1s).astype(np.float32)
    w1[rng.random((F, N)) < rng.choice([0.0, 0.5, 0.9])] = 0
    w2 = (rng.standard_normal((P, F)) * w2s).astype(np.float32)
    if rng.random() < 0.3:
        w2[rng.integers(0, P)] = np.float32(np.nextafter(np.float32(1), np.float32(0)))   # rows that land on 1 - 2^-24
    if rng.random() < 0.2:
        w1[rng.integers(0, F)] = 1.0e7                                                    # spike counts >= 2^23
    dens = rng.choice([1, 3, 30, 255])
    frames = rng.integers(0, dens + 1, size=(Q, roi, roi)).astype(np.uint8)
    frames[rng.random((Q, roi, roi)) < rng.choice([0.0, 0.7, 0.97])] = 0
    if rng.random() < 0.3:
        frames[rng.integers(0, Q)] = 0
    eng = LensEngine(w1, w2, roi, dims, T, device="cuda")
    U = O.rand_U(T, roi * roi)
    for carry in (False, True):
        eng.reset_state()
        out = eng.forward_counts(torch.from_numpy(frames).cuda(), carry=carry)
        ref = O.forward(frames, dims, U, w1, w2, carry=carry)
        if not np.array_equal(out.cpu().numpy(), ref, equal_nan=True):
            print("MISMATCH", dict(case=case, dims=dims, k=k, F=F, P=P, T=T, Q=Q, w1s=w1s, w2s=w2s, dens=dens, carry=carry))
            sys.exit(1)
    # several carried streams: contiguous runs of the batch, each with its own state (and a second call on that state)
    ns = int(rng.integers(2, 6))
    if Q >= ns:
        eng.reset_state(ns)
        bounds = [s * Q // ns for s in range(ns + 1)]
        states = [O.State(N, F, P) for _ in range(ns)]
        for rep in range(2):
            outs = eng.forward_counts(torch.from_numpy(frames).cuda(), carry=True).cpu().numpy()
            for si in range(ns):
                a, b = bounds[si], bounds[si + 1]
                refs = O.forward(frames[a:b], dims, U, w1, w2, carry=True, state=states[si])
                if not np.array_equal(outs[a:b], refs, equal_nan=True):
                    print("MISMATCH (streams)", dict(case=case, dims=dims, k=k, F=F, P=P, T=T, Q=Q, ns=ns, stream=si, rep=rep, w1s=w1s, w2s=w2s, dens=dens))
                    sys.exit(1)
        eng.reset_state()
    # the whole batch as S identical carried streams, enough of them that the 64-place tiles are read through L2
    # (k_output_carry_l2): every stream must repeat the single stream's counts (`ref` is the carried run's)
    lib = _lib.lib()
    S = next((n for n in (130, 260, 520) if lib.lens_output_reads_through_l2(F, P, 1, n) and
              lib.lens_output_tile_places(F, P, 1, n) == 64 and n * Q <= 8192), 0)
    if S:
        engs = LensEngine(w1, w2, roi, dims, T, device="cuda", max_batch=Q * S)
        engs.reset_state(S)
        outs = engs.forward_counts(torch.from_numpy(frames).cuda().repeat(S, 1, 1), carry=True).cpu().numpy().reshape(S, Q, P)
        if not np.array_equal(outs, np.broadcast_to(ref, (S, Q, P)), equal_nan=True):
            print("MISMATCH (through L2)", dict(case=case, dims=dims, k=k, F=F, P=P, T=T, Q=Q, S=S, w1s=w1s, w2s=w2s, dens=dens))
            sys.exit(1)
        n_l2 += 1
        del engs
    # ... and as 20 identical streams on the wide LDS tiles that carried streams share 16 at a time (layout forced, width
    # by turns: the cost model picks it only for many streams on large databases)
    os.environ["LENS_AMD_CARRY_SHARED"] = str(1 + case % 4)
    try:
        if lib.lens_output_layout(F, P, 1, 20) == 3 and 20 * Q <= 8192:
            engs = LensEngine(w1, w2, roi, dims, T, device="cuda", max_batch=Q * 20)
            engs.reset_state(20)
            outs = engs.forward_counts(torch.from_numpy(frames).cuda().repeat(20, 1, 1), carry=True).cpu().numpy().reshape(20, Q, P)
            if not np.array_equal(outs, np.broadcast_to(ref, (20, Q, P)), equal_nan=True):
                print("MISMATCH (shared tiles)", dict(case=case, dims=dims, k=k, F=F, P=P, T=T, Q=Q, w1s=w1s, w2s=w2s, dens=dens,
                                                        tile=engs.tile_places(True)))
                sys.exit(1)
            n_shared += 1
            del engs
    finally:
        os.environ.pop("LENS_AMD_CARRY_SHARED", None)
    L = int(rng.choice([1, 2, 3]))
    if Q >= L and P >= L and np.isfinite(ref).all():
        S = O.seqmatch(ref, L)
        assert np.array_equal(seqmatch(out, L).cpu().numpy().T, S), ("seqmatch", case)
        K = int(min(25, 64))
        gi, gv = seqmatch_topk(out, L, K)
        oi, ov = O.topk(S, K)
        if not (np.array_equal(gi.cpu().numpy(), oi) and np.array_equal(gv.cpu().numpy(), ov)):
            g = gi.cpu().numpy(); gvv = gv.cpu().numpy()
            bad = np.argwhere((g != oi) | (gvv != ov))
            r = int(bad[0][0])
            print("TOPK MISMATCH", dict(case=case, dims=dims, k=k, F=F, P=P, T=T, Q=Q, L=L, K=K, w1s=w1s, w2s=w2s, dens=dens), "row", r)
            print(" gpu idx", g[r].tolist()); print(" ref idx", oi[r].tolist()); print(" gpu val", gvv[r].tolist()); print(" ref val", ov[r].tolist())
            print(" S col", S[:, r].tolist()[:80])
            sys.exit(1)
    if case % 10 == 9 or BIG:        # (a big case takes the oracle a minute or more: keep the log moving)
        print(f"  {case + 1} cases, {time.time() - t0:.0f} s", flush=True)
print(f"{n_cases} cases ok in {time.time() - t0:.1f} s ({n_l2} of them also as many carried streams through L2, {n_shared} as 20 streams on shared tiles)")
