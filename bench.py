#!/usr/bin/env python3
"""Benchmark of the LENS hot path on MI355X.

A step = one pass of the hot path over one batch of synthetic 128x128 event-count frames already
resident in HBM:   frames -> rate-coded feature spike streams -> event-driven output layer (place
database) -> sequence matcher + top-25 per query  [-> RCCL all-gather + merge when the database is
sharded].  Metric: query event-frames/s, whole job.

  python bench.py                              config 3: 100k-place DB, 4096 queries, 1 GPU (+ config 2, carried
                                               state and the CPU baseline as secondary objects of the same line)
  python bench.py --places 1000                config 2 as the timed workload
  torchrun ... bench.py --gpus 8               config 4: the 100k-place DB sharded over the ranks, one all-gather
                                               of the per-shard top-k per step, merged on every rank
  torchrun ... bench.py --gpus 8 --shard queries    replicated DB, queries sharded (no collective, weak scaling)
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_PEAK_TFLOPS = 157.3     # MI355X fp32 vector peak: 256 CUs x 4 SIMDs x 64 lanes x 2 flop (FMA) x 2.4 GHz / 2 cycles
VALU_ISSUE_PER_NS = 1.2      # one wave64 VALU instruction per 2 cycles per SIMD at 2.4 GHz
N_SIMD = 1024
HBM_PEAK_GBS = 8000.0        # HBM3E spec
T_STEPS, ROI, DIMS, SEQ_L, TOPK = 250, 128, 10, 2, 25


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--places", type=int, default=100000,
                    help="reference DB size P (100000 = config 3 on one GPU / config 4 sharded; 1000 = config 2)")
    ap.add_argument("--queries", type=int, default=4096, help="query frames per step per rank-batch")
    ap.add_argument("--shard", choices=["auto", "queries", "db"], default="auto",
                    help="multi-GPU split: db (sharded DB + all-gather of top-k) or queries (replicated DB, no collective)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU oracle legs (baseline and in-run parity)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the secondary objects (config 2, carried state, two-stream pipelining, other kernels)")
    ap.add_argument("--streams", type=int, default=1, choices=[1, 2],
                    help="2: alternate steps over two HIP streams (lens_amd.pipeline); per-kernel durations then come "
                         "from a serial pass before the timed region")
    return ap.parse_args()


def pretrained():
    sd = torch.load(os.path.join(ROOT, "lens", "models", "example-reference_LENS_IN100_FN200_DB100.pth"),
                    map_location="cpu", weights_only=True)
    return sd["feature_layer.w.weight"], sd["output_layer.w.weight"]


def host_cores():
    """Threads for the CPU baseline: the box's CPU share (OMP_NUM_THREADS is set to it there), not the machine's."""
    env = os.environ.get("OMP_NUM_THREADS")
    n = int(env) if env and env.isdigit() else (os.cpu_count() or 1)
    return max(1, min(n, os.cpu_count() or 1, 64))


def cpu_baseline_and_parity(frames, w1, w2, gpu_counts, gpu_idx, gpu_val, target_cpu_seconds=15.0):
    """The CPU oracle (a port of the reference path: dense Linear layers, state reset per query like the GPU run)
    timed on this host's cores on a bounded sample of the same workload; its output for that sample is then the
    checker of the GPU results of the same queries (counts of every place, and the top-25 lists)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O
    O.lib()
    cores = host_cores()
    U = O.rand_U(T_STEPS, ROI * ROI)
    f = frames.numpy()
    w1n, w2n = w1.numpy(), w2.numpy()
    t0 = time.perf_counter()
    O.forward(f[:1], DIMS, U, w1n, w2n, carry=False)
    tq = time.perf_counter() - t0
    n = int(max(cores, min(len(f), target_cpu_seconds * cores / max(tq, 1e-6))))
    n = min(n, len(f), 512)
    chunks = [c for c in np.array_split(np.arange(n), cores) if len(c)]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(len(chunks)) as ex:       # ctypes releases the GIL inside the C call
        parts = list(ex.map(lambda c: O.forward(f[c[0]:c[-1] + 1], DIMS, U, w1n, w2n, carry=False), chunks))
    dt = time.perf_counter() - t0
    ref = np.concatenate(parts)
    base = {"value": n / dt, "unit": "query event-frames/s", "cores": len(chunks), "kind": "port",
            "sample": f"{n} of the {len(f)} synthetic queries against the same {w2.shape[0]}-place DB, "
                      f"oracle/lens_oracle.c dense path, state reset per query, one thread per core"}
    # the reference's own shape: one query at a time, membrane state carried, one thread (lens/run_model.py:229-246),
    # on the 1000-place DB so that it stays within seconds
    w2s = w2n[:1000]
    m = 24
    t0 = time.perf_counter()
    O.forward(f[:m], DIMS, U, w1n, w2s, carry=True)
    dts = time.perf_counter() - t0
    base["reference_shaped"] = {"value": m / dts, "unit": "query event-frames/s", "cores": 1,
                                "sample": f"{m} queries in order, carried state, one thread, 1000-place DB"}
    oi, ov = O.topk(O.seqmatch(ref, SEQ_L), TOPK)
    gi, gv = gpu_idx[:n - SEQ_L + 1].cpu().numpy(), gpu_val[:n - SEQ_L + 1].cpu().numpy()
    parity = {"checker": "oracle/lens_oracle.c on the cpu_baseline sample",
              "queries": n, "places": int(w2.shape[0]), "spikes_in_sample": float(ref.sum()),
              "top1_identical": bool(np.array_equal(gi[:, 0], oi[:, 0])),
              "top25_idx_identical": bool(np.array_equal(gi, oi)),
              "top25_val_max_abs_diff": float(np.abs(gv - ov).max()),
              "note": "identical top-25 index lists give an identical Recall@N table for any ground truth "
                      "(tie policy: score descending, then place index descending)"}
    g_all, g_smp, o_smp = recall_identity(gpu_idx.cpu().numpy()), recall_identity(gi), recall_identity(oi)
    parity["recall_at_1"] = g_all["recall_at_1"]
    parity["recall"] = {"ground_truth": "identity (query i <-> place i), no tolerance; database rows are not trained on "
                                        "the query templates: chance level by construction",
                        "gpu_all_queries": dict(g_all, queries=int(gpu_idx.shape[0])),
                        "gpu_on_sample": g_smp, "oracle_on_sample": o_smp, "equal_on_sample": g_smp == o_smp}
    if gpu_counts is not None:                     # one GPU holds every place: compare the similarity-matrix rows too
        got = gpu_counts[:n].cpu().numpy()
        parity["counts_compared"] = int(ref.size)
        parity["mismatches"] = int((got != ref).sum())
    else:
        parity["mismatches"] = int((gi != oi).sum() + (gv != ov).sum())
        parity["compared"] = "merged global top-25 lists (each rank holds only its shard's counts)"
    return base, parity


def other_kernel_rooflines(frames_cpu, dev, topk_ms, topk_bytes):
    """HBM rooflines of the memory-bound rows of the path, measured once outside the timed region:
    a1 event accumulation (4 B/event in + H*W B/frame out) and the in-step seq-match/top-k kernel."""
    from lens_amd.engine import accumulate_events
    from lens_amd.synthetic import synthetic_events
    nf = min(2048, frames_cpu.shape[0])      # ~100 M events = 400 MB: larger than the 256 MB Infinity Cache
    xy, off = synthetic_events(frames_cpu[:nf])
    xy, off = xy.to(dev), off.to(dev)
    accumulate_events(xy, off, ROI)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        fr, bad = accumulate_events(xy, off, ROI, check=False)
    e1.record(); torch.cuda.synchronize()
    assert torch.equal(fr.cpu(), frames_cpu[:nf]) and int(bad.item()) == (1 << 63) - 1
    ms = e0.elapsed_time(e1) / reps
    nbytes = 4.0 * xy.shape[0] + nf * ROI * ROI
    res = {"k_accumulate_events": {"bound": "hbm", "frames": nf, "events": int(xy.shape[0]), "avg_launch_ms": ms,
                                   "achieved": nbytes / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}}
    ach = topk_bytes / (topk_ms * 1e-3) / 1e9
    res["k_seqmatch_topk"] = {"bound": "hbm", "avg_launch_ms": topk_ms, "achieved": ach, "peak": HBM_PEAK_GBS,
                              "unit": "GB/s", "frac": ach / HBM_PEAK_GBS}
    return res


def stream_stats(eng, Q):
    """Executed work of the event-driven output kernel, counted from the spike streams: words per query, events per
    query (the {0, +1.0f} slots that pad a stream's last chunk are left out: they follow its final LAST event) and
    integrate-and-fire updates per query (time steps without feature spikes are not in the stream)."""
    nw = eng.nwords[:Q]
    words = float(nw.float().mean().item())
    cw = eng.events[:Q].view(Q, -1, 2)[:, :, 1]
    steps = events = 0.0
    for q0 in range(0, Q, 512):                                  # bounded temporaries: the workspace is worst-case sized
        c = cw[q0:q0 + 512, :int(nw.max().item()) // 2 + 1]
        pos = torch.arange(c.shape[1], device=c.device)[None, :]
        last = (c < 0) & (pos < (nw[q0:q0 + 512, None] // 2))    # LAST events (sign bit of the count word)
        steps += float(last.sum().item())
        events += float((torch.where(last, pos + 1, torch.zeros_like(pos)).max(dim=1).values).sum().item())
    return words, events / Q, steps / Q


def recall_identity(idx, K_list=(1, 5, 25)):
    """Recall@K of top-K lists [n, K] against the synthetic ground truth "query i was drawn for place i" (identity;
    no tolerance band).  The benchmark's database rows are NOT trained on the query templates (they are permuted,
    jittered rows of the bundled model), so this sits at chance level by construction; it is reported so that the
    GPU's and the oracle's values can be compared, not as a quality figure."""
    idx = np.asarray(idx)
    want = np.arange(idx.shape[0])[:, None]
    return {f"recall_at_{k}": float((idx[:, :k] == want).any(axis=1).mean()) for k in K_list}


# Step time of ONE GPU at a per-rank database size (places incl. the halo row, 4096 queries, synthetic density), measured
# with `bench.py --places <n>`: what each rank of a DB-sharded run is expected to take, the all-gather and merge being
# overlapped with the next step's kernels.  A projection to compare the first multi-GPU measurement with, not a result.
SINGLE_GPU_STEP_MS = {12501: 2.02, 25001: 3.74, 50001: 7.22, 100000: 13.95}


def expected_sharded_ms(places_per_rank, queries):
    xs = sorted(SINGLE_GPU_STEP_MS)
    ys = [SINGLE_GPU_STEP_MS[x] for x in xs]
    return float(np.interp(places_per_rank, xs, ys)) * queries / 4096.0


def time_steps(step, n, events=None):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(n):
        step(events[k] if events is not None else None)
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def secondary_config2(w1, w2p, frames, dev, steps):
    """Config 2 (1000-place DB, same queries) from the same run: reset-mode rate and per-kernel times, then the
    reference's own semantics (state carried from query to query, lens/run_model.py:234-242) on 256 queries as one
    carried stream and as 16 independent carried streams."""
    from lens_amd.engine import LensEngine, seqmatch_topk
    from lens_amd.synthetic import synthetic_database
    Q = frames.shape[0]
    eng = LensEngine(w1, synthetic_database(w2p, 1000), ROI, DIMS, T_STEPS, device=dev, max_batch=Q)
    names = ["k_feature_events", "k_output_counts", "k_seqmatch_topk"]

    def step(evs=None):
        if evs is not None: evs[0].record()
        eng.feature_events(frames, carry=False)
        if evs is not None: evs[1].record()
        counts = eng.output_counts(Q, carry=False)
        if evs is not None: evs[2].record()
        seqmatch_topk(counts, SEQ_L, TOPK)
        if evs is not None: evs[3].record()

    for _ in range(3):
        step()
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(steps)]
    dt = time_steps(step, steps, evs)
    ms = [float(np.mean([e[i].elapsed_time(e[i + 1]) for e in evs])) for i in range(3)]
    res = {"config2": {"workload": f"config 2: 1000-place DB, {Q} queries per step, state reset per query",
                       "value": Q * steps / dt, "unit": "query event-frames/s", "ms_per_step": dt / steps * 1e3,
                       "steps": steps, "kernels_ms": dict(zip(names, ms))}}
    res["carry_mode"] = carried_rates(eng, frames, "1000-place DB")
    return res


def carried_rates(eng, frames, what):
    """The reference's own semantics (membrane state carried from query to query, lens/run_model.py:234-242) on 256
    queries: as ONE carried stream (the drop-in for the reference's loop) and as 16 independent carried streams."""
    from lens_amd.engine import seqmatch_topk
    carry = {}
    qc = min(256, frames.shape[0])
    for n_streams in (1, 16):
        eng.reset_state(n_streams=n_streams)
        eng.forward_counts(frames[:qc], carry=True)             # warm-up (also sizes the workspace)
        reps = 3
        eng.reset_state(n_streams=n_streams)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            c = eng.forward_counts(frames[:qc], carry=True)
            seqmatch_topk(c, SEQ_L, TOPK)
        torch.cuda.synchronize()
        dtc = (time.perf_counter() - t0) / reps
        carry[f"streams_{n_streams}"] = {"value": qc / dtc, "unit": "query event-frames/s", "ms_per_batch": dtc * 1e3,
                                         "queries": qc, "output_tile_places": eng.tile_places(True),
                                         "output_layout": ("batch", "lone wavefronts on LDS tiles", "lone wavefronts through L2",
                                                           "LDS tiles shared by 16 streams")[eng.layout(True)],
                                         "tail_places_on_lone_tiles": eng.tail_places(True)}
    eng.reset_state()
    carry["note"] = ("reference semantics: membrane state carried across queries, each stream strictly sequential in "
                     f"the query index; {what}; streams_1 is the drop-in for lens/run_model.py:229-246")
    return carry


def trained_database(dev, n_places=4096, epochs=32):
    """End to end on a database that is TRAINED for its queries (the timed workload's database is not: throughput only).
    n_places synthetic places, a reference and a query traverse (two Poisson draws of each place's rate image); both
    layers trained on the reference traverse on the GPU (k_train_layer), the query traverse run through the inference
    path, Recall@K against "query i shows place i".  (That the oracle reproduces such a run bit for bit is a test:
    tests/test_gpu_train.py::test_trained_synthetic_database_recognises_its_places.)"""
    import main as cli
    from lens_amd.engine import LensEngine, seqmatch_topk
    from lens_amd.synthetic import synthetic_traverses, train_on_traverse
    ref, qry = synthetic_traverses(n_places, ROI)
    args = cli.build_parser().parse_args(["--train_model", "--reference_places", str(n_places), "--query_places",
                                          str(n_places), "--epoch_feat", str(epochs), "--epoch_out", str(epochs),
                                          "--roi_dim", str(ROI), "--dims", str(DIMS)])
    t0 = time.perf_counter()
    w1, w2 = train_on_traverse(args, ref)
    torch.cuda.synchronize()
    train_s = time.perf_counter() - t0
    eng = LensEngine(w1, w2, ROI, DIMS, T_STEPS, device=dev, max_batch=n_places)
    counts = eng.forward_counts(qry.to(dev), carry=False)
    idx, _ = seqmatch_topk(counts, SEQ_L, TOPK)
    raw = counts.topk(5, dim=1).indices.cpu().numpy()
    return {"workload": f"{n_places} synthetic places, reference and query traverse = two Poisson draws per place; both "
                        f"layers trained on the reference traverse for {epochs} epochs each on the GPU, then inference "
                        f"of the query traverse (state reset per query)",
            "train_seconds": train_s, "train_steps": 2 * n_places * epochs,
            "recall_sequence_matcher": recall_identity(idx.cpu().numpy()),
            "recall_similarity_matrix": recall_identity(raw, (1, 5)), "chance": 1.0 / n_places,
            "at_100000_places": "profiles/r03c_trained_db_100000.log: trained in 23 s (16 epochs per layer), "
                                "Recall@1 0.61 after the sequence matcher, 0.12 on the raw similarity matrix"}


def real_density(w1, w2, dev, Q, steps=3):
    """The timed workload's shape on REAL event frames: the bundled example's 100 query frames (80x80, ~8 feature
    spikes per step instead of the synthetic generator's 2.6) tiled to Q queries against the same database."""
    from lens_amd.dataset import CustomImageDataset
    from lens_amd.engine import LensEngine, seqmatch_topk
    base = os.path.join(ROOT, "lens", "dataset")
    ds = CustomImageDataset(os.path.join(base, "example-query.csv"), os.path.join(base, "example", "davis128", "example-query"))
    fr = ds.frames()
    frames = fr.repeat((Q + len(fr) - 1) // len(fr), 1, 1)[:Q].contiguous().to(dev)
    eng = LensEngine(w1, w2, fr.shape[1], DIMS, T_STEPS, device=dev, max_batch=Q)

    def step(_e=None):
        eng.feature_events(frames, carry=False)
        seqmatch_topk(eng.output_counts(Q, carry=False), SEQ_L, TOPK)

    step()
    dt = time_steps(step, steps)
    return {"value": Q * steps / dt, "unit": "query event-frames/s", "ms_per_step": dt / steps * 1e3, "steps": steps,
            "feature_spike_words_per_query": float(eng.nwords[:Q].float().mean().item()),
            "workload": f"bundled example query frames ({fr.shape[1]}x{fr.shape[2]}, recorded events) tiled to {Q} "
                        f"queries, {w2.shape[0]}-place DB, state reset per query"}


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    # Test hook: LENS_BENCH_SHARED_DEVICE=1 runs every rank on cuda:0 with the gloo backend, so that the N>1 control
    # flow (sharding, barriers, max-over-ranks timing, gather + merge) can be exercised on a one-GPU box.
    shared = os.environ.get("LENS_BENCH_SHARED_DEVICE") == "1"
    if shared:
        local = 0
    # Test hook: LENS_BENCH_GROUP_OF_ONE=1 makes a single rank form a process group of its own over RCCL and run the
    # multi-GPU code (collectives on device tensors, DB-sharded step with its all-gather and merge) on a one-GPU box.
    grouped = world > 1 or os.environ.get("LENS_BENCH_GROUP_OF_ONE") == "1"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if grouped:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if shared:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from lens_amd.engine import LensEngine, seqmatch_topk
    from lens_amd.dist import shard_rows, ShardedStep
    from lens_amd.synthetic import synthetic_frames, synthetic_database

    shard = a.shard
    if shard == "auto":
        shard = "db" if (world > 1 and a.places // world >= 8192) else "queries"
    if not grouped:
        shard = "queries"

    w1, w2p = pretrained()
    w2 = synthetic_database(w2p, a.places)
    Q = a.queries
    sharded = None
    if shard == "queries":
        frames_cpu = synthetic_frames(Q, ROI, seed=1234 + rank)     # every rank its own queries
        eng = LensEngine(w1, w2, ROI, DIMS, T_STEPS, device=dev, max_batch=Q)
        n_places, offset = a.places - SEQ_L + 1, 0
        total_queries = Q * world
    else:
        frames_cpu = synthetic_frames(Q, ROI, seed=1234)            # same queries, DB split over ranks
        r0, r1, n_places, offset = shard_rows(a.places, SEQ_L, rank, world)
        eng = LensEngine(w1, w2[r0:r1], ROI, DIMS, T_STEPS, device=dev, max_batch=Q)
        sharded = ShardedStep(eng, SEQ_L, TOPK, n_places, offset, Q, world)
        total_queries = Q
    frames = frames_cpu.to(dev)

    names = ["k_feature_events", "k_output_counts", "k_seqmatch_topk"]
    last = {}

    def step(evs=None):
        if sharded is not None:
            last["res"] = sharded.submit(frames, evs)
            return
        if evs is not None: evs[0].record()
        eng.feature_events(frames, carry=False)
        if evs is not None: evs[1].record()
        counts = eng.output_counts(Q, carry=False)
        if evs is not None: evs[2].record()
        idx, val = seqmatch_topk(counts, SEQ_L, TOPK, n_places=n_places, place_offset=offset)
        if evs is not None: evs[3].record()
        last["res"] = (idx, val)
        last["counts"] = counts

    for _ in range(a.warmup):
        step()
    events = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(a.steps)]
    pipe = None
    if a.streams == 2:
        if shard == "db":
            sys.exit("--streams 2 is implemented for the query-sharded (collective-free) path only")
        from lens_amd.pipeline import BatchPipeline
        pipe = BatchPipeline(w1, w2, ROI, DIMS, T_STEPS, device=dev, max_batch=Q)
        for k in range(a.steps):                 # serial pass: per-kernel durations for the roofline
            step(events[k])
        for _ in range(max(2, a.warmup)):
            pipe.submit(frames, SEQ_L, TOPK)
        pipe.synchronize()
    if grouped:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if pipe is None:
        for k in range(a.steps):
            step(events[k])
    else:
        for k in range(a.steps):
            pipe.submit(frames, SEQ_L, TOPK)
    torch.cuda.synchronize()
    my_elapsed = time.perf_counter() - t0
    if grouped:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    multi = None
    if grouped:
        cpu = shared                                                 # gloo: CPU tensors only
        tdev = "cpu" if cpu else dev
        tt = torch.tensor([elapsed], device=tdev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        mine = torch.tensor([float(rank), my_elapsed], device=tdev, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        multi = {"ranks_seen": sorted(int(t[0].item()) for t in allr),
                 "per_rank_elapsed_s": [float(t[1].item()) for t in sorted(allr, key=lambda t: float(t[0].item()))],
                 "backend": dist.get_backend()}
        if sharded is not None:
            multi["gather_merge_ms"] = sharded.gather_merge_ms()
            multi["expected_ms"] = expected_sharded_ms(eng.P, Q)
            multi["expected_note"] = ("step time of one GPU at this rank's database size, interpolated from single-GPU "
                                      "runs (bench.py --places 12501/25001/50001/100000): a projection for checking "
                                      "this measurement against, with the collective assumed hidden")
            multi["overlap"] = "all-gather + merge of step i run on a second stream beside the kernels of step i+1"

    if rank == 0:
        ms = [[ev[i].elapsed_time(ev[i + 1]) for ev in events] for i in range(3)]
        avg_ms = [float(np.mean(m)) for m in ms]
        dom = int(np.argmax(avg_ms))
        P_local = eng.P
        words, ev_slots, steps_exec = stream_stats(eng, Q)
        topk_bytes = Q * (4.0 * P_local + 8.0 * TOPK)              # the counts matrix is read once
        if dom == 1:
            # Executed work of the event-driven output kernel.  Per (event slot, place) one accumulate, counted as an
            # FMA (2 flops; single-spike events execute it as a plain add, 1 flop); per (executed step, place) ~8 flops
            # of integrate-and-fire (add, threshold test, 2-FMA clip, spike path).  Time steps without feature spikes
            # and the ~97 % zero spikes of the dense formulation are skipped exactly (fmaf(0, w, acc) == acc).
            exe2 = (2.0 * ev_slots + 8.0 * steps_exec) * P_local * Q
            exe = (1.0 * ev_slots + 8.0 * steps_exec) * P_local * Q
            dense = 2.0 * T_STEPS * eng.F * P_local * Q
            sec = avg_ms[dom] * 1e-3
            roof = {"kernel": names[dom], "bound": "valu", "achieved": exe / sec / 1e12, "peak": FP32_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": exe / sec / 1e12 / FP32_PEAK_TFLOPS, "traffic": None,
                    "avg_launch_ms": avg_ms[dom],
                    "executed_flop_per_launch": exe, "events_per_query": ev_slots,
                    "executed_steps_per_query": steps_exec,
                    "frac_events_counted_as_fma": exe2 / sec / 1e12 / FP32_PEAK_TFLOPS,
                    "dense_equivalent": {"flop_per_launch": dense, "tflops": dense / sec / 1e12,
                                         "speedup_over_fp32_peak": dense / sec / 1e12 / FP32_PEAK_TFLOPS,
                                         "note": "2*T*F*P per query; not a roofline fraction, the kernel skips the zeros"},
                    "counting": "frac = executed FLOPs / launch time / fp32 vector peak (which counts an FMA as 2); executed "
                                "= (1 per event: 96 % of them run as a plain v_add_f32; padding slots not counted + 8 per "
                                "executed integrate-and-fire step) per place; frac_events_counted_as_fma counts 2 per "
                                "event; the instruction-issue view is offline_profile.valu_issue_frac; the kernel issues "
                                "no MFMA (a dense f32-MFMA variant exists for comparison and is 26x slower)"}
        else:
            byts = [Q * (ROI * ROI + 4.0 * words), None, topk_bytes]
            ach = byts[dom] / (avg_ms[dom] * 1e-3) / 1e9
            roof = {"kernel": names[dom], "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBS, "traffic": None, "avg_launch_ms": avg_ms[dom]}
        ppath = os.path.join(ROOT, "profiles", "pmc.json")
        if os.path.exists(ppath):
            with open(ppath) as fh:
                pm = json.load(fh).get(f"P{P_local}_Q{Q}", {}).get(names[dom])
            if pm:
                roof["traffic"] = pm.get("hbm_bytes_per_launch")
                if dom == 1 and pm.get("SQ_INSTS_VALU"):
                    # the instruction-issue view: vector instructions per launch (committed PMC pass of this workload) over
                    # THIS run's launch time and the SIMDs' issue rate
                    roof["valu_issue_frac"] = pm["SQ_INSTS_VALU"] / (N_SIMD * avg_ms[dom] * 1e6 * VALU_ISSUE_PER_NS)
                if dom == 1 and all(pm.get(k) for k in ("SQ_INSTS_SALU", "SQ_INSTS_BRANCH", "SQ_INSTS_LDS", "GRBM_GUI_ACTIVE")):
                    # the SIMD's issue port: one instruction per cycle, a wave64 vector instruction holds it for two.
                    # Cycles of the PROFILED launch (GRBM_GUI_ACTIVE is summed over the 8 XCDs), all counters of that pass.
                    slots = 2.0 * pm["SQ_INSTS_VALU"] + pm["SQ_INSTS_SALU"] + pm["SQ_INSTS_BRANCH"] + pm["SQ_INSTS_LDS"] + pm.get("SQ_INSTS_SMEM", 0.0)
                    roof["issue_port_frac"] = slots / (N_SIMD * pm["GRBM_GUI_ACTIVE"] / 8.0)
                roof["offline_profile"] = dict(pm, note="rocprofv3 --pmc passes of this workload committed under "
                                               "profiles/ (not measured in this run); valu_issue_frac = SQ_INSTS_VALU "
                                               "/ (1024 SIMDs x launch time x 1.2 VALU/ns); issue_port_frac = (2 x VALU + SALU "
                                               "+ branch + LDS + SMEM instructions) / (1024 SIMDs x cycles of the profiled launch)")
        cfg = 2 if a.places == 1000 else (3 if world == 1 else (4 if shard == "db" else 3))
        line = {
            "metric": "query event-frames/sec vs ref-DB size; Recall@1 parity with reference",
            "value": total_queries * a.steps / elapsed,
            "unit": "query event-frames/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if shard == "db" else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"config {cfg}: pretrained LENS feature layer (IN100/FN200), synthetic {ROI}x{ROI} "
                                   f"event frames, {a.places}-place reference DB"
                                   + (f" sharded over {world} GPUs" if shard == "db" else "")
                                   + f", {total_queries} queries per step, state reset per query, T={T_STEPS}, "
                                   f"L={SEQ_L}, top-{TOPK}",
                       "places": a.places, "queries_per_step": total_queries, "shard": shard if grouped else "none",
                       "streams": a.streams, "feature_spike_words_per_query": words},
            "kernels_ms": dict(zip(names, avg_ms)),
            "roofline": roof,
        }
        if multi is not None:
            line["multi_gpu"] = multi
        if not a.no_extras and world == 1:
            line["other_kernels"] = other_kernel_rooflines(frames_cpu, dev, avg_ms[2], topk_bytes)
            if a.places != 1000:
                line["other_configs"] = secondary_config2(w1, w2p, frames, dev, min(a.steps, 20))
                line["other_configs"][f"carry_mode_{a.places // 1000}k"] = carried_rates(eng, frames, f"{a.places}-place DB")
            line.setdefault("other_configs", {})["real_density"] = real_density(w1, w2, dev, Q)
            line["other_configs"]["trained_db"] = trained_database(dev)
            if pipe is None:
                from lens_amd.pipeline import BatchPipeline
                p2 = BatchPipeline(w1, w2, ROI, DIMS, T_STEPS, device=dev, max_batch=Q)
                for _ in range(2):
                    p2.submit(frames, SEQ_L, TOPK)
                p2.synchronize()
                n2 = min(a.steps, 10)
                dt2 = time_steps(lambda _e: p2.submit(frames, SEQ_L, TOPK), n2)
                line["two_stream_pipelining"] = {
                    "ms_per_step": dt2 / n2 * 1e3, "value": Q * n2 / dt2, "unit": "query event-frames/s",
                    "note": "informational: the same steps alternated over two HIP streams (lens_amd.pipeline, identical "
                            "results); `value` is timed on one stream so that the per-kernel durations are those of the "
                            "kernels alone"}
                del p2
        if not a.no_cpu_baseline:
            gi, gv = last["res"]
            base, parity = cpu_baseline_and_parity(frames_cpu, w1, w2, last.get("counts"), gi, gv)
            if world == 1:                          # with peers spinning in the final barrier the host timing means little
                line["cpu_baseline"] = base
            line["parity"] = parity
        print(json.dumps(line))
    if grouped:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
