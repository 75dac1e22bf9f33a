#!/usr/bin/env python3
"""Copy the judged summaries of a scripts/profile.sh run from gpurun_out/ into profiles/ and refresh
profiles/pmc.json: per kernel and workload, the rocprofv3 kernel-trace average launch time, the instruction
counters and the HBM bytes per launch from the FETCH_SIZE / WRITE_SIZE PMC passes (bench.py quotes these as its
offline profile).  Usage: collect_profiles.py <round-tag> <gpurun_out/prof_dir> <config-key e.g. P100000_Q4096>"""
import csv, glob, json, os, shutil, subprocess, sys, collections
tag, src, key = sys.argv[1], sys.argv[2].rstrip("/"), sys.argv[3]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)
name = f"{tag}_{key}"
md = subprocess.check_output([sys.executable, os.path.join(root, "scripts", "summarize_prof.py"), src]).decode()
md = "\n".join(l for l in md.splitlines() if "at::native" not in l and "rocclr" not in l)
open(os.path.join(dst, f"{name}_summary.md"), "w").write(md + "\n")
KERNELS = ("k_output_counts", "k_feature_events", "k_seqmatch_topk")
def short(n):
    for k in KERNELS:
        if k in n: return k
    return None
avg_ms = {}
for f in glob.glob(os.path.join(src, "trace", "*", "*kernel_stats.csv")):
    shutil.copy(f, os.path.join(dst, f"{name}_kernel_stats.csv"))
    for r in csv.DictReader(open(f)):
        k = short(r["Name"])
        if k: avg_ms[k] = float(r["AverageNs"]) / 1e6
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(src, "pmc_*", "*", "*counter_collection.csv")):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if k: per[(r["Dispatch_Id"], k, r["Counter_Name"])] += float(r["Counter_Value"])
    for (d, k, c), v in per.items(): vals[k][c].append(v)
ppath = os.path.join(dst, "pmc.json")
T = json.load(open(ppath)) if os.path.exists(ppath) else {}
T["_note"] = ("Per launch, from separate rocprofv3 passes (kernel trace; one --pmc pass per counter group). "
              "hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE) KB for kernels whose fabric reads are 16 B/lane "
              "streaming loads (gfx950 tallies those 128-B requests at 64 B) and (FETCH_SIZE + WRITE_SIZE) KB for the "
              "dword-per-lane k_seqmatch_topk (calibrated: it reads the 4*Q*P-byte counts matrix once); hbm_bytes_raw "
              "is the uncorrected sum.  k_output_counts mixes float4 tile staging with scalar stream loads, so its true "
              "figure lies between the two.  valu_issue_frac = SQ_INSTS_VALU / (1024 SIMDs x launch time x 1.2 "
              "VALU instructions per ns).")
FACTOR = {"k_output_counts": 2.0, "k_feature_events": 1.0, "k_seqmatch_topk": 1.0}
T[key] = {}
mean = lambda v: sum(v) / max(1, len(v))
for k in vals:
    c = {n: mean(v) for n, v in vals[k].items()}
    e = {"tag": tag, "avg_launch_ms": avg_ms.get(k)}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        e["FETCH_SIZE_KB"], e["WRITE_SIZE_KB"] = c["FETCH_SIZE"], c["WRITE_SIZE"]
        e["hbm_bytes_per_launch"] = (FACTOR[k] * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
        e["hbm_bytes_raw"] = (c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
    for n in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_BRANCH", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_WAVES",
              "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "GRBM_GUI_ACTIVE",
              "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT"):
        if n in c: e[n] = c[n]
    if "SQ_INSTS_VALU" in c and avg_ms.get(k):
        e["valu_issue_frac"] = c["SQ_INSTS_VALU"] / (1024 * avg_ms[k] * 1e6 * 1.2)
    T[key][k] = e
json.dump(T, open(ppath, "w"), indent=1)
print(open(os.path.join(dst, f"{name}_summary.md")).read()[:600]); print(json.dumps(T[key], indent=1)[:1500])
