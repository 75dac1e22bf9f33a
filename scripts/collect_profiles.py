#!/usr/bin/env python3
"""Copy the judged summaries of a scripts/profile.sh run from gpurun_out/ into profiles/ and refresh
profiles/traffic.json (HBM bytes per launch from the FETCH_SIZE / WRITE_SIZE PMC passes).
Usage: collect_profiles.py <round-tag> <gpurun_out/prof_dir> <config-key e.g. P1000_Q4096>"""
import csv, glob, json, os, shutil, subprocess, sys, collections
tag, src, key = sys.argv[1], sys.argv[2].rstrip("/"), sys.argv[3]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)
name = f"{tag}_{key}"
md = subprocess.check_output([sys.executable, os.path.join(root, "scripts", "summarize_prof.py"), src]).decode()
md = "\n".join(l for l in md.splitlines() if "at::native" not in l and "rocclr" not in l)
open(os.path.join(dst, f"{name}_summary.md"), "w").write(md + "\n")
for f in glob.glob(os.path.join(src, "trace", "*", "*kernel_stats.csv")):
    shutil.copy(f, os.path.join(dst, f"{name}_kernel_stats.csv"))
def short(n):
    for k in ("k_output_counts", "k_feature_events", "k_seqmatch_topk"):
        if k in n: return k
    return None
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(src, "pmc_*SIZE*", "*", "*counter_collection.csv")):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if k: per[(r["Dispatch_Id"], k, r["Counter_Name"])] += float(r["Counter_Value"])
    for (d, k, c), v in per.items(): vals[k][c].append(v)
tpath = os.path.join(dst, "traffic.json")
T = json.load(open(tpath)) if os.path.exists(tpath) else {}
# Read-side correction (MI355X_MICROARCH.md, HBM section): on gfx950 FETCH_SIZE tallies 128-B requests at 64 B
# for wide (16 B/lane) streaming loads -> x2 for k_output_counts (its HBM reads are the float4 staging of the
# W2^T tiles).  Calibrated here for dword-per-lane reads: k_seqmatch_topk at P=100k reports exactly the
# 1.6384e9 B of the counts matrix, so x1 for the 4 B/lane kernels.  WRITE_SIZE is exact (checked: 4*Q*P).
FACTOR = {"k_output_counts": 2.0, "k_feature_events": 1.0, "k_seqmatch_topk": 1.0}
T["_note"] = ("HBM bytes per launch from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes: "
              "bytes = (c*FETCH_SIZE + WRITE_SIZE)*1024 with c = 2 for 16 B/lane streaming loads (gfx950 tallies "
              "128-B requests at 64 B) and c = 1 for dword-per-lane loads (calibrated on k_seqmatch_topk, which "
              "reads the 4*Q*P-byte counts matrix exactly once from HBM); WRITE_SIZE is exact.")
T[key] = {}
for k in vals:
    fs = sum(vals[k]["FETCH_SIZE"]) / max(1, len(vals[k]["FETCH_SIZE"]))
    ws = sum(vals[k]["WRITE_SIZE"]) / max(1, len(vals[k]["WRITE_SIZE"]))
    T[key][k] = (FACTOR[k] * fs + ws) * 1024
    T[key][k + "_raw"] = {"FETCH_SIZE_KB": fs, "WRITE_SIZE_KB": ws}
json.dump(T, open(tpath, "w"), indent=1)
print(open(os.path.join(dst, f"{name}_summary.md")).read()[:600]); print(json.dumps(T[key], indent=1))
