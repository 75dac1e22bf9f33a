#!/usr/bin/env python3
"""Summarise a scripts/profile.sh output directory: per-kernel time stats + PMC counter sums per launch."""
import csv, glob, os, sys, collections
d = sys.argv[1]
def short(n):
    for k in ("k_output_counts", "k_feature_events", "k_seqmatch_topk", "k_accumulate_events", "k_topk_merge", "k_seqmatch"):
        if k in n:
            rest = n[n.index(k) + len(k):]
            return k + (rest[:rest.index(">") + 1] if rest.startswith("<") else "")
    return n[:60]
print(f"# {os.path.basename(d.rstrip('/'))}")
for f in glob.glob(os.path.join(d, "trace", "*", "*kernel_stats.csv")):
    print("\n## kernel time (rocprofv3 --kernel-trace --stats)\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---|---|---|---|---|---|")
    for r in csv.DictReader(open(f)):
        print(f"| {short(r['Name'])} | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e3:.1f} | {float(r['MinNs'])/1e3:.1f} | {float(r['MaxNs'])/1e3:.1f} | {float(r['Percentage']):.1f} |")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(d, "pmc_*", "*", "*counter_collection.csv")):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        per[(r["Dispatch_Id"], short(r["Kernel_Name"]), r["Counter_Name"])] += float(r["Counter_Value"])
    for (disp, k, c), v in per.items():
        agg[k][c].append(v)
print("\n## PMC counters (mean per launch over the profiled launches)\n")
for k in sorted(agg):
    if not k.startswith("k_"): continue
    print(f"\n### {k}\n")
    print("| counter | mean per launch | launches |")
    print("|---|---|---|")
    for c in sorted(agg[k]):
        v = agg[k][c]
        print(f"| {c} | {sum(v)/len(v):.4g} | {len(v)} |")
