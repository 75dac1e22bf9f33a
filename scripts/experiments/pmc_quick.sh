#!/bin/bash
# PMC counters of the three hot kernels on the quick_bench workload (development aid). Usage: scripts/experiments/pmc_quick.sh <tag> [quick_bench args]
set -u
TAG=${1:-x}; shift || true
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmcq_$TAG
mkdir -p $OUT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_IFETCH SQ_INSTS SQ_INST_CYCLES_SALU" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_ICACHE_BUSY_CYCLES SQC_TC_INST_REQ SQC_DCACHE_REQ SQC_DCACHE_MISSES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/g$i -- python3 scripts/quick_bench.py --iters 2 "$@" > $OUT/g$i.log 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
per = collections.defaultdict(float)          # a counter is reported per XCD / shader engine: summed per dispatch
for f in glob.glob(sys.argv[1] + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = next((n for n in ("k_output_counts", "k_feature_events", "k_seqmatch_topk") if n in r["Kernel_Name"]), None)
        if k:
            per[(k, r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
for (k, d, c), v in per.items():
    acc[k][c].append(v)
for k in acc:
    if True:
        print("==", k)
        for c in sorted(acc[k]):
            v = acc[k][c]
            print(f"   {c:32s} {sum(v)/len(v):14.4g}   n={len(v)}")
PY
