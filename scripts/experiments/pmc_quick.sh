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
for f in glob.glob(sys.argv[1] + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in acc:
    if "k_output" in k or "k_feature" in k:
        print("==", k)
        for c in sorted(acc[k]):
            v = acc[k][c]
            print(f"   {c:32s} {sum(v)/len(v):14.4g}   n={len(v)}")
PY
