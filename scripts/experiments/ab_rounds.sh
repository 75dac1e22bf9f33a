#!/bin/bash
# Same-box sweep of LENS_AMD_OUTPUT_ROUNDS (workgroups per CU of the output kernel's batch mode): ab_rounds.sh <reps> "<r> <r> ..." [bench args]
R=$1; RS=$2; shift 2
for r in $(seq $R); do
  for W in $RS; do
    export LENS_AMD_OUTPUT_ROUNDS=$W
    timeout -k 10 200 python bench.py --no-extras --no-cpu-baseline --steps 20 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rounds', $W, d['kernels_ms']['k_output_counts'], d['ms_per_step'])" || exit 1
  done
done
