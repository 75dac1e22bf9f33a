#!/bin/bash
# Same-box comparison of the output kernel's batch scheduling: whole-tile rounds on / off.  ab_whole.sh <reps> [bench args]
R=$1; shift
for r in $(seq $R); do
  for W in 1 0; do
    export LENS_AMD_OUTPUT_WHOLE=$W
    timeout -k 10 200 python bench.py --no-extras --no-cpu-baseline --steps 20 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('whole', $W, d['kernels_ms']['k_output_counts'], d['ms_per_step'])" || exit 1
  done
done
