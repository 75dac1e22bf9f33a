#!/bin/bash
# Same-box A/B of library builds on k_seqmatch_topk (config 3): ab_topk.sh <rounds> <lib> [<lib> ...]
R=$1; shift
for r in $(seq $R); do
  for L in "$@"; do
    if [ "$L" = default ]; then unset LENS_AMD_SO; else export LENS_AMD_SO=$PWD/$L; fi
    timeout -k 10 200 python bench.py --no-extras --no-cpu-baseline --steps 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', d['kernels_ms']['k_seqmatch_topk'], d['ms_per_step'])" || exit 1
  done
done
