#!/bin/bash
# Same-box A/B of library builds on the real-density object of bench.py: ab_real.sh <rounds> <lib> [<lib> ...]
R=$1; shift
for r in $(seq $R); do
  for L in "$@"; do
    if [ "$L" = default ]; then unset LENS_AMD_SO; else export LENS_AMD_SO=$PWD/$L; fi
    timeout -k 10 300 python bench.py --no-cpu-baseline --steps 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['other_configs']['real_density']; print('$L', r['ms_per_step'], r['value'])" || exit 1
  done
done
