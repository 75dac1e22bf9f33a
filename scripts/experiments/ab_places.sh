#!/bin/bash
# Same-box comparison of library builds over database sizes: ab_places.sh "<P> <P> ..." <lib> [<lib> ...]
PS=$1; shift
for P in $PS; do
  for L in "$@"; do
    if [ "$L" = default ]; then unset LENS_AMD_SO; else export LENS_AMD_SO=$PWD/$L; fi
    timeout -k 10 200 python bench.py --places $P --no-extras --no-cpu-baseline --steps 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print($P, '$L', d['roofline']['avg_launch_ms'], d['ms_per_step'])" || exit 1
  done
done
