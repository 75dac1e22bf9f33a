#!/bin/bash
# Same-box A/B of library builds: scripts/experiments/ab.sh <rounds> <lib> [<lib> ...]   ("default" = the in-tree build)
# Prints k_output_counts' average launch time of bench.py's config 3 for every library, alternating, <rounds> times.
R=$1; shift
for r in $(seq $R); do
  for L in "$@"; do
    if [ "$L" = default ]; then unset LENS_AMD_SO; else export LENS_AMD_SO=$PWD/$L; fi
    timeout -k 10 200 python bench.py --no-extras --no-cpu-baseline --steps 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', d['roofline']['avg_launch_ms'], d['ms_per_step'])" || exit 1
  done
done
