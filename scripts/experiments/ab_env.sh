#!/bin/bash
# Same-box comparison of "<lib>:<LENS_AMD_OUTPUT_PPL>" pairs: scripts/experiments/ab_env.sh <rounds> default:0 variants/x.so:2 ...
R=$1; shift
for r in $(seq $R); do
  for S in "$@"; do
    L=${S%%:*}; P=${S##*:}
    if [ "$L" = default ]; then unset LENS_AMD_SO; else export LENS_AMD_SO=$PWD/$L; fi
    if [ "$P" = 0 ]; then unset LENS_AMD_OUTPUT_PPL; else export LENS_AMD_OUTPUT_PPL=$P; fi
    timeout -k 10 200 python bench.py --no-extras --no-cpu-baseline --steps 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$S', d['roofline']['avg_launch_ms'], d['ms_per_step'])" || exit 1
  done
done
