#!/bin/bash
# Same-box comparison of tile widths over database sizes: ab_ppl.sh <rounds> "<P> <P> ..." "<ppl> <ppl> ..."   (ppl 0 = the library's choice)
R=$1; PS=$2; WS=$3
for r in $(seq $R); do
  for P in $PS; do
    for W in $WS; do
      if [ "$W" = 0 ]; then unset LENS_AMD_OUTPUT_PPL; else export LENS_AMD_OUTPUT_PPL=$W; fi
      timeout -k 10 200 python bench.py --places $P --no-extras --no-cpu-baseline --steps 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print($P, 'ppl', $W, d['roofline']['avg_launch_ms'], d['ms_per_step'])" || exit 1
    done
  done
done
