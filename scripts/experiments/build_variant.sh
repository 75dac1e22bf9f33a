#!/bin/bash
# Build a kernel variant into variants/<name>.so from a scratch copy of the sources:
#   scripts/experiments/build_variant.sh <name> [VAR=value ...]   (the variables reach the two generators)
# Use with LENS_AMD_SO=variants/<name>.so (scripts/experiments/ab.sh).  EXTRA_HIPCC adds compiler flags, e.g.
#   EXTRA_HIPCC=-DLENS_OUTPUT_JUMP scripts/experiments/build_variant.sh jump LENS_GEN_JUMP=1
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
W=/tmp/lens_variant_$NAME
rm -rf $W; mkdir -p $W/lens_amd/csrc $W/include $ROOT/variants
cp $ROOT/lens_amd/csrc/* $W/lens_amd/csrc/; cp $ROOT/include/*.h $W/include/
env "$@" python3 $ROOT/scripts/gen_output_loop.py $W/lens_amd/csrc > /dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -Xclang -target-feature -Xclang -packed-fp32-ops \
  -fPIC -shared -w $EXTRA_HIPCC -o $ROOT/variants/$NAME.so $W/lens_amd/csrc/lens_kernels.hip
ls -la $ROOT/variants/$NAME.so
