#!/bin/bash
# tuning builds of ONE kernel file: [FILE=fused_voice.hip] tools/build_variant.sh <name> [-D...]  ->  scratch/variants/lib_<name>.so
# (same ABI, loaded with SIG_LIB_PATH; for fused_voice.hip -DSIG_TUNE_SINE_ONLY restricts the template instantiations
# to what tools/tune_kernels.py launches so a variant builds in seconds)
set -euo pipefail
cd "$(dirname "$0")/../signals_amd/csrc"
name=$1; shift
file=${FILE:-fused_voice.hip}
stem=${file%.hip}
mkdir -p ../../scratch/variants
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-function "$@" \
    -c "$file" -o ../../scratch/variants/${stem}_$name.o
objs=$(ls *.o | grep -v "^${stem}.o$")
$HIPCC --offload-arch=gfx950 -shared -fPIC -o ../../scratch/variants/lib_$name.so ../../scratch/variants/${stem}_$name.o $objs
echo "built scratch/variants/lib_$name.so"
