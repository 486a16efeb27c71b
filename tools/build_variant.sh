#!/bin/bash
# tuning builds of fused_voice.hip: tools/build_variant.sh <name> [-D...]  ->  scratch/variants/lib_<name>.so
# (same ABI, loaded with SIG_LIB_PATH; -DSIG_TUNE_SINE_ONLY restricts the template instantiations to what
# tools/tune_kernels.py launches so a variant builds in seconds)
set -euo pipefail
cd "$(dirname "$0")/../signals_amd/csrc"
name=$1; shift
mkdir -p ../../scratch/variants
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-function -DSIG_TUNE_SINE_ONLY "$@" \
    -c fused_voice.hip -o ../../scratch/variants/fused_$name.o
objs=$(ls *.o | grep -v '^fused_voice.o$')
$HIPCC --offload-arch=gfx950 -shared -fPIC -o ../../scratch/variants/lib_$name.so ../../scratch/variants/fused_$name.o $objs
echo "built scratch/variants/lib_$name.so"
