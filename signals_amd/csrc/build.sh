#!/bin/bash
# Build the gfx950 kernel library in-tree.  hipcc cross-compiles without a GPU.
#   -ffp-contract=off : a*b+c stays two roundings (numpy/scipy operator order); fused ops are written fma().
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-function"
OBJS=()
for f in *.hip; do
  o="${f%.hip}.o"
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ sig_common.h -nt "$o" ] || [ sig_osc.h -nt "$o" ] || [ sig_biquad.h -nt "$o" ] || [ sig_adsr.h -nt "$o" ] || [ sig_bus_tile.h -nt "$o" ] || [ ../../include/signals_amd.h -nt "$o" ]; then
    $HIPCC $FLAGS -c "$f" -o "$o" &
  fi
  OBJS+=("$o")
done
wait
$HIPCC --offload-arch=gfx950 -shared -fPIC -o libsignals_amd.so "${OBJS[@]}"
echo "built $(pwd)/libsignals_amd.so"
# torch-free example of the C ABI (examples/c2_direct.cpp); tests/test_abi_direct.py runs it on the GPU box
$HIPCC --offload-arch=gfx950 -O2 -I ../../include ../../examples/c2_direct.cpp -L . -lsignals_amd \
    -Wl,-rpath,'$ORIGIN/../signals_amd/csrc' -o ../../examples/c2_direct 2>/dev/null
echo "built $(cd ../../examples && pwd)/c2_direct"
