#!/bin/bash
# Build the gfx950 kernel library in-tree.  hipcc cross-compiles without a GPU.
#   -ffp-contract=off : a*b+c stays two roundings (numpy/scipy operator order); fused ops are written fma().
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-function"
OBJS=()
PIDS=()
NAMES=()
for f in *.hip; do
  o="${f%.hip}.o"
  stale=0
  [ -f "$o" ] || stale=1
  extra=""; [ "$f" = fused_voice_b.hip ] && extra=fused_voice.hip      # (it is that file, compiled for the other waveforms)
  for dep in "$f" $extra sig_common.h sig_osc.h sig_biquad.h sig_adsr.h sig_bus_tile.h sig_mix_tile.h sig_steady.h ../../include/signals_amd.h; do
    [ "$stale" = 1 ] || { [ "$dep" -nt "$o" ] && stale=1; } || true
  done
  if [ "$stale" = 1 ]; then
    rm -f "$o"                      # a failed compile must not leave the old object for the link step
    $HIPCC $FLAGS -c "$f" -o "$o" &
    PIDS+=($!)
    NAMES+=("$f")
  fi
  OBJS+=("$o")
done
failed=0
for i in "${!PIDS[@]}"; do            # a bare `wait` returns 0 whatever the children returned
  if ! wait "${PIDS[$i]}"; then
    echo "build.sh: compiling ${NAMES[$i]} failed" >&2
    failed=1
  fi
done
[ "$failed" = 0 ] || exit 1
$HIPCC --offload-arch=gfx950 -shared -fPIC -o libsignals_amd.so "${OBJS[@]}"
echo "built $(pwd)/libsignals_amd.so"
# torch-free example of the C ABI (examples/c2_direct.cpp); tests/test_abi_direct.py runs it on the GPU box
$HIPCC --offload-arch=gfx950 -O2 -I ../../include ../../examples/c2_direct.cpp -L . -lsignals_amd \
    -Wl,-rpath,'$ORIGIN/../signals_amd/csrc' -o ../../examples/c2_direct
echo "built $(cd ../../examples && pwd)/c2_direct"
