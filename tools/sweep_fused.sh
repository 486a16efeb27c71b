#!/bin/bash
# sweep of the fused walker's launch geometry (and of tuning builds, tools/build_variant.sh) on the C2 shapes;
# run on the GPU box:  tools/sweep_fused.sh fusedbus|fused [variant names...]
mode=${1:-fusedbus}; shift || true
libs=("" "$@")
for lib in "${libs[@]}"; do
  [ -n "$lib" ] && export SIG_LIB_PATH=$PWD/scratch/variants/lib_$lib.so
  for vpt in 1 2 4; do for span in 1 2 4 8; do
    echo "lib=${lib:-product} vpt=$vpt span=$span $(SIG_FUSED_VPT=$vpt SIG_FUSED_SPAN=$span TUNE=$mode python tools/tune_kernels.py 2>&1 | grep fused | sed 's/.*K=/K=/' | tr '\n' ' ')"
  done; done
  echo "lib=${lib:-product} default $(TUNE=$mode python tools/tune_kernels.py 2>&1 | grep fused | sed 's/.*K=/K=/' | tr '\n' ' ')"
done
