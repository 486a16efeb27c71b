#!/bin/bash
# Issue-side counters of the headline kernel, run on the GPU box from the repo root:  tools/profile_sq.sh <outdir> [bench flags]
# One rocprofv3 --pmc pass per counter group (8 SQ slots + 2 GRBM slots per pass), program directly after `--`.
set -eo pipefail
out=${1:?outdir}; shift
root=$PWD
mkdir -p $root/$out
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $root/$out/counters_available.txt 2>&1 || true
pass() {
  name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $root/$out/pmc_$name -- \
      python3 $root/bench.py --steps 6 --warmup 2 --prewarm-ms 100 --no-cpu-baseline --no-kernel-timing --single-mode $EXTRA \
      > $root/$out/pmc_$name.log 2>&1
  echo "pass $name done"
}
EXTRA="$*"
pass issue SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE
pass lds SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE
