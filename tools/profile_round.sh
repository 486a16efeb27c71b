#!/bin/bash
# rocprofv3 evidence for one round, run on the GPU box from the repo root:  tools/profile_round.sh <tag>
# -> gpurun_out/{stats,pmc}<tag>_*  (then tools/collect_profiles.py <tag> rNN on the build side)
set -eo pipefail
tag=${1:?tag}
root=$PWD
cd /tmp && export TMPDIR=/tmp
for mode in fused materialised; do
  flag=""; [ $mode = materialised ] && flag="--materialised"
  rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/stats${tag}_$mode -- \
      python3 $root/bench.py --steps 20 --warmup 5 --no-cpu-baseline --single-mode $flag \
      > $root/gpurun_out/stats${tag}_$mode.json 2> $root/gpurun_out/stats${tag}_$mode.err
  echo "stats $mode done"
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $root/gpurun_out/pmc${tag}_${mode}_$ctr -- \
        python3 $root/bench.py --steps 4 --warmup 1 --prewarm-ms 0 --no-cpu-baseline --no-kernel-timing --single-mode $flag \
        > $root/gpurun_out/pmc${tag}_${mode}_$ctr.log 2>&1
    echo "pmc $mode $ctr done"
  done
done
