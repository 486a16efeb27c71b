#!/bin/bash
# rocprofv3 evidence for one round, run on the GPU box from the repo root:  tools/profile_round.sh <tag>
# -> gpurun_out/prof<tag>/...   (then `python tools/collect_profiles.py <tag> rNN` on the build side)
# Every pass puts the program directly after `--`; counters are collected in passes of their own (no trace domains).
set -eo pipefail
tag=${1:?tag}
root=$PWD
out=$root/gpurun_out/prof$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $root/bench.py --no-cpu-baseline --single-mode"
modes="fused materialised"
[ "${2:-}" = configs ] && modes=""          # tools/profile_round.sh <tag> configs: only the passes over tools/measure_configs.py
for mode in $modes; do
  flag=""; [ $mode = materialised ] && flag="--materialised"
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$mode -- $BENCH --steps 20 --warmup 5 $flag \
      > $out/stats_$mode.json 2> $out/stats_$mode.err
  echo "stats $mode done"
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/pmc_${mode}_$ctr -- $BENCH --steps 4 --warmup 1 --prewarm-ms 0 --no-kernel-timing $flag \
        > $out/pmc_${mode}_$ctr.log 2>&1
    echo "pmc $mode $ctr done"
  done
done
if [ -n "$modes" ]; then
# issue-side counters of the headline (fused) schedule: 8 SQ slots + GRBM per pass
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE \
    --kernel-trace --output-format csv -d $out/pmc_fused_SQ_issue -- $BENCH --steps 6 --warmup 2 --prewarm-ms 100 --no-kernel-timing > $out/pmc_fused_SQ_issue.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE \
    --kernel-trace --output-format csv -d $out/pmc_fused_SQ_lds -- $BENCH --steps 6 --warmup 2 --prewarm-ms 100 --no-kernel-timing > $out/pmc_fused_SQ_lds.log 2>&1
# the fused schedule at 256 blocks per batch (BASELINE's literal "1024 voices x 256 blocks"): one wave per block span
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_k256 -- $BENCH --steps 200 --warmup 20 --blocks 256 \
    > $out/stats_k256.json 2> $out/stats_k256.err
echo "SQ passes done"
fi
# BASELINE configs 3 and 5 (tools/measure_configs.py renders both through the engine's default schedule)
CFG="python3 $root/tools/measure_configs.py 6"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_configs -- $CFG > $out/stats_configs.json 2> $out/stats_configs.err
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/pmc_configs_$ctr -- $CFG > $out/pmc_configs_$ctr.log 2>&1
done
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE \
    --kernel-trace --output-format csv -d $out/pmc_configs_SQ_issue -- $CFG > $out/pmc_configs_SQ_issue.log 2>&1
echo "configs done"
