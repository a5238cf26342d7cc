#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, one rocprofv3 --pmc pass each) of the hot-path kernels on the
# long-record shapes of tools/microbench_long.py, and the wave-state counters of the counting kernel
# next to the stats kernel on the bench workload.  Results under gpurun_out/; summaries go to profiles/.
#   usage: tools/pmc_shapes.sh <tag>      e.g. r02
tag=${1:-rNN}
export TMPDIR=/tmp
mkdir -p gpurun_out
for which in chrom ont mixed; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    WHICH=$which timeout -s KILL 150 rocprofv3 --pmc $ctr --kernel-include-regex "skh" --output-format csv \
      -d gpurun_out/pmc_${which}_$ctr -- python3 tools/microbench_long.py > gpurun_out/pmc_${which}_$ctr.log 2>&1 \
      || { echo "pass $which $ctr failed"; tail -5 gpurun_out/pmc_${which}_$ctr.log; exit 1; }
  done
done
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT \
  --kernel-include-regex "skh" --output-format csv -d gpurun_out/pmc_sq -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-verify --no-e2e --no-configs --no-collective \
  > gpurun_out/pmc_sq.log 2>&1 || { echo "SQ pass failed"; tail -5 gpurun_out/pmc_sq.log; exit 1; }
python3 tools/pmc_shapes_summary.py $tag
