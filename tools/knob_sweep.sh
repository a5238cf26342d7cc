#!/bin/bash
# Short-read length distributions under the library's tuning knobs (lanes per record, steps per
# wavefront, staged or direct reverse kernel).  GPU box only; results in gpurun_out/knob_sweep.log.
mkdir -p gpurun_out
export WHICH=${WHICH:-trim,r100,r250}
{
echo "--- defaults"; timeout -k 10 120 python3 tools/microbench_long.py || exit 1
for g in 0 1 2 3; do
  echo "--- SKH_LOGG=$g"; SKH_LOGG=$g timeout -k 10 120 python3 tools/microbench_long.py || exit 1
done
for s in 1 3 4; do
  echo "--- SKH_STAGED_STEPS=$s SKH_COUNT_STEPS=$s"; SKH_STAGED_STEPS=$s SKH_COUNT_STEPS=$s timeout -k 10 120 python3 tools/microbench_long.py || exit 1
done
echo "--- SKH_STAGED=0"; SKH_STAGED=0 timeout -k 10 120 python3 tools/microbench_long.py || exit 1
} > gpurun_out/knob_sweep.log 2>&1
grep -E "^---|^==|revcomp|count4" gpurun_out/knob_sweep.log
