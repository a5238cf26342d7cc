#!/bin/bash
# Runs of chunks per workgroup of the stats kernel: the benchmark batch and other length
# distributions.  GPU box only; results in gpurun_out/knob_sweep4.log.
mkdir -p gpurun_out
export WHICH=${WHICH:-chrom,ont,trim,r250}
{
for s in 1 2 4 8 16; do
  echo "--- SKH_STATS_STEPS=$s"
  SKH_STATS_STEPS=$s timeout -k 10 200 python3 bench.py --steps 20 --warmup 3 --cpu-seconds 0 --no-e2e | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench 150bp stats', d['kernels']['stats'])" || exit 1
  SKH_STATS_STEPS=$s timeout -k 10 120 python3 tools/microbench_long.py || exit 1
done
} > gpurun_out/knob_sweep4.log 2>&1
grep -E "^---|^==|stats" gpurun_out/knob_sweep4.log
