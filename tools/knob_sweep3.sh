#!/bin/bash
# Steps per wavefront of the short counting kernel: the benchmark batch (150 bp) and the short-read
# distributions.  GPU box only; results in gpurun_out/knob_sweep3.log.
mkdir -p gpurun_out
export WHICH=${WHICH:-trim,r100,r250,mirna}
{
for s in 1 2 3 4 6 8; do
  echo "--- SKH_COUNT_STEPS=$s"
  SKH_COUNT_STEPS=$s timeout -k 10 200 python3 bench.py --steps 20 --warmup 3 --cpu-seconds 0 --no-e2e | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench 150bp', {k: (v['ms'] if 'ms' in v else v) for k, v in d['kernels'].items() if k != 'stats'})" || exit 1
  SKH_COUNT_STEPS=$s timeout -k 10 120 python3 tools/microbench_long.py || exit 1
done
} > gpurun_out/knob_sweep3.log 2>&1
grep -E "^---|^==|count4|bench" gpurun_out/knob_sweep3.log
