#!/bin/bash
# One tuning knob of the library (an SKH_* variable read by skh_init) over a list of values: the kernels
# of the benchmark batch (150 bp) and the length distributions of tools/microbench_long.py.  GPU box only.
#   usage: tools/knob_sweep.sh VAR "v1 v2 ..." [WHICH]
#   e.g.   tools/knob_sweep.sh SKH_STAGED_STEPS "2 3 4 6 8" trim,r100,r250,mirna
#          tools/knob_sweep.sh SKH_COUNT_STEPS "1 2 3 4"    tools/knob_sweep.sh SKH_STATS_STEPS "1 2 4 8 16" chrom,ont
#          tools/knob_sweep.sh SKH_SPAN_U "2 3 4 6 8" ont    tools/knob_sweep.sh SKH_LOGG "0 1 2 3"   tools/knob_sweep.sh SKH_STAGED "0 1"
var=$1; values=$2
export WHICH=${3:-trim,r100,r250}
mkdir -p gpurun_out
log=gpurun_out/knob_sweep_$var.log
{
for v in $values; do
  echo "--- $var=$v"
  env $var=$v timeout -k 10 200 python3 bench.py --steps 20 --warmup 3 --cpu-seconds 0 --no-e2e --no-configs --no-collective \
    | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench 150bp', {k: v['ms'] for k, v in d['kernels'].items()})" || exit 1
  env $var=$v timeout -k 10 200 python3 tools/microbench_long.py || exit 1
done
} > $log 2>&1
grep -E "^---|^==|revcomp|inplace|count4|stats|bench" $log
