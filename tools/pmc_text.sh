#!/bin/bash
# HBM traffic of the text stages: one rocprofv3 --pmc pass per counter over tools/microbench_text.py.
#   tools/pmc_text.sh <outdir under gpurun_out>
set -e
out=gpurun_out/$1
mkdir -p $out
export TMPDIR=/tmp SKIP_CHECK=1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 180 rocprofv3 --pmc $c --kernel-include-regex "skh" --output-format csv -d $out/pmc_$c -- python3 tools/microbench_text.py > $out/pmc_$c.log 2>&1
  python3 tools/pmc_summary.py $out/pmc_$c > $out/$c.txt
done
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o text -- python3 tools/microbench_text.py > $out/stats.log 2>&1
python3 tools/prof_summary.py $out/stats > $out/kernel_stats_text.txt
timeout -k 10 200 python3 tools/microbench_text.py > $out/text_stages.txt 2>&1
