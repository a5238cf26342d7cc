#!/bin/bash
# usage: tools/pmc_run.sh <tag> <counters...>   (one rocprofv3 --pmc pass over a short bench run)
tag=$1; shift
export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc "$@" --kernel-include-regex "skh" --output-format csv -d gpurun_out/pmc_$tag -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-verify > gpurun_out/pmc_$tag.log 2>&1 || { echo "pass $tag failed"; tail -5 gpurun_out/pmc_$tag.log; exit 1; }
python tools/pmc_summary.py gpurun_out/pmc_$tag | grep -A12 -E "reverse_(short|span)|count_short"
