#!/bin/bash
# Round profile: bench line, rocprofv3 kernel stats of the same command, two PMC passes for
# HBM traffic.  Results under gpurun_out/; copy the summaries into profiles/ afterwards.
set -e
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 400 python3 bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err
tail -c 3000 gpurun_out/bench.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -o bench -- python3 bench.py --steps 20 --warmup 3 --cpu-seconds 0 --no-e2e --no-configs --no-collective > gpurun_out/prof_bench.log 2>&1
python3 tools/prof_summary.py gpurun_out/prof_bench > gpurun_out/kernel_stats_bench.txt
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "skh" --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --sustain 0 --no-verify --no-e2e --no-configs --no-collective > gpurun_out/pmc_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "skh" --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --sustain 0 --no-verify --no-e2e --no-configs --no-collective > gpurun_out/pmc_write.log 2>&1
python3 tools/make_traffic_json.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/hbm_traffic.json 10000000 150 > /dev/null
grep -E "skh::|kernel " gpurun_out/kernel_stats_bench.txt | cut -c1-150
