#!/bin/bash
# Times the reverse cases for a range of grid caps (workgroups), staged and direct kernels.
set -e
mkdir -p gpurun_out
: > gpurun_out/grid.log
for staged in 1 0; do
for g in 512 768 1024 1280 1536 2048; do
    echo "== staged=$staged max_grid=$g" | tee -a gpurun_out/grid.log
    SKH_STAGED=$staged SKH_MAX_GRID=$g CASES="${CASES:-revcomp seq only,reverse only}" timeout -k 10 120 python3 tools/microbench.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/grid.log
done
done
