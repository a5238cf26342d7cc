#!/bin/bash
# usage: sweep_env.sh VAR "v1 v2 ..." : runs tools/microbench.py (CASES filter applies) once per value
set -e
mkdir -p gpurun_out
var=$1; shift
for v in $1; do
    echo "== $var=$v" | tee -a gpurun_out/sweep_env.log
    env $var=$v timeout -k 10 120 python3 tools/microbench.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/sweep_env.log
done
