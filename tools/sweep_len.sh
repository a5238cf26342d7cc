#!/bin/bash
# reverse kernels at several fixed read lengths, lane-group/staged path vs span path
set -e
mkdir -p gpurun_out; : > gpurun_out/sweep_len.log
for L in 250 400 600 1000 2000 4000; do
  n=$((1500000000 / L))
  for su in 0 4; do
    echo "== L=$L span_u=$su" | tee -a gpurun_out/sweep_len.log
    NREC=$n READLEN=$L SKH_SPAN_U=$su CASES="revcomp seq only,reverse only" timeout -k 10 120 python3 tools/microbench.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/sweep_len.log
  done
done
