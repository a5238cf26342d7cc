#!/bin/bash
# A/B of library variants under variants/ on the same box: reverse cases at three read lengths
set -e
mkdir -p gpurun_out; : > gpurun_out/ab.log
for rep in 1 2; do
for L in 150 100 250; do
  n=$((1500000000 / L))
  for lib in variants/*.so; do
    echo "== rep=$rep L=$L $lib" | tee -a gpurun_out/ab.log
    SKH_LIB=$PWD/$lib NREC=$n READLEN=$L CASES="${CASES:-revcomp seq only,reverse only}" timeout -k 10 120 python3 tools/microbench.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/ab.log
  done
done
done
