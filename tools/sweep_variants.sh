#!/bin/bash
# Times the reverse cases with each library variant under variants/ (one process each,
# stops at the first failure).
set -e
mkdir -p gpurun_out
for lib in variants/*.so; do
    echo "== $lib" | tee -a gpurun_out/variants.log
    SKH_LIB=$PWD/$lib CASES="${CASES:-revcomp seq only,reverse only,torch copy}" timeout -k 10 120 python3 tools/microbench.py 2>&1 | tee -a gpurun_out/variants.log
done
