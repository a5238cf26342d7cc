#!/bin/bash
# HBM traffic of the hot-path kernels, one operation per pass: for every length distribution and every operation
# two rocprofv3 --pmc passes (FETCH_SIZE, then WRITE_SIZE) over tools/microbench_long.py.
#   tools/pmc_groups.sh <outdir under gpurun_out> "<shapes>" "<ops>"
out=gpurun_out/$1; shapes=${2:-"r100 r250 trim mirna mixed chrom ont"}; ops=${3:-"revcomp inplace count4"}
export TMPDIR=/tmp
mkdir -p $out
for which in $shapes; do
  for op in $ops; do
    for ctr in FETCH_SIZE WRITE_SIZE; do
      d=$out/${which}_${op}_$ctr
      WHICH=$which OPS=$op timeout -s KILL 150 rocprofv3 --pmc $ctr --kernel-include-regex "skh" --output-format csv \
        -d $d -- python3 tools/microbench_long.py > $d.log 2>&1 || { echo "pass $which $op $ctr failed"; tail -5 $d.log; exit 1; }
      echo "$which $op $ctr done"
    done
  done
done
