#!/bin/bash
# Phase timing of the seqkit-hip driver (SKH_TIMING=1) on one synthetic FASTQ file, page cache warm,
# output to /dev/null; read threads 1 vs default.  GPU box only.
set -e
mkdir -p gpurun_out
NREC=${NREC:-10000000}
FQ=/tmp/e2e_big.fq
python3 tools/gen_fastx.py $FQ $NREC 150
ls -l $FQ
cat $FQ > /dev/null
{
for thr in 1 4 8; do
  for cmd in "seq -r -p" "fx2tab -n -l -g -B AGCT" "stats"; do
    echo "== SKH_READ_THREADS=$thr  $cmd"
    s=$(date +%s.%N)
    SKH_READ_THREADS=$thr SKH_TIMING=1 timeout -k 10 300 ./seqkit_amd/seqkit-hip $cmd $FQ 2>&1 > /dev/null
    e=$(date +%s.%N)
    echo "$s $e" | awk -v n=$NREC '{printf "wall %.3f s  %.3f Gbases/s\n", $2-$1, n*150/($2-$1)/1e9}'
  done
done
} 2>&1 | tee gpurun_out/e2e_phases.log
