#!/bin/bash
# Block-size sweep and fixed start-up cost of the seqkit-hip driver.  GPU box only.
set -e
mkdir -p gpurun_out
NREC=${NREC:-10000000}
FQ=/tmp/e2e_big.fq
[ -f $FQ ] || python3 tools/gen_fastx.py $FQ $NREC 150
head -n 16 $FQ > /tmp/tiny.fq
cat $FQ > /dev/null
w() { local s=$(date +%s.%N); "$@" > /dev/null; local e=$(date +%s.%N); echo "$s $e" | awk '{printf "wall %.3f s\n", $2-$1}'; }
{
echo "== tiny file (fixed cost)"; for i in 1 2 3; do w ./seqkit_amd/seqkit-hip stats /tmp/tiny.fq; done
for mb in 8 16 32 64 128 256; do
  echo "== seq -r -p, blocks of $mb MiB"
  for i in 1 2; do SKH_TIMING=1 w ./seqkit_amd/seqkit-hip seq -r -p --batch-bytes $((mb<<20)) $FQ; done
done
} 2>&1 | tee gpurun_out/e2e_sweep.log
