#!/bin/bash
# seq -r -p with the output going to a regular file instead of /dev/null.  GPU box only.
set -e
mkdir -p gpurun_out
NREC=${NREC:-10000000}
FQ=/tmp/e2e_big.fq
[ -f $FQ ] || python3 tools/gen_fastx.py $FQ $NREC 150
cat $FQ > /dev/null
./seqkit_amd/seqkit-hip stats $FQ > /dev/null
OUT=${OUT:-/tmp/e2e_out.fq}
w() { local s=$(date +%s.%N); "$@"; local e=$(date +%s.%N); echo "$s $e" | awk -v n=$NREC '{printf "wall %.3f s  %.3f Gbases/s\n", $2-$1, n*150/($2-$1)/1e9}'; }
{
df -h $(dirname $OUT) | tail -1
for i in 1 2; do
  rm -f $OUT
  echo "gpu driver -o file: $(SKH_TIMING=1 w ./seqkit_amd/seqkit-hip seq -r -p -o $OUT $FQ 2>&1 | tr '\n' ' ')"
done
rm -f $OUT
echo "cpu oracle -o file: $(w ./oracle/seqkit_oracle seq -r -p -o $OUT $FQ)"
rm -f $OUT
} 2>&1 | tee gpurun_out/e2e_file.log
