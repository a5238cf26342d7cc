#!/bin/bash
# End-to-end rate of `convert` (FASTQ quality re-encoding) of the seqkit-hip driver and of the CPU
# restatement on the same synthetic file, page cache warm, output to /dev/null; then both outputs
# once more through md5sum (they must be equal).  GPU box only.
set -e
mkdir -p gpurun_out
NREC=${NREC:-10000000}
L=150
FQ=/tmp/e2e.fq
python3 tools/gen_fastx.py $FQ $NREC $L
ls -l $FQ
cat $FQ > /dev/null
./seqkit_amd/seqkit-hip stats $FQ > /dev/null     # first GPU process on a fresh box pays for driver start-up
t() { local s=$(date +%s.%N); "$@" > /dev/null 2>/dev/null; local e=$(date +%s.%N); echo "$s $e" | awk -v n=$NREC -v l=$L '{printf "%8.3f s  %8.3f Gbases/s", $2-$1, n*l/($2-$1)/1e9}'; }
{
echo "records $NREC x $L bp"
for cmd in "convert --to Illumina-1.3+" "convert --from Sanger --to Illumina-1.3+" "convert --to Illumina-1.3+ -f"; do
  echo "gpu driver  $cmd: $(t timeout -k 10 300 ./seqkit_amd/seqkit-hip $cmd $FQ)"
  echo "gpu driver  $cmd (again): $(t timeout -k 10 300 ./seqkit_amd/seqkit-hip $cmd $FQ)"
  echo "cpu oracle  $cmd: $(t timeout -k 10 300 ./oracle/seqkit_oracle $cmd $FQ)"
  echo "md5 gpu / cpu: $(./seqkit_amd/seqkit-hip $cmd $FQ 2>/dev/null | md5sum | cut -c1-32) $(./oracle/seqkit_oracle $cmd $FQ 2>/dev/null | md5sum | cut -c1-32)"
done
echo "stdin (one-pass reader): $(t timeout -k 10 300 ./seqkit_amd/seqkit-hip convert --to Illumina-1.3+ < $FQ)"
} | tee gpurun_out/e2e_convert.log
