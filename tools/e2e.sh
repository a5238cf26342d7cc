#!/bin/bash
# End-to-end (text in -> text out) rates of the seqkit-hip driver and of the CPU restatement on
# the same synthetic files, page cache warm, output to /dev/null; then both outputs once more through
# md5sum (they must be equal).  GPU box only.
set -e
mkdir -p gpurun_out
NREC=${NREC:-4000000}
L=150
FQ=/tmp/e2e.fq
python3 tools/gen_fastx.py $FQ $NREC $L
FA=/tmp/e2e.fa
python3 tools/gen_fastx.py $FA $NREC $L 0xC3 --fasta --width 60
ls -l $FQ $FA
cat $FQ $FA > /dev/null
./seqkit_amd/seqkit-hip stats $FA > /dev/null     # first GPU process on a fresh box pays for driver start-up
t() { local s=$(date +%s.%N); "$@" > /dev/null; local e=$(date +%s.%N); echo "$s $e" | awk -v n=$NREC -v l=$L '{printf "%8.3f s  %8.3f Gbases/s", $2-$1, n*l/($2-$1)/1e9}'; }
{
echo "records $NREC x $L bp"
for cmd in "seq -r -p" "fx2tab -n -l -g -B AGCT" "stats"; do
  for f in $FQ $FA; do
    echo "gpu driver  $cmd $(basename $f): $(t timeout -k 10 300 ./seqkit_amd/seqkit-hip $cmd $f)"
    echo "cpu oracle  $cmd $(basename $f): $(t timeout -k 10 300 ./oracle/seqkit_oracle $cmd $f)"
    echo "md5 gpu / cpu: $(./seqkit_amd/seqkit-hip $cmd $f | md5sum | cut -c1-32) $(./oracle/seqkit_oracle $cmd $f | md5sum | cut -c1-32)"
  done
done
} | tee gpurun_out/e2e.log
