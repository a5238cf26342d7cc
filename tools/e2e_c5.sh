#!/bin/bash
# Human-genome-shaped FASTA (24 chromosome-length records, 3.1 Gbases) through the seqkit-hip
# driver and through the CPU restatement: outputs compared byte for byte (md5), wall times, page
# cache warm.  GPU box only.
set -e
mkdir -p gpurun_out
SCALE=${SCALE:-1.0}
FA=/tmp/e2e_c5.fa
python3 tools/gen_chromosomes.py $FA $SCALE
ls -l $FA
cat $FA > /dev/null
./seqkit_amd/seqkit-hip stats $FA > /dev/null      # first GPU process on a fresh box pays for driver start-up
BASES=$(./oracle/seqkit_oracle stats -T $FA | awk 'NR==2 {print $5}')
t() { local s=$(date +%s.%N); "$@" > /dev/null; local e=$(date +%s.%N); echo "$s $e" | awk -v b=$BASES '{printf "%8.3f s  %8.3f Gbases/s\n", $2-$1, b/($2-$1)/1e9}'; }
{
echo "bases $BASES"
for cmd in "seq -r -p" "seq -r -p -w 0" "stats -a" "fx2tab -n -l -g -B AGCT"; do
  echo "gpu driver  $cmd: $(SKH_TIMING=1 t timeout -k 10 600 ./seqkit_amd/seqkit-hip $cmd $FA)"
  echo "cpu oracle  $cmd: $(t timeout -k 10 600 ./oracle/seqkit_oracle $cmd $FA)"
  echo "md5 gpu / cpu: $(./seqkit_amd/seqkit-hip $cmd $FA | md5sum | cut -c1-32) $(./oracle/seqkit_oracle $cmd $FA | md5sum | cut -c1-32)"
done
# the unwrapped output read back: one text line per chromosome
./seqkit_amd/seqkit-hip seq -w 0 $FA > /tmp/e2e_c5_w0.fa
echo "gpu driver  seq -r -p (unwrapped input): $(SKH_TIMING=1 t timeout -k 10 600 ./seqkit_amd/seqkit-hip seq -r -p /tmp/e2e_c5_w0.fa)"
echo "md5 gpu / cpu: $(./seqkit_amd/seqkit-hip seq -r -p /tmp/e2e_c5_w0.fa | md5sum | cut -c1-32) $(./oracle/seqkit_oracle seq -r -p /tmp/e2e_c5_w0.fa | md5sum | cut -c1-32)"
} 2>&1 | tee gpurun_out/e2e_c5.log
