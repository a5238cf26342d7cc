#!/bin/bash
# Nanopore-like FASTQ (1 kbp .. 1 Mbp reads, skewed) through the seqkit-hip driver and through the
# CPU restatement: outputs compared byte for byte (md5), wall times, page cache warm.  GPU box only.
set -e
mkdir -p gpurun_out
BASES_WANTED=${BASES_WANTED:-3e9}
FQ=/tmp/e2e_c4.fq
python3 tools/gen_ont.py $FQ $BASES_WANTED
ls -l $FQ
cat $FQ > /dev/null
./seqkit_amd/seqkit-hip stats $FQ > /dev/null      # first GPU process on a fresh box pays for driver start-up
BASES=$(./oracle/seqkit_oracle stats -T $FQ | awk 'NR==2 {print $5}')
t() { local s=$(date +%s.%N); "$@" > /dev/null; local e=$(date +%s.%N); echo "$s $e" | awk -v b=$BASES '{printf "%8.3f s  %8.3f Gbases/s\n", $2-$1, b/($2-$1)/1e9}'; }
{
echo "bases $BASES"
for cmd in "stats" "stats -a -T" "seq -r -p" "fx2tab -n -l -g -B AGCT"; do
  echo "gpu driver  $cmd: $(t timeout -k 10 600 ./seqkit_amd/seqkit-hip $cmd $FQ)"
  echo "cpu oracle  $cmd: $(t timeout -k 10 600 ./oracle/seqkit_oracle $cmd $FQ)"
  echo "md5 gpu / cpu: $(./seqkit_amd/seqkit-hip $cmd $FQ | md5sum | cut -c1-32) $(./oracle/seqkit_oracle $cmd $FQ | md5sum | cut -c1-32)"
done
} 2>&1 | tee gpurun_out/e2e_c4.log
