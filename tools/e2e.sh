#!/bin/bash
# End-to-end (file text in -> text out) runs of the seqkit-hip driver next to the CPU restatement on the
# same synthetic file: wall times, page cache warm, outputs compared byte for byte (md5).  GPU box only.
#   usage: tools/e2e.sh <mode>
#     reads    10 M x 150 bp FASTQ and FASTA: seq -r -p, fx2tab, stats          (NREC=...)
#     c4       nanopore-like FASTQ, 1 kbp .. 1 Mbp reads                          (BASES_WANTED=3e9)
#     c5       24 chromosome-length FASTA records, 3.1 Gbases                     (SCALE=1.0)
#     convert  quality re-encoding
#     file     seq -r -p with the output going to a regular file                  (OUT=/tmp/e2e_out.fq)
#     phases   SKH_TIMING=1 phase totals, 1 / 4 / 8 read threads
#     blocks   text block size 8 .. 256 MiB and the fixed start-up cost
#     gpus     --gpus 1 / 2 / 4 / 8 as far as the box has devices
set -e
mode=${1:-reads}
mkdir -p gpurun_out
DRV=./seqkit_amd/seqkit-hip; ORA=./oracle/seqkit_oracle
L=${L:-150}
NREC=${NREC:-10000000}
BASES=$((NREC * L))
t() { local s=$(date +%s.%N); "$@" > /dev/null; local e=$(date +%s.%N); echo "$s $e" | awk -v b=$BASES '{printf "%8.3f s  %8.3f Gbases/s", $2-$1, b/($2-$1)/1e9}'; }
pair() {   # pair "<command>" file: driver and oracle timed, then their md5
  echo "gpu driver  $1 $(basename $2): $(t timeout -k 10 600 $DRV $1 $2)"
  echo "cpu oracle  $1 $(basename $2): $(t timeout -k 10 600 $ORA $1 $2)"
  echo "md5 gpu / cpu: $($DRV $1 $2 2>/dev/null | md5sum | cut -c1-32) $($ORA $1 $2 2>/dev/null | md5sum | cut -c1-32)"
}
warm() { cat "$@" > /dev/null; $DRV stats $1 > /dev/null; }   # page cache; the first GPU process on a fresh box pays for driver start-up
case $mode in
reads)
  FQ=/tmp/e2e.fq; FA=/tmp/e2e.fa
  python3 tools/gen_fastx.py $FQ $NREC $L; python3 tools/gen_fastx.py $FA $NREC $L 0xC3 --fasta --width 60
  warm $FQ $FA
  { echo "records $NREC x $L bp"
    for cmd in "seq -r -p" "fx2tab -n -l -g -B AGCT" "stats"; do for f in $FQ $FA; do pair "$cmd" $f; done; done; } | tee gpurun_out/e2e_reads.log ;;
c4)
  FQ=/tmp/e2e_c4.fq; python3 tools/gen_ont.py $FQ ${BASES_WANTED:-3e9}; warm $FQ
  BASES=$($ORA stats -T $FQ | awk 'NR==2 {print $5}')
  { echo "bases $BASES"; for cmd in "stats" "stats -a -T" "seq -r -p" "fx2tab -n -l -g -B AGCT"; do pair "$cmd" $FQ; done; } 2>&1 | tee gpurun_out/e2e_c4.log ;;
c5)
  FA=/tmp/e2e_c5.fa; python3 tools/gen_chromosomes.py $FA ${SCALE:-1.0}; warm $FA
  BASES=$($ORA stats -T $FA | awk 'NR==2 {print $5}')
  { echo "bases $BASES"; for cmd in "seq -r -p" "seq -r -p -w 0" "stats -a" "fx2tab -n -l -g -B AGCT"; do pair "$cmd" $FA; done
    $DRV seq -w 0 $FA > /tmp/e2e_c5_w0.fa; pair "seq -r -p" /tmp/e2e_c5_w0.fa; } 2>&1 | tee gpurun_out/e2e_c5.log ;;
convert)
  FQ=/tmp/e2e.fq; python3 tools/gen_fastx.py $FQ $NREC $L; warm $FQ
  { echo "records $NREC x $L bp"
    for cmd in "convert --to Illumina-1.3+" "convert --from Sanger --to Illumina-1.3+" "convert --to Illumina-1.3+ -f"; do pair "$cmd" $FQ 2>/dev/null; done
    echo "stdin (one-pass reader): $(t timeout -k 10 300 $DRV convert --to Illumina-1.3+ < $FQ 2>/dev/null)"; } | tee gpurun_out/e2e_convert.log ;;
file)
  FQ=/tmp/e2e_big.fq; [ -f $FQ ] || python3 tools/gen_fastx.py $FQ $NREC $L; warm $FQ
  OUT=${OUT:-/tmp/e2e_out.fq}
  { df -h $(dirname $OUT) | tail -1
    for i in 1 2; do rm -f $OUT; echo "gpu driver -o file: $(SKH_TIMING=1 t $DRV seq -r -p -o $OUT $FQ 2>&1 | tr '\n' ' ')"; done
    rm -f $OUT; echo "cpu oracle -o file: $(t $ORA seq -r -p -o $OUT $FQ)"; rm -f $OUT; } 2>&1 | tee gpurun_out/e2e_file.log ;;
phases)
  FQ=/tmp/e2e_big.fq; [ -f $FQ ] || python3 tools/gen_fastx.py $FQ $NREC $L; warm $FQ
  { for thr in 1 4 8; do for cmd in "seq -r -p" "fx2tab -n -l -g -B AGCT" "stats"; do
      echo "== SKH_READ_THREADS=$thr  $cmd: $(SKH_READ_THREADS=$thr SKH_TIMING=1 t timeout -k 10 300 $DRV $cmd $FQ 2>&1 | tr '\n' ' ')"
    done; done; } 2>&1 | tee gpurun_out/e2e_phases.log ;;
blocks)
  FQ=/tmp/e2e_big.fq; [ -f $FQ ] || python3 tools/gen_fastx.py $FQ $NREC $L; warm $FQ
  head -n 16 $FQ > /tmp/tiny.fq
  { echo "== tiny file (fixed cost)"; for i in 1 2 3; do echo "$(BASES=600 t $DRV stats /tmp/tiny.fq)"; done
    for mb in 8 16 32 64 128 256; do for i in 1 2; do
      echo "== seq -r -p, blocks of $mb MiB: $(SKH_TIMING=1 t $DRV seq -r -p --batch-bytes $((mb<<20)) $FQ 2>&1 | tr '\n' ' ')"
    done; done; } 2>&1 | tee gpurun_out/e2e_blocks.log ;;
gpus)
  FQ=/tmp/e2e_big.fq; [ -f $FQ ] || python3 tools/gen_fastx.py $FQ $NREC $L; warm $FQ
  ndev=$(python3 -c "import torch; print(torch.cuda.device_count())")
  { echo "devices visible: $ndev"
    for n in 1 2 4 8; do [ $n -le $ndev ] || break
      for cmd in "seq -r -p" "fx2tab -n -l -g -B AGCT" "stats"; do
        echo "--gpus $n  $cmd: $(t timeout -k 10 300 $DRV $cmd --gpus $n $FQ)   md5 $($DRV $cmd --gpus $n $FQ | md5sum | cut -c1-32)"
      done; done
    echo "two contexts on one GPU (--devices 0,0): $(t timeout -k 10 300 $DRV seq -r -p --devices 0,0 $FQ)"; } 2>&1 | tee gpurun_out/e2e_gpus.log ;;
*) echo "unknown mode $mode"; exit 2 ;;
esac
