#!/bin/bash
# one rocprofv3 --pmc pass per counter group over the reverse-complement of one length distribution:
#   tools/pmc_shape_pass.sh <shape of tools/microbench_long.py> <tag>
shape=$1; tag=$2
export TMPDIR=/tmp WHICH=$shape OPS=revcomp
mkdir -p gpurun_out
i=0
for group in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
             "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM" \
             "TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
             "TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum" "TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $group --kernel-include-regex "skh" --output-format csv -d gpurun_out/pmcs_${tag}_$i \
     -- python3 tools/microbench_long.py > gpurun_out/pmcs_${tag}_$i.log 2>&1 || { echo "pass $i failed"; tail -5 gpurun_out/pmcs_${tag}_$i.log; exit 1; }
  python3 tools/pmc_summary.py gpurun_out/pmcs_${tag}_$i | grep -A9 "reverse_staged\|reverse_rows"
done
