#!/bin/bash
# Chunks per lane of the span kernel on nanopore-like read lengths.  GPU box only; results in
# gpurun_out/knob_sweep5.log.
mkdir -p gpurun_out
export WHICH=ont
{
for u in 2 3 4 6 8; do
  echo "--- SKH_SPAN_U=$u"
  SKH_SPAN_U=$u timeout -k 10 200 python3 tools/microbench_long.py || exit 1
done
} > gpurun_out/knob_sweep5.log 2>&1
grep -E "^---|^==|revcomp|roundtrip" gpurun_out/knob_sweep5.log
