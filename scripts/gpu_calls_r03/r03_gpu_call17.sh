#!/bin/bash
# Round 3, GPU call 17: kernel timeline of the whitened loop (rocprofv3 kernel trace) with the statistics on all / a part of the chip.
set -u
R=$(pwd)
O=$R/gpurun_out/r03q
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in 1 -64 -128; do
  CLEORA_GRAM_CO_BLOCKS=$v timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$v -o loop -- python $R/scripts/r03_probe.py loop > $O/loop_$v.log 2>&1
  grep "^{" $O/loop_$v.log | cut -c1-300
  python $R/scripts/loop_timeline.py $O/trace_$v "CLEORA_GRAM_CO_BLOCKS=$v" | tee -a $O/timeline.jsonl
  # keep only the windows' worth of trace out of the merge budget
  find $O/trace_$v -name "*.csv" -size +20M -delete
done
