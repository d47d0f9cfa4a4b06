#!/bin/bash
# Round 3, GPU call 25: the split-bf16 Gram on a PART of the chip (its 96 KiB of LDS force one block per CU) beside the SpMM.
set -u
R=$(pwd)
O=$R/gpurun_out/r03y
mkdir -p $O
export TMPDIR=/tmp
( time timeout 420 python scripts/r03_probe.py loop_masks ) > $O/loop_part_split_c3.jsonl 2> $O/err1.log
cut -c1-330 $O/loop_part_split_c3.jsonl | sed 's/"mode": "loop_masks", "n": 9999997, "nnz": 199974337, "d": 256, //'; tail -3 $O/err1.log
( time timeout 300 python scripts/r03_probe.py loop_masks 1000000 10000000 256 ) > $O/loop_part_split_c2.jsonl 2> $O/err2.log
cut -c1-330 $O/loop_part_split_c2.jsonl | sed 's/"mode": "loop_masks", "n": 999994, "nnz": 19998418, "d": 256, //'; tail -3 $O/err2.log
cd /tmp
for v in "CLEORA_GRAM_CO_BLOCKS=-128" "CLEORA_GRAM_CO_BLOCKS=-64"; do
  env $v timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o loop -- python $R/scripts/r03_probe.py loop > $O/loop_t.log 2>&1
  python $R/scripts/loop_timeline.py $O/trace "$v split" | sed 's/gram32/gramXX/g' | tee -a $O/timeline.jsonl | cut -c1-600
  rm -rf $O/trace
done
