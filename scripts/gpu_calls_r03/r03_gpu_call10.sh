#!/bin/bash
# Round 3, GPU call 10: the whitened loop with the statistics stream on its own compute units (CU masks).
set -u
R=$(pwd)
O=$R/gpurun_out/r03j
mkdir -p $O
export TMPDIR=/tmp
( time timeout 420 python scripts/r03_probe.py loop_masks ) > $O/loop_masks.jsonl 2> $O/loop_masks.err
cat $O/loop_masks.jsonl | cut -c1-400; tail -5 $O/loop_masks.err
