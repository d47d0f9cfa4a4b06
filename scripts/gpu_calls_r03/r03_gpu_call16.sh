#!/bin/bash
# Round 3, GPU call 16: the whitened loop with the statistics on a PART of the chip (a grid of G blocks, one per CU) beside the SpMM.
set -u
R=$(pwd)
O=$R/gpurun_out/r03p
mkdir -p $O
export TMPDIR=/tmp
( time timeout 420 python scripts/r03_probe.py loop_masks ) > $O/loop_part_c3.jsonl 2> $O/loop_part_c3.err
cut -c1-330 $O/loop_part_c3.jsonl | sed 's/"mode": "loop_masks", //'; tail -3 $O/loop_part_c3.err
( time timeout 300 python scripts/r03_probe.py loop_masks 1000000 10000000 256 ) > $O/loop_part_c2.jsonl 2> $O/loop_part_c2.err
cut -c1-330 $O/loop_part_c2.jsonl | sed 's/"mode": "loop_masks", //'; tail -3 $O/loop_part_c2.err
