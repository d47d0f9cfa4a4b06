#!/bin/bash
# Round 3, GPU call 11: the plain SpMM confined to parts of the chip (CU masks).
set -u
R=$(pwd)
O=$R/gpurun_out/r03k
mkdir -p $O
export TMPDIR=/tmp
( time timeout 300 python scripts/r03_probe.py spmm_masks ) > $O/spmm_masks.jsonl 2> $O/spmm_masks.err
cat $O/spmm_masks.jsonl | cut -c1-300; tail -5 $O/spmm_masks.err
