#!/bin/bash
# Round 3, GPU call 12: the whitened loop with statistics and SpMM on interleaved halves / quarters of the chip.
set -u
R=$(pwd)
O=$R/gpurun_out/r03l
mkdir -p $O
export TMPDIR=/tmp
( time timeout 420 python scripts/r03_probe.py loop_masks ) > $O/loop_masks_c3.jsonl 2> $O/loop_masks_c3.err
cut -c1-330 $O/loop_masks_c3.jsonl | sed 's/"mode": "loop_masks", //'; tail -3 $O/loop_masks_c3.err
( time timeout 300 python scripts/r03_probe.py loop_masks 1000000 10000000 256 ) > $O/loop_masks_c2.jsonl 2> $O/loop_masks_c2.err
cut -c1-330 $O/loop_masks_c2.jsonl | sed 's/"mode": "loop_masks", //'; tail -3 $O/loop_masks_c2.err
( time timeout 300 python scripts/r03_probe.py spmm_masks ) > $O/spmm_masks.jsonl 2> $O/spmm_masks.err
cut -c1-200 $O/spmm_masks.jsonl | sed 's/"mode": "spmm_masks", "n": 9999997, "nnz": 199974337, "d": 256, //'
