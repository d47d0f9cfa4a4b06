#!/bin/bash
# Round 3, GPU call 26: statistics strictly BEFORE the SpMM (no co-running) against the overlapped order.
set -u
R=$(pwd)
O=$R/gpurun_out/r03z
mkdir -p $O
export TMPDIR=/tmp
( time timeout 420 python scripts/r03_probe.py loop_masks ) > $O/loop_order_c3.jsonl 2> $O/err1.log
cut -c1-330 $O/loop_order_c3.jsonl | sed 's/"mode": "loop_masks", "n": 9999997, "nnz": 199974337, "d": 256, //'; tail -3 $O/err1.log
( time timeout 300 python scripts/r03_probe.py loop_masks 1000000 10000000 256 ) > $O/loop_order_c2.jsonl 2> $O/err2.log
cut -c1-330 $O/loop_order_c2.jsonl | sed 's/"mode": "loop_masks", "n": 999994, "nnz": 19998418, "d": 256, //'; tail -3 $O/err2.log
