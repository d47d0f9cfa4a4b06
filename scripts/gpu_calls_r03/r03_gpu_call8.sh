#!/bin/bash
# Round 3, GPU call 8: bench at config 4's size on one GPU (post-loop checks in row slabs).
set -u
R=$(pwd)
O=$R/gpurun_out/r03h
mkdir -p $O
export TMPDIR=/tmp
python -c 'import oracle; oracle.build()' > $O/oracle_build.log 2>&1
( time timeout 900 python bench.py --config C4s --steps 6 --warmup 2 ) > $O/bench_c4s.log 2>&1
grep "^{" $O/bench_c4s.log | cut -c1-3500; tail -5 $O/bench_c4s.log | cut -c1-400
