#!/bin/bash
# Round 3, GPU call 36: bench at config 2 with the final loop.
set -u
R=$(pwd)
O=$R/gpurun_out/r03last
mkdir -p $O
export TMPDIR=/tmp
python -c 'import oracle; oracle.build()' > $O/oracle_build.log 2>&1
timeout 200 python bench.py --config C2 --steps 40 --warmup 3 > $O/bench_c2.log 2>&1
grep "^{" $O/bench_c2.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'], d['checks']['oracle_rows_bit_equal']); w=d['whitened']; print(w['iterations'], w['ms_per_iter'], w['marginal_ms_per_iter'], w['sequential_ms_per_iter'], w['kernels_ms'])"
