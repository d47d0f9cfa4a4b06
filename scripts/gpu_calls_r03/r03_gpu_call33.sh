#!/bin/bash
# Round 3, GPU call 33: default bench on the final bench.py (marginal iteration cost of the whitened loop).
set -u
R=$(pwd)
O=$R/gpurun_out/r03last
mkdir -p $O
export TMPDIR=/tmp
python -c 'import oracle; oracle.build()' > $O/oracle_build.log 2>&1
timeout 600 python bench.py > $O/bench_final.log 2>&1
grep "^{" $O/bench_final.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic']); w=d['whitened']; print(w['ms_per_iter'], w['marginal_ms_per_iter'], w['kernels_ms']); print(d['cpu_baseline']['value'])"
tail -2 $O/bench_final.log | cut -c1-200
