#!/bin/bash
# Round 3, GPU call 28: BASELINE config 5 at full size with the final loop (split-bf16 Gram before the SpMM at d = 1024).
set -u
R=$(pwd)
O=$R/gpurun_out/r03c5
mkdir -p $O
export TMPDIR=/tmp
python -c 'import oracle; oracle.build()' > $O/oracle_build.log 2>&1
( time timeout 900 python bench.py --config C5 --steps 10 --warmup 2 --whiten-iters 3 ) > $O/bench_c5.log 2>&1
grep "^{" $O/bench_c5.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'], d['checks']); w=d['whitened']; print(w['ms_per_iter'], w['sequential_ms_per_iter'], w['kernels_ms']); print(w['gram_intermediate_roofline']); print(w['checks'])"
tail -4 $O/bench_c5.log | cut -c1-300
( time CLEORA_STATS_BEFORE_SPMM=0 CLEORA_GRAM=f32 timeout 900 python bench.py --config C5 --steps 4 --warmup 1 --whiten-iters 3 --no-cpu-baseline ) > $O/bench_c5_r03early.log 2>&1
grep "^{" $O/bench_c5_r03early.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); w=d['whitened']; print('f32 gram, overlapped order:', w['ms_per_iter'], w['kernels_ms'])"
