#!/bin/bash
# Round 3, GPU call 29: config 5 (d = 1024): the four combinations of Gram form and order of statistics / SpMM.
set -u
R=$(pwd)
O=$R/gpurun_out/r03c5b
mkdir -p $O
export TMPDIR=/tmp
for v in "CLEORA_STATS_BEFORE_SPMM=0" "CLEORA_STATS_BEFORE_SPMM=1" "CLEORA_STATS_BEFORE_SPMM=0 CLEORA_GRAM=f32" "CLEORA_STATS_BEFORE_SPMM=0 CLEORA_GRAM_CO_BLOCKS=-128"; do
  tag=$(echo "$v" | tr ' =' '__')
  ( env $v timeout 600 python bench.py --config C5 --steps 3 --warmup 1 --whiten-iters 5 --no-cpu-baseline ) > $O/bench_$tag.log 2>&1
  grep "^{" $O/bench_$tag.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); w=d['whitened']; print('$v', round(w['ms_per_iter'],2), {k: round(x,1) for k,x in w['kernels_ms'].items()})" | tee -a $O/summary.txt
done
