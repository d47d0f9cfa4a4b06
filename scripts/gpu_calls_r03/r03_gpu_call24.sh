#!/bin/bash
# Round 3, GPU call 24: the bench's whitened loop under the route switches (is 52.7 ms the box, the host route, or the CPU baseline's threads?).
set -u
R=$(pwd)
O=$R/gpurun_out/r03x
mkdir -p $O
export TMPDIR=/tmp
python -c 'import oracle; oracle.build()' > $O/oracle_build.log 2>&1
run() {
  tag=$1; shift
  ( env "$@" timeout 400 python bench.py --steps 10 --warmup 2 $EXTRA ) > $O/bench_$tag.log 2>&1
  grep "^{" $O/bench_$tag.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); w=d['whitened']; print('$tag', 'spmm', round(d['ms_per_step'],2), 'whitened', round(w['ms_per_iter'],2), 'seq', round(w['sequential_ms_per_iter']['c_loop_reference_order'],2), {k: round(v,2) for k,v in w['kernels_ms'].items()})"
}
EXTRA="--no-cpu-baseline"; run host_nocpu CLEORA_CHOLESKY=host
EXTRA="--no-cpu-baseline"; run library_nocpu CLEORA_CHOLESKY=library
EXTRA=""; run host_cpu CLEORA_CHOLESKY=host
EXTRA=""; run library_cpu CLEORA_CHOLESKY=library
EXTRA="--no-cpu-baseline"; run host_nocpu_f32gram CLEORA_CHOLESKY=host CLEORA_GRAM=f32
EXTRA="--no-cpu-baseline"; run host_nocpu_again CLEORA_CHOLESKY=host
