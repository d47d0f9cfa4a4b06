#!/bin/bash
# Round 3, GPU call 18: kernel timeline of the whitened loop: which neighbour slows the SpMM (Gram / the d x d step's small kernels).
set -u
R=$(pwd)
O=$R/gpurun_out/r03r
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for v in "CLEORA_GRAM_CO_BLOCKS=1" "CLEORA_CHOLESKY=kernel" "CLEORA_SOLVE_AFTER_SPMM=1" "CLEORA_GRAM_CO_BLOCKS=-128 CLEORA_SOLVE_AFTER_SPMM=1" "CLEORA_GRAM_CO_BLOCKS=-128 CLEORA_CHOLESKY=kernel" "CLEORA_GRAM_CO_BLOCKS=-64 CLEORA_SOLVE_AFTER_SPMM=1"; do
  i=$((i+1))
  env $v timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$i -o loop -- python $R/scripts/r03_probe.py loop > $O/loop_$i.log 2>&1
  grep "^{" $O/loop_$i.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', 'whitened_ms_per_iter', round(d['whitened_ms_per_iter'],2), 'plain', round(d['spmm_rows_kernel_ms'],2))"
  python $R/scripts/loop_timeline.py $O/trace_$i "$v" | tee -a $O/timeline.jsonl | cut -c1-700
  rm -rf $O/trace_$i
done
