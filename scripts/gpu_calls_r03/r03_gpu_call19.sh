#!/bin/bash
# Round 3, GPU call 19: the d x d step on the host — whitening tests, loop timings at configs 3 and 2, kernel timeline.
set -u
R=$(pwd)
O=$R/gpurun_out/r03s
mkdir -p $O
export TMPDIR=/tmp
python -c 'import oracle; oracle.build()' > $O/oracle_build.log 2>&1
( time timeout 900 python -m pytest tests/test_gpu_whiten.py tests/test_gpu_parity_at_scale.py -m gpu -q --no-header -p no:cacheprovider -x ) > $O/pytest_whiten.log 2>&1
tail -6 $O/pytest_whiten.log
cd /tmp
i=0
for v in "CLEORA_CHOLESKY=host" "CLEORA_CHOLESKY=library" "CLEORA_CHOLESKY=host CLEORA_GRAM_CO_BLOCKS=-128" "CLEORA_CHOLESKY=host CLEORA_GRAM_CO_BLOCKS=2"; do
  i=$((i+1))
  env $v timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$i -o loop -- python $R/scripts/r03_probe.py loop > $O/loop_$i.log 2>&1
  grep "^{" $O/loop_$i.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', 'whitened_ms_per_iter', round(d['whitened_ms_per_iter'],2), 'plain', round(d['spmm_rows_kernel_ms'],2))"
  python $R/scripts/loop_timeline.py $O/trace_$i "$v" | tee -a $O/timeline.jsonl | cut -c1-600
  rm -rf $O/trace_$i
done
cd $R
for v in "CLEORA_CHOLESKY=host" "CLEORA_CHOLESKY=library" "CLEORA_CHOLESKY=host CLEORA_GRAM_CO_BLOCKS=2" "CLEORA_CHOLESKY=host CLEORA_GRAM_CO_BLOCKS=-128"; do
  env $v timeout 200 python scripts/r03_probe.py loop 1000000 10000000 256 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C2', '$v', 'whitened_ms_per_iter', round(d['whitened_ms_per_iter'],2), 'plain', round(d['spmm_rows_kernel_ms'],2))" | tee -a $O/c2_loops.txt
done
