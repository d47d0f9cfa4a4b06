#!/bin/bash
# Round 3, GPU call 20: the split-bf16 Gram (gram16_kernel): accuracy tests, kernel timings, the loop.
set -u
R=$(pwd)
O=$R/gpurun_out/r03t
mkdir -p $O
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_gpu_whiten.py -m gpu -q --no-header -p no:cacheprovider -x -k "intermediate_gram" ) > $O/pytest_gram.log 2>&1
tail -15 $O/pytest_gram.log
for v in "CLEORA_GRAM=split" "CLEORA_GRAM=f32"; do
  env $v timeout 300 python scripts/r03_probe.py kernels 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', {k: round(v,3) if isinstance(v,float) and abs(v)>1e-3 else v for k,v in d.items() if 'stats' in k or 'gram' in k or 'mean' in k})" | tee -a $O/kernels.txt
  env $v timeout 300 python scripts/r03_probe.py kernels 2000000 1024 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('d1024 $v', {k: round(v,3) if isinstance(v,float) and abs(v)>1e-3 else v for k,v in d.items() if 'stats' in k or 'gram' in k or 'mean' in k})" | tee -a $O/kernels.txt
  env $v timeout 300 python scripts/r03_probe.py loop 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', 'whitened_ms_per_iter', round(d['whitened_ms_per_iter'],2), 'plain', round(d['spmm_rows_kernel_ms'],2), d['whitened_cov_minus_identity'])" | tee -a $O/loops.txt
done
