#!/bin/bash
# Round 3, GPU call 34: gram16_kernel with the two waves of a SIMD in opposite phases (CLEORA_GRAM16_ORDER = 0 | 1 | 2).
set -u
R=$(pwd)
O=$R/gpurun_out/r03order
mkdir -p $O
export TMPDIR=/tmp
for v in 0 1 2; do
  CLEORA_GRAM16_ORDER=$v timeout 300 python scripts/r03_probe.py kernels 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('order $v', {k: (float('%.3g' % v) if isinstance(v,float) else v) for k,v in d.items() if 'stats_ms_intermediate1' in k or 'gram_f32' in k or 'diag_rel_err_mean' in k})" | tee -a $O/order.txt
done
for v in 0 1 2; do
  CLEORA_GRAM16_ORDER=$v timeout 300 python scripts/r03_probe.py kernels 2000000 1024 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('d1024 order $v', {k: (float('%.3g' % v) if isinstance(v,float) else v) for k,v in d.items() if 'stats_ms_intermediate1' in k or 'gram_f32' in k})" | tee -a $O/order.txt
done
( time timeout 600 python -m pytest tests/test_gpu_whiten.py -m gpu -q --no-header -p no:cacheprovider -x -k "intermediate_gram" ) > $O/pytest_gram.log 2>&1
tail -3 $O/pytest_gram.log
