#!/bin/bash
# Round 3, GPU call 21: where the split-bf16 Gram's error sits (diagonal bias? accumulation length?).
set -u
R=$(pwd)
O=$R/gpurun_out/r03u
mkdir -p $O
export TMPDIR=/tmp
for v in "CLEORA_GRAM=split" "CLEORA_GRAM=split CLEORA_GRAM_SUB_ROWS=512" "CLEORA_GRAM=split CLEORA_GRAM_SUB_ROWS=128" "CLEORA_GRAM=split CLEORA_GRAM_SUB_ROWS=32" "CLEORA_GRAM=f32"; do
  env $v timeout 300 python scripts/r03_probe.py kernels 4000000 256 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', {k: (float('%.3g' % v) if isinstance(v,float) else v) for k,v in d.items() if 'stats_ms_intermediate1' in k or 'gram' in k})" | tee -a $O/gram_error.txt
done
