#!/bin/bash
# Round 3, GPU call 15: what a co-resident kernel costs the SpMM (synthetic side kernels).
set -u
R=$(pwd)
O=$R/gpurun_out/r03o
mkdir -p $O
export TMPDIR=/tmp
( time timeout 420 python scripts/r03_probe.py side_load ) > $O/side_load.jsonl 2> $O/side_load.err
python - <<'PY'
import json
for l in open("gpurun_out/r03o/side_load.jsonl"):
    d = json.loads(l); print({k: (round(v, 2) if isinstance(v, float) else v) for k, v in d.items() if k != "mode"})
PY
tail -4 $O/side_load.err
