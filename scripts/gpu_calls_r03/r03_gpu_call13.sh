#!/bin/bash
# Round 3, GPU call 13: do the SpMM and the whitening kernels overlap at all (alone / both at once; plain and CU-masked streams)?
set -u
R=$(pwd)
O=$R/gpurun_out/r03m
mkdir -p $O
export TMPDIR=/tmp
( time timeout 420 python scripts/r03_probe.py overlap ) > $O/overlap.jsonl 2> $O/overlap.err
python - <<'PY'
import json
for l in open("gpurun_out/r03m/overlap.jsonl"):
    d = json.loads(l); print({k: (round(v, 2) if isinstance(v, float) else v) for k, v in d.items()})
PY
tail -4 $O/overlap.err
