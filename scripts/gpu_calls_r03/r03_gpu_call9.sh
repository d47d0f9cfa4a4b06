#!/bin/bash
# Round 3, GPU call 9: top-k selection from a short list (tests + timings).
set -u
R=$(pwd)
O=$R/gpurun_out/r03i
mkdir -p $O
export TMPDIR=/tmp
python -c 'import oracle; oracle.build()' > $O/oracle_build.log 2>&1
( time timeout 600 python -m pytest tests/test_gpu_variants.py -m gpu -q --no-header -p no:cacheprovider -x ) > $O/pytest_variants.log 2>&1
tail -30 $O/pytest_variants.log
timeout 300 python scripts/n34_probe.py > $O/n34.log 2>&1; cat $O/n34.log | cut -c1-200
CLEORA_TOPK=rounds timeout 300 python scripts/n34_probe.py > $O/n34_rounds.log 2>&1; grep "top-" $O/n34_rounds.log | cut -c1-120
