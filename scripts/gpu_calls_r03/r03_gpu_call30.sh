#!/bin/bash
# Round 3, GPU call 30: after the d == 256 rule for the order of statistics and SpMM: whitening / parity tests, smoke, default bench.
set -u
R=$(pwd)
O=$R/gpurun_out/r03last
mkdir -p $O
export TMPDIR=/tmp
python -c 'import oracle; oracle.build()' > $O/oracle_build.log 2>&1
( time timeout 900 python -m pytest tests/test_gpu_whiten.py tests/test_gpu_parity_at_scale.py tests/test_gpu_embed.py tests/test_zz_c_host.py -m gpu -q --no-header -p no:cacheprovider ) > $O/pytest_sel.log 2>&1
tail -5 $O/pytest_sel.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
( time timeout 600 python bench.py ) > $O/bench.log 2>&1
tail -1 $O/bench.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic']); w=d['whitened']; print(w['ms_per_iter'], w['kernels_ms'])"
