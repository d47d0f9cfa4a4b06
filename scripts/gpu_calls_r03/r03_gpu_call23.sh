#!/bin/bash
# Round 3, GPU call 23: the whole -m gpu suite and the default bench after the split Gram / host d x d step / top-k short list.
set -u
R=$(pwd)
O=$R/gpurun_out/r03w
mkdir -p $O
export TMPDIR=/tmp
python -c 'import oracle; oracle.build()' > $O/oracle_build.log 2>&1
( time timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider --durations=8 ) > $O/pytest_all.log 2>&1
tail -22 $O/pytest_all.log
( time timeout 600 python bench.py ) > $O/bench.log 2>&1
grep "^{" $O/bench.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline'].get('traffic_note')); w=d['whitened']; print(w['ms_per_iter'], w['sequential_ms_per_iter'], w['kernels_ms']); print(w['gram_intermediate_roofline']); print(w['checks'])"
tail -3 $O/bench.log | cut -c1-300
