#!/bin/bash
# Round 3, GPU call 6: the f32 Gram at d = 512 / 1024 (off-diagonal super-tile blocks), config 5 with it.
set -u
R=$(pwd)
O=$R/gpurun_out/r03f
mkdir -p $O
export TMPDIR=/tmp
python -c 'import oracle; oracle.build()' > $O/oracle_build.log 2>&1
( time timeout 600 python -m pytest tests/test_gpu_whiten.py tests/test_gpu_parity_at_scale.py tests/test_gpu_dropin.py -m gpu -q --no-header -p no:cacheprovider --durations=5 ) > $O/pytest_new.log 2>&1
tail -14 $O/pytest_new.log
timeout 200 python scripts/r03_probe.py kernels 2000000 1024 > $O/kernels_d1024.json 2> $O/kernels_d1024.err; cat $O/kernels_d1024.json; tail -2 $O/kernels_d1024.err
timeout 200 python scripts/r03_probe.py kernels 4000000 512 > $O/kernels_d512.json 2> $O/kernels_d512.err; cat $O/kernels_d512.json
( time timeout 900 python bench.py --config C5 --steps 10 --warmup 2 --whiten-iters 3 ) > $O/bench_c5.log 2>&1
grep "^{" $O/bench_c5.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); w=d['whitened']; print(d['ms_per_step'], d['roofline']['frac']); print(w['ms_per_iter'], w['kernels_ms'], w['gram_intermediate_roofline'])"
