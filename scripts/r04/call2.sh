#!/bin/bash
# Round 4, GPU call 2: the row-partitioned loops and the peer-direct transport through the C ABI (two / three ranks sharing the GPU),
# the plain-C host over them, the Gram / projection tests after their tolerances were restated, bench.py at N = 1 and at 2 shared ranks.
set -u
R=$(pwd); O=$R/gpurun_out/r04b; mkdir -p $O; export TMPDIR=/tmp
python -c 'import oracle; oracle.build()' > $O/oracle_build.log 2>&1
( time timeout 900 python -m pytest tests/test_gpu_sharded_abi.py -m gpu -q --no-header -p no:cacheprovider -x ) > $O/pytest_sharded_abi.log 2>&1
tail -25 $O/pytest_sharded_abi.log | cut -c1-400
( time timeout 900 python -m pytest tests/test_gpu_whiten.py tests/test_gpu_comm.py tests/test_gpu_sharded.py -m gpu -q --no-header -p no:cacheprovider ) > $O/pytest_whiten_comm.log 2>&1
tail -12 $O/pytest_whiten_comm.log | cut -c1-300
( time timeout 1200 python -m pytest tests/test_zz_c_host.py -m gpu -q --no-header -p no:cacheprovider ) > $O/pytest_c_host.log 2>&1
tail -12 $O/pytest_c_host.log | cut -c1-300
( time timeout 600 python bench.py --gpus 2 --share-gpu --backend local --nodes 1000000 --pairs 9500000 --steps 3 --warmup 1 --whiten-iters 3 ) > $O/bench_2rank_local.json 2> $O/bench_2rank_local.err
echo "2-rank local rc=$?"; cut -c1-3000 $O/bench_2rank_local.json; tail -8 $O/bench_2rank_local.err | cut -c1-300
( time timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline ) > $O/bench_c3.json 2> $O/bench_c3.err
echo "bench rc=$?"; tail -3 $O/bench_c3.err | cut -c1-300
python - <<'PY'
import json
try:
    j = json.loads(open("gpurun_out/r04b/bench_c3.json").read().strip().splitlines()[-1])
    w = j.get("whitened", {})
    print("bench_c3 ms_per_step", round(j["ms_per_step"], 3), "frac", round(j["roofline"]["frac"], 4), "whitened", w.get("ms_per_iter"), "marginal", w.get("marginal_ms_per_iter"))
    print("parallelism:", j["config"]["parallelism"], "| placement:", j.get("placement_tuning"))
except Exception as e:
    print("bench_c3 unreadable", e)
PY
