set -u
R=$(pwd); O=$R/gpurun_out/r04m; mkdir -p $O; export TMPDIR=/tmp
python -c 'import oracle; oracle.build()' > $O/oracle_build.log 2>&1
( time timeout 900 python -m pytest tests/test_gpu_whiten.py tests/test_gpu_sharded.py tests/test_gpu_sharded_abi.py tests/test_gpu_dropin.py tests/test_gpu_edge_semantics.py -m gpu -q --no-header -p no:cacheprovider ) > $O/pytest.log 2>&1
tail -6 $O/pytest.log | cut -c1-300
timeout 300 python scripts/r04/kernel_probe.py > $O/probe_c3.json 2> $O/probe.err; cat $O/probe_c3.json
timeout 300 python scripts/r04/kernel_probe.py 2000000 1024 > $O/probe_d1024.json 2>> $O/probe.err; cat $O/probe_d1024.json
timeout 300 python scripts/r04/kernel_probe.py 10000000 128 > $O/probe_d128.json 2>> $O/probe.err; cat $O/probe_d128.json
timeout 300 python scripts/r04/kernel_probe.py 1000003 256 > $O/probe_ragged.json 2>> $O/probe.err; cat $O/probe_ragged.json
( time timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --whiten-iters 16 ) > $O/bench_c3.json 2> $O/bench_c3.err
python - <<'PY'
import json
try:
    j = json.loads(open("gpurun_out/r04m/bench_c3.json").read().strip().splitlines()[-1])
    w = j.get("whitened", {})
    print("bench_c3 ms_per_step", round(j["ms_per_step"], 3), "frac", round(j["roofline"]["frac"], 4), "whitened(16)", w.get("ms_per_iter"), "marginal", w.get("marginal_ms_per_iter"), w.get("kernels_ms"))
except Exception as e:
    print("bench_c3 unreadable", e)
PY
