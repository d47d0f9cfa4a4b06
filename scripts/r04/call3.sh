#!/bin/bash
# Round 4, GPU call 3: the projection kernel with the A split shared between the two column-half waves: tests (results must not move)
# and timings at the C3 / C5 shapes.
set -u
R=$(pwd); O=$R/gpurun_out/r04c; mkdir -p $O; export TMPDIR=/tmp
python -c 'import oracle; oracle.build()' > $O/oracle_build.log 2>&1
( time timeout 900 python -m pytest tests/test_gpu_whiten.py tests/test_gpu_sharded.py -m gpu -q --no-header -p no:cacheprovider ) > $O/pytest_whiten.log 2>&1
tail -12 $O/pytest_whiten.log | cut -c1-300
timeout 300 python scripts/r04/kernel_probe.py > $O/probe_c3.json 2> $O/probe.err; cat $O/probe_c3.json
timeout 300 python scripts/r04/kernel_probe.py 2000000 1024 > $O/probe_d1024.json 2>> $O/probe.err; cat $O/probe_d1024.json
timeout 300 python scripts/r04/kernel_probe.py 10000000 128 > $O/probe_d128.json 2>> $O/probe.err; cat $O/probe_d128.json
timeout 300 python scripts/r04/kernel_probe.py 1000000 256 > $O/probe_c2.json 2>> $O/probe.err; cat $O/probe_c2.json
( time timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline ) > $O/bench_c3.json 2> $O/bench_c3.err
python - <<'PY'
import json
try:
    j = json.loads(open("gpurun_out/r04c/bench_c3.json").read().strip().splitlines()[-1])
    w = j.get("whitened", {})
    print("bench_c3 ms_per_step", round(j["ms_per_step"], 3), "frac", round(j["roofline"]["frac"], 4), "untuned", j["roofline"].get("frac_untuned"), "whitened", w.get("ms_per_iter"), "marginal", w.get("marginal_ms_per_iter"), w.get("kernels_ms"))
except Exception as e:
    print("bench_c3 unreadable", e)
PY
