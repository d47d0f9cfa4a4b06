#!/bin/bash
# Round 4, GPU call 1: the pruned library under the whole GPU suite, the new 40-iteration parity test, kernel probes of the
# three-product Gram against the six-product one, the C3 bench line, and bench.py launching its own two ranks.
set -u
R=$(pwd); O=$R/gpurun_out/r04a; mkdir -p $O; export TMPDIR=/tmp
python -c 'import oracle; oracle.build()' > $O/oracle_build.log 2>&1
( time timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider ) > $O/pytest_gpu.log 2>&1
tail -40 $O/pytest_gpu.log | cut -c1-300
timeout 300 python scripts/r04/kernel_probe.py > $O/probe_lean.json 2> $O/probe_lean.err; cat $O/probe_lean.json
CLEORA_X_GRAM6=1 timeout 300 python scripts/r04/kernel_probe.py > $O/probe_six.json 2> $O/probe_six.err; cat $O/probe_six.json
timeout 300 python scripts/r04/kernel_probe.py 2000000 1024 > $O/probe_lean_d1024.json 2>> $O/probe_lean.err; cat $O/probe_lean_d1024.json
CLEORA_X_GRAM6=1 timeout 300 python scripts/r04/kernel_probe.py 2000000 1024 > $O/probe_six_d1024.json 2>> $O/probe_six.err; cat $O/probe_six_d1024.json
( time timeout 900 python bench.py --steps 20 --warmup 3 ) > $O/bench_c3.json 2> $O/bench_c3.err; cut -c1-1500 $O/bench_c3.json; tail -3 $O/bench_c3.err
CLEORA_X_GRAM6=1 timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_c3_six.json 2> $O/bench_c3_six.err
python - <<'PY'
import json
for f in ("bench_c3", "bench_c3_six"):
    try:
        j = json.loads(open(f"gpurun_out/r04a/{f}.json").read().strip().splitlines()[-1])
        w = j.get("whitened", {})
        print(f, "ms_per_step", round(j["ms_per_step"], 3), "frac", round(j["roofline"]["frac"], 4), "whitened ms/iter", w.get("ms_per_iter"), "marginal", w.get("marginal_ms_per_iter"), "kernels", w.get("kernels_ms"))
    except Exception as e:
        print(f, "unreadable", e)
PY
( time timeout 600 python bench.py --gpus 2 --share-gpu --backend gloo --nodes 1000000 --pairs 9500000 --steps 3 --warmup 1 ) > $O/bench_2rank_selflaunch.json 2> $O/bench_2rank_selflaunch.err
echo "self-launch rc=$?"; cut -c1-600 $O/bench_2rank_selflaunch.json; tail -5 $O/bench_2rank_selflaunch.err | cut -c1-300
