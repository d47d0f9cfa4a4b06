#!/bin/bash
# Round 4, GPU call 7: the other BASELINE workloads with the round's final kernels (config 2, config 5 at full size through the host
# builder, config 4's size on one GPU).
set -u
R=$(pwd); O=$R/gpurun_out/r04g; mkdir -p $O; export TMPDIR=/tmp
python -c 'import oracle; oracle.build()' > $O/oracle_build.log 2>&1
( time timeout 600 python bench.py --config C2 --whiten-iters 16 ) > $O/bench_c2.json 2> $O/bench_c2.err; echo "C2 rc=$?"
( time timeout 1200 python bench.py --config C5 --steps 10 --warmup 2 --whiten-iters 5 ) > $O/bench_c5.json 2> $O/bench_c5.err; echo "C5 rc=$?"
( time timeout 1200 python bench.py --config C4s --steps 6 --warmup 2 ) > $O/bench_c4s.json 2> $O/bench_c4s.err; echo "C4s rc=$?"
python - <<'PY'
import json
for c in ("c2", "c5", "c4s"):
    try:
        j = json.loads(open(f"gpurun_out/r04g/bench_{c}.json").read().strip().splitlines()[-1])
        w = j.get("whitened") or {}
        print(c, "n", j["config"]["n"], "nnz", j["config"]["nnz"], "d", j["config"]["d"], "ms/step", round(j["ms_per_step"], 3), "frac", round(j["roofline"]["frac"], 4),
              "checks", {k: v for k, v in j["checks"].items() if k != "drift_at_scale"}, "whitened", w.get("ms_per_iter"), "marginal", w.get("marginal_ms_per_iter"), w.get("kernels_ms"), w.get("skipped"),
              "cpu", (j.get("cpu_baseline") or {}).get("value"))
    except Exception as e:
        print(c, "unreadable", e)
PY
