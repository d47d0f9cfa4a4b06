#!/bin/bash
O=gpurun_out/r02o; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 600 python -m pytest tests/test_gpu_dropin.py tests/test_gpu_comm.py tests/test_gpu_sharded.py -m gpu -q --maxfail=10 --durations=5 ) > $O/pytest.log 2>&1; tail -8 $O/pytest.log
timeout 300 python scripts/alloc_cost_probe.py > $O/alloc_cost.log 2>&1; grep -v amdgpu.ids $O/alloc_cost.log | tail -9
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
b=json.loads(open("gpurun_out/r02o/bench.json").read().strip().splitlines()[-1])
print(b["value"], b["ms_per_step"], b["roofline"]["frac"], b["roofline"]["traffic"], b["placement_tuning"], b["whitened"]["ms_per_iter"], b["whitened"]["sequential_ms_per_iter"])
PY
