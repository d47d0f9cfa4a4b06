#!/bin/bash
O=gpurun_out/r02r; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 300 python -m pytest tests/test_gpu_whiten.py tests/test_gpu_edge_scale.py -m gpu -q --maxfail=10 --durations=3 -k "not baseline_sizes" ) > $O/pytest.log 2>&1; tail -7 $O/pytest.log
timeout 200 python scripts/whiten_stage_probe.py > $O/stage.log 2>&1; grep -v amdgpu $O/stage.log | tail -6
timeout 300 python scripts/overlap_loop_probe.py > $O/loop_c3.log 2>&1; tail -3 $O/loop_c3.log
