#!/bin/bash
O=gpurun_out/r02f; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 600 python -m pytest tests/test_gpu_variants.py tests/test_gpu_whiten.py -m gpu -q --maxfail=10 ) > $O/pytest.log 2>&1; tail -12 $O/pytest.log
timeout 300 python scripts/overlap_loop_probe.py > $O/loop1.log 2>&1; grep -v amdgpu $O/loop1.log
CLEORA_GRAM_CO_BLOCKS=2 timeout 300 python scripts/overlap_loop_probe.py > $O/loop2.log 2>&1; grep -v amdgpu $O/loop2.log
timeout 300 python scripts/overlap_loop_probe.py --c2 > $O/loop_c2.log 2>&1; grep -v amdgpu $O/loop_c2.log
for dbg in 1 2 3; do CLEORA_PROJECT_DEBUG=$dbg timeout 200 python scripts/whiten_stage_probe.py 1 > $O/stage_dbg$dbg.log 2>&1; grep -v amdgpu $O/stage_dbg$dbg.log | sed "s/^/dbg=$dbg /"; done
