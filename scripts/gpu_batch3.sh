#!/bin/bash
O=gpurun_out/r02c; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 600 python -m pytest tests/test_gpu_dropin.py tests/test_gpu_whiten.py tests/test_gpu_spmm.py tests/test_gpu_edge_semantics.py tests/test_gpu_variants.py -m gpu -q --maxfail=10 ) > $O/pytest.log 2>&1; tail -15 $O/pytest.log
timeout 300 python scripts/whiten_stage_probe.py > $O/stage_new.log 2>&1; grep -v amdgpu $O/stage_new.log
timeout 400 python scripts/hostptr_probe.py > $O/hostptr.log 2>&1; grep -v amdgpu $O/hostptr.log
