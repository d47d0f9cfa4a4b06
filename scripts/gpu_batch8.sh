#!/bin/bash
O=gpurun_out/r02h; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 900 python -m pytest tests/test_gpu_whiten.py tests/test_gpu_dropin.py tests/test_gpu_edge_semantics.py tests/test_gpu_variants.py tests/test_zz_c_host.py -m gpu -q --maxfail=10 ) > $O/pytest.log 2>&1; tail -30 $O/pytest.log
timeout 300 python scripts/overlap_loop_probe.py > $O/loop_chol.log 2>&1; grep -v amdgpu $O/loop_chol.log | sed "s/^/chol /"
CLEORA_WHITEN_PCA_ALWAYS=1 timeout 300 python scripts/overlap_loop_probe.py > $O/loop_pca.log 2>&1; grep -v amdgpu $O/loop_pca.log | sed "s/^/pca  /"
timeout 300 python scripts/overlap_loop_probe.py --c2 > $O/loop_c2_chol.log 2>&1; grep -v amdgpu $O/loop_c2_chol.log | sed "s/^/chol /"
CLEORA_WHITEN_PCA_ALWAYS=1 timeout 300 python scripts/overlap_loop_probe.py --c2 > $O/loop_c2_pca.log 2>&1; grep -v amdgpu $O/loop_c2_pca.log | sed "s/^/pca  /"
