#!/bin/bash
O=gpurun_out/r02d; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 900 python -m pytest tests/test_gpu_dropin.py tests/test_gpu_whiten.py tests/test_gpu_edge_semantics.py tests/test_gpu_variants.py tests/test_gpu_sharded.py tests/test_gpu_comm.py -m gpu -q --maxfail=10 ) > $O/pytest.log 2>&1; tail -25 $O/pytest.log
( time timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench1.log 2>&1; tail -c 4000 $O/bench1.log
hipcc --offload-arch=gfx950 -O3 scripts/mfma_peak.hip -o /tmp/mfma_peak 2>/dev/null && timeout 200 /tmp/mfma_peak > $O/mfma_peak.log 2>&1; grep mixed $O/mfma_peak.log
