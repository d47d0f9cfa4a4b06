#!/bin/bash
O=gpurun_out/r02l; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 300 python -m pytest tests/test_gpu_whiten.py tests/test_gpu_dropin.py tests/test_gpu_variants.py -m gpu -q --maxfail=10 --durations=5 ) > $O/pytest.log 2>&1; tail -6 $O/pytest.log
timeout 200 python scripts/overlap_loop_probe.py --c2 > $O/loop_c2.log 2>&1; tail -4 $O/loop_c2.log
timeout 400 python scripts/overlap_loop_probe.py > $O/loop_c3.log 2>&1; tail -4 $O/loop_c3.log
bash scripts/profile_round.sh r02 > $O/profile_round.log 2>&1; tail -15 $O/profile_round.log
