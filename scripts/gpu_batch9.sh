#!/bin/bash
O=gpurun_out/r02i; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 1500 python -m pytest tests -m gpu -q --maxfail=10 --durations=25 ) > $O/pytest.log 2>&1; tail -45 $O/pytest.log
