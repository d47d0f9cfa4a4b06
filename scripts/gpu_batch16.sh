#!/bin/bash
O=gpurun_out/r02p; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 1200 python -m pytest tests -x -q -m gpu --durations=12 ) > $O/pytest.log 2>&1; tail -22 $O/pytest.log
( time timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > $O/smoke.log 2>&1; tail -4 $O/smoke.log
