#!/bin/bash
O=gpurun_out/r02j; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 600 python -m pytest tests/test_gpu_whiten.py tests/test_zz_c_host.py tests/test_gpu_dropin.py -m gpu -q --maxfail=10 --durations=8 ) > $O/pytest.log 2>&1; tail -25 $O/pytest.log
timeout 400 python scripts/overlap_loop_probe.py > $O/loop_c3.log 2>&1; cat $O/loop_c3.log | tail -6
timeout 200 python scripts/overlap_loop_probe.py --c2 > $O/loop_c2.log 2>&1; cat $O/loop_c2.log | tail -6
timeout 300 python scripts/n34_probe.py > $O/n34.log 2>&1; tail -12 $O/n34.log
timeout 60 rocprofv3 -L 2>/dev/null | grep -i -E "mall|hbm|dram|EA_RD|EA_WR|TCC_EA" | cut -c1-160 | head -40 > $O/counters.txt; wc -l $O/counters.txt; head -30 $O/counters.txt
