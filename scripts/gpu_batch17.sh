#!/bin/bash
O=gpurun_out/r02q; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 300 python -m pytest tests/test_gpu_variants.py tests/test_gpu_dropin.py -m gpu -q --maxfail=10 --durations=5 ) > $O/pytest.log 2>&1; tail -12 $O/pytest.log
timeout 300 python scripts/n34_probe.py > $O/n34.log 2>&1; tail -16 $O/n34.log
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -o n34 -- python $GRAFT_REPO_ROOT/scripts/n34_probe.py > $GRAFT_REPO_ROOT/$O/trace.log 2>&1
head -16 $GRAFT_REPO_ROOT/$O/trace/n34_kernel_stats.csv | cut -c1-230
cd $GRAFT_REPO_ROOT; find $O -type f ! -name "*_kernel_stats.csv" ! -name "*.log" -delete
