#!/bin/bash
O=gpurun_out/r02n; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
CLEORA_TUNE_TRACE=1 timeout 300 python scripts/alloc_cost_probe.py > $O/alloc_cost.log 2>&1; grep -v amdgpu.ids $O/alloc_cost.log | tail -40
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/col -o col -- python $GRAFT_REPO_ROOT/scripts/column_probe.py > $GRAFT_REPO_ROOT/$O/column.log 2>&1
grep -v "^W2026\|^E2026\|amdgpu.ids" $GRAFT_REPO_ROOT/$O/column.log | tail -8
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/c4 -o c4 -- python $GRAFT_REPO_ROOT/scripts/c4_probe.py > $GRAFT_REPO_ROOT/$O/c4.log 2>&1
grep -v "^W2026\|^E2026\|amdgpu.ids" $GRAFT_REPO_ROOT/$O/c4.log | tail -4
cd $GRAFT_REPO_ROOT
find $O -type f ! -name "*_kernel_stats.csv" ! -name "*.log" -delete
