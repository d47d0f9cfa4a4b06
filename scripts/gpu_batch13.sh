#!/bin/bash
O=gpurun_out/r02m; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 300 python -m pytest tests/test_gpu_whiten.py tests/test_gpu_dropin.py -m gpu -q --maxfail=10 --durations=5 ) > $O/pytest.log 2>&1; tail -6 $O/pytest.log
timeout 200 python scripts/overlap_loop_probe.py --c2 > $O/loop_c2.log 2>&1; tail -4 $O/loop_c2.log
timeout 400 python scripts/overlap_loop_probe.py > $O/loop_c3.log 2>&1; tail -4 $O/loop_c3.log
cd /tmp && export TMPDIR=/tmp
for r in kernel library; do
  CLEORA_CHOLESKY=$r timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace_$r -o c2 -- python $GRAFT_REPO_ROOT/scripts/overlap_loop_probe.py --c2 > $GRAFT_REPO_ROOT/$O/trace_$r.log 2>&1
  head -25 $GRAFT_REPO_ROOT/$O/trace_$r/c2_kernel_stats.csv | cut -c1-200
done
cd $GRAFT_REPO_ROOT
find $O -type f ! -name "*_kernel_stats.csv" ! -name "*.log" -delete
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 3000 $O/bench.json
