#!/bin/bash
# rocprofv3 kernel stats of the DEFAULT (whitened) loop at C3 on the round's final kernels: what a marginal iteration is made of
# (from the repository root: gpurun --timeout 900 -- 'bash scripts/r04/whitened_loop_trace.sh'; summary -> profiles/r04_whitened_loop_kernel_stats.csv)
set -u
R=$(pwd); O=$R/gpurun_out/r04o; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o loop -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --whiten-iters 16 > $O/bench.log 2>&1
tail -1 $O/bench.log | cut -c1-300
find $O/trace -name "*_kernel_stats.csv" -exec cp {} $O/loop_kernel_stats.csv \;
find $O/trace -type f ! -name "*_kernel_stats.csv" -delete
grep cleora $O/loop_kernel_stats.csv | cut -c1-260 | head -30
