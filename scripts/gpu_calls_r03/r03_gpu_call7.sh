#!/bin/bash
# Round 3, GPU call 7: attention kernels with the reduce-scatter reduction; bench at config 2 and at config 4's size on one GPU.
set -u
R=$(pwd)
O=$R/gpurun_out/r03g
mkdir -p $O
export TMPDIR=/tmp
python -c 'import oracle; oracle.build()' > $O/oracle_build.log 2>&1
( time timeout 400 python -m pytest tests/test_gpu_variants.py -m gpu -q --no-header -p no:cacheprovider ) > $O/pytest_variants.log 2>&1
tail -5 $O/pytest_variants.log
timeout 300 python scripts/n34_probe.py > $O/n34.log 2>&1; head -4 $O/n34.log
( time timeout 300 python bench.py --config C2 --steps 40 --warmup 3 ) > $O/bench_c2.log 2>&1
grep "^{" $O/bench_c2.log | cut -c1-1500
( time timeout 900 python bench.py --config C4s --steps 6 --warmup 2 ) > $O/bench_c4s.log 2>&1
grep "^{" $O/bench_c4s.log | cut -c1-3500; tail -5 $O/bench_c4s.log | cut -c1-400
