#!/bin/bash
# Round 4, the last full GPU call: the whole GPU suite, smoke and the profiling recipe on the round's final tree
# (from the repository root: gpurun --timeout 3300 -- 'bash scripts/r04/final_call.sh').
set -u
R=$(pwd); O=$R/gpurun_out/r04h; mkdir -p $O; export TMPDIR=/tmp
python -c 'import oracle; oracle.build()' > $O/oracle_build.log 2>&1
( time timeout 1800 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider ) > $O/pytest_gpu.log 2>&1
tail -15 $O/pytest_gpu.log | cut -c1-300
( time python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; grep "smoke" $O/smoke.log | cut -c1-300
LITE=1 bash scripts/profile_round.sh r04 > $O/profile_round.log 2>&1; tail -6 $O/profile_round.log | cut -c1-400
cp $R/gpurun_out/prof_r04/bench_plain.log $O/bench_final.log 2>/dev/null
( time timeout 600 python bench.py --gpus 2 --share-gpu --nodes 1000000 --pairs 9500000 --steps 3 --warmup 1 --whiten-iters 3 ) > $O/bench_2rank_local.json 2> $O/bench_2rank_local.err
echo "2-rank rc=$?"
