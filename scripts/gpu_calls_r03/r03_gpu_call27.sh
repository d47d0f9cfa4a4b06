#!/bin/bash
# Round 3, GPU call 27 (final): the whole -m gpu suite, then the round's profile recipe (kernel stats, HBM traffic PMC, whitening PMC, default bench).
set -u
R=$(pwd)
O=$R/gpurun_out/r03final
mkdir -p $O
export TMPDIR=/tmp
python -c 'import oracle; oracle.build()' > $O/oracle_build.log 2>&1
( time timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider --durations=6 ) > $O/pytest_all.log 2>&1
tail -14 $O/pytest_all.log
LITE=1 bash scripts/profile_round.sh r03 > $O/profile_round.log 2>&1
tail -8 $O/profile_round.log | cut -c1-400
