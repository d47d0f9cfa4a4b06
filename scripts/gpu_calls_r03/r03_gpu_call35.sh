#!/bin/bash
# Round 3, GPU call 35 (last): the whole -m gpu suite and smoke() on the final tree.
set -u
R=$(pwd)
O=$R/gpurun_out/r03last
mkdir -p $O
export TMPDIR=/tmp
python -c 'import oracle; oracle.build()' > $O/oracle_build.log 2>&1
( time timeout 1200 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider --durations=5 ) > $O/pytest_all_final.log 2>&1
tail -12 $O/pytest_all_final.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
