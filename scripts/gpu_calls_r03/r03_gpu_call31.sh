#!/bin/bash
# Round 3, GPU call 31: whitening / parity / drop-in / C-host tests on the final build.
set -u
R=$(pwd)
O=$R/gpurun_out/r03last
mkdir -p $O
export TMPDIR=/tmp
python -c 'import oracle; oracle.build()' > $O/oracle_build.log 2>&1
( time timeout 900 python -m pytest tests/test_gpu_whiten.py tests/test_gpu_parity_at_scale.py tests/test_gpu_dropin.py tests/test_gpu_sharded.py tests/test_zz_c_host.py -m gpu -q --no-header -p no:cacheprovider ) > $O/pytest_sel.log 2>&1
tail -5 $O/pytest_sel.log
