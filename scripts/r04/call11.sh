#!/bin/bash
# Round 4, GPU call 11: the whitening tests and the d = 128 / d = 96 shapes after the projection lost two (ring, unroll) instantiations.
set -u
R=$(pwd); O=$R/gpurun_out/r04k; mkdir -p $O; export TMPDIR=/tmp
python -c 'import oracle; oracle.build()' > $O/oracle_build.log 2>&1
( time timeout 900 python -m pytest tests/test_gpu_whiten.py tests/test_gpu_dropin.py tests/test_gpu_variants.py -m gpu -q --no-header -p no:cacheprovider ) > $O/pytest.log 2>&1
tail -6 $O/pytest.log | cut -c1-300
timeout 300 python scripts/r04/kernel_probe.py 10000000 128 > $O/probe_d128.json 2> $O/probe.err; cat $O/probe_d128.json
