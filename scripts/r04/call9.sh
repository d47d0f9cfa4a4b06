#!/bin/bash
# Round 4, GPU call 9: the two tests added after call 8 (bench.py launching its own two ranks on the shared GPU; the blended default loop
# at C2 size) and the communicator test with the peer transport on an RCCL communicator.
set -u
R=$(pwd); O=$R/gpurun_out/r04i; mkdir -p $O; export TMPDIR=/tmp
python -c 'import oracle; oracle.build()' > $O/oracle_build.log 2>&1
( time timeout 1200 python -m pytest tests/test_gpu_bench_launch.py tests/test_gpu_comm.py "tests/test_gpu_parity_at_scale.py::test_default_loop_with_a_residual_blend_at_c2_size" -m gpu -q --no-header -p no:cacheprovider ) > $O/pytest.log 2>&1
tail -15 $O/pytest.log | cut -c1-400
