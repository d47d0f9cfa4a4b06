# the round's GPU suite with durations; run on the GPU box through gpurun
set -u
R=$(pwd); O=$R/gpurun_out/${1:-r05s}; mkdir -p $O; export TMPDIR=/tmp
python -c 'import oracle; oracle.build()' > $O/oracle_build.log 2>&1
( time timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider --durations=25 ${2:-} ) > $O/pytest_gpu.log 2>&1
tail -45 $O/pytest_gpu.log | cut -c1-220
