set -u
R=$(pwd); O=$R/gpurun_out/r04l; mkdir -p $O; export TMPDIR=/tmp
python -c 'import oracle; oracle.build()' > $O/oracle_build.log 2>&1
( time timeout 1800 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider --durations=25 ) > $O/pytest_gpu.log 2>&1
tail -45 $O/pytest_gpu.log | cut -c1-200
( time python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; grep "smoke" $O/smoke.log | cut -c1-300
( time timeout 600 python bench.py --steps 20 --warmup 3 ) > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
