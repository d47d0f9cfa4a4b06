set -u
R=$(pwd); O=$R/gpurun_out/r04q; mkdir -p $O; export TMPDIR=/tmp
python -c 'import oracle; oracle.build()' > $O/oracle_build.log 2>&1
timeout 300 python scripts/r04/kernel_probe.py 10000000 256 > $O/probe_c3.json 2> $O/probe_c3.err
timeout 200 python scripts/r04/kernel_probe.py 2000000 1024 > $O/probe_d1024.json 2> $O/probe_d1024.err
( time timeout 600 python -m pytest tests/test_gpu_whiten.py -m gpu -q --no-header -p no:cacheprovider -x ) > $O/pytest_whiten.log 2>&1
cat $O/probe_c3.json $O/probe_d1024.json | cut -c1-700
tail -5 $O/pytest_whiten.log | cut -c1-300
