set -u
R=$(pwd); O=$R/gpurun_out/r04u; mkdir -p $O; export TMPDIR=/tmp
python -c 'import oracle; oracle.build()' > $O/oracle_build.log 2>&1
( time timeout 500 python -m pytest tests/test_gpu_sharded_abi.py tests/test_gpu_comm.py tests/test_zz_c_host.py -m gpu -q --no-header -p no:cacheprovider ) > $O/pytest.log 2>&1
tail -6 $O/pytest.log | cut -c1-300
