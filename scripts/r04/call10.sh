#!/bin/bash
# Round 4, GPU call 10: four ranks sharing the GPU through bench.py's own launcher (three peers per rank in the peer-direct transport);
# the sharded ABI test with the symmetric values and the L1 + whitening path added.
set -u
R=$(pwd); O=$R/gpurun_out/r04j; mkdir -p $O; export TMPDIR=/tmp
python -c 'import oracle; oracle.build()' > $O/oracle_build.log 2>&1
( time timeout 900 python -m pytest tests/test_gpu_sharded_abi.py -k world_of_one -m gpu -q --no-header -p no:cacheprovider ) > $O/pytest.log 2>&1
tail -8 $O/pytest.log | cut -c1-400
( time timeout 900 python bench.py --gpus 4 --share-gpu --nodes 1000000 --pairs 9500000 --steps 3 --warmup 1 --whiten-iters 3 ) > $O/bench_4rank_local.json 2> $O/bench_4rank_local.err
echo "4-rank rc=$?"; tail -4 $O/bench_4rank_local.err | cut -c1-300
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r04j/bench_4rank_local.json").read().strip().splitlines()[-1])
print(j["selftest"], j["config"]["per_iteration_ms"], j["config"]["ceiling"]["links"], {k: v["ms_per_step"] for k, v in j["partitions"].items()}, j.get("whitened_sharded"))
PY
