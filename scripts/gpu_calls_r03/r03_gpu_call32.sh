#!/bin/bash
# Round 3, GPU call 32: the N > 1 code path of bench.py once more on the final build (two ranks sharing the one GPU over gloo, config 2).
set -u
R=$(pwd)
O=$R/gpurun_out/r03last
mkdir -p $O
export TMPDIR=/tmp
( time timeout 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --share-gpu --backend gloo --config C2 --steps 2 --warmup 1 ) > $O/share2.log 2>&1
grep "^{" $O/share2.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['n_gpus'], d['config']['partition'], d['config']['collectives'], round(d['ms_per_step'],2), d['selftest']); print({k: round(v['ms_per_step'],2) for k,v in d['partitions'].items()})"
tail -4 $O/share2.log | cut -c1-300
