#!/bin/bash
# the N > 1 code path of bench.py on the one-GPU box (developer mode: both partitions, gloo exchange), final sources
O=gpurun_out/r02t; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 170 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --share-gpu --backend gloo --steps 2 --warmup 1 --watchdog 150 ) > $O/share2.log 2>&1; grep -v "^W2026\|amdgpu" $O/share2.log | tail -c 1800
