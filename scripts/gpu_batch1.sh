#!/bin/bash
# Round-2 GPU batch 1: correctness of the refactor, the new bench line, and the probes that decide the next steps.
O=gpurun_out/r02a; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 900 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; tail -5 $O/pytest.log
( time timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench1.log 2>&1; tail -c 3000 $O/bench1.log
hipcc --offload-arch=gfx950 -O3 scripts/mfma_peak.hip -o /tmp/mfma_peak 2>/dev/null && timeout 120 /tmp/mfma_peak > $O/mfma_peak.log 2>&1; cat $O/mfma_peak.log
timeout 200 python scripts/eigh_graph_probe.py > $O/eigh_graph.log 2>&1; cat $O/eigh_graph.log
timeout 400 python scripts/placement_probe.py > $O/placement.log 2>&1; cat $O/placement.log
( time timeout 170 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --share-gpu --backend gloo --steps 2 --warmup 1 --watchdog 150 ) > $O/share2.log 2>&1; tail -c 2500 $O/share2.log
( time timeout 170 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 4 --share-gpu --backend gloo --partition column --steps 2 --warmup 1 --watchdog 150 ) > $O/share4.log 2>&1; tail -c 1500 $O/share4.log
