#!/bin/bash
O=gpurun_out/r02g; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
for cfg in "1 1" "1 2" "0 1"; do set -- $cfg; CLEORA_GRAM_FIRST=$1 CLEORA_GRAM_CO_BLOCKS=$2 timeout 300 python scripts/overlap_loop_probe.py > $O/loop_$1_$2.log 2>&1; grep -v amdgpu $O/loop_$1_$2.log | sed "s/^/gram_first=$1 /"; done
CLEORA_GRAM_FIRST=1 timeout 300 python scripts/overlap_loop_probe.py --c2 > $O/loop_c2.log 2>&1; grep -v amdgpu $O/loop_c2.log
