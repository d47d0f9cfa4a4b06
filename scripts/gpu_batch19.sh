#!/bin/bash
# final refresh of the round's evidence: PMC passes (re-stamped for the current sources), kernel stats, then the bench line
O=gpurun_out/r02s; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
bash scripts/profile_round.sh r02 > $O/profile_round.log 2>&1
python scripts/summarize_profile.py r02 > $O/summarize.log 2>&1; tail -3 $O/summarize.log | cut -c1-200
timeout 200 python scripts/whiten_stage_probe.py > $O/stage.log 2>&1
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json
mkdir -p $O/profiles; cp profiles/hbm_traffic.json profiles/r02_*.json profiles/r02_kernel_stats.csv profiles/r02_whiten_kernel_stats.csv $O/profiles/
