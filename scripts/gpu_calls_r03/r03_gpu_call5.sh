#!/bin/bash
# Round 3, GPU call 5: loop scheduling switches, the attention kernels after the norm prefetch, the profile recipe of the round.
set -u
R=$(pwd)
O=$R/gpurun_out/r03e
mkdir -p $O
export TMPDIR=/tmp
python -c 'import oracle; oracle.build()' > $O/oracle_build.log 2>&1
( time timeout 600 python -m pytest tests/test_gpu_variants.py tests/test_gpu_parity_at_scale.py -m gpu -q --no-header -p no:cacheprovider --durations=5 ) > $O/pytest_new.log 2>&1
tail -12 $O/pytest_new.log
timeout 300 python scripts/n34_probe.py > $O/n34.log 2>&1; head -4 $O/n34.log
for v in default "CLEORA_GRAM_CO_BLOCKS=2" "CLEORA_GRAM_FIRST=0"; do
  tag=$(echo "$v" | tr ' =' '__')
  if [ "$v" = default ]; then timeout 400 python scripts/r03_probe.py loop > $O/loop_$tag.json 2> $O/loop_$tag.err
  else env $v timeout 400 python scripts/r03_probe.py loop > $O/loop_$tag.json 2> $O/loop_$tag.err; fi
  cat $O/loop_$tag.json
done
LITE=1 bash scripts/profile_round.sh r03 > $O/profile_round.log 2>&1
tail -12 $O/profile_round.log | cut -c1-400
