#!/bin/bash
# Round 3, GPU call 4: 128-row projection tiles, the fused attention SpMM, the whole -m gpu suite with durations.
set -u
R=$(pwd)
O=$R/gpurun_out/r03d
mkdir -p $O
export TMPDIR=/tmp
python -c 'import oracle; oracle.build()' > $O/oracle_build.log 2>&1
CLEORA_PROJECT=split64 timeout 120 python scripts/r03_probe.py project >> $O/project_forms.jsonl 2>> $O/project.err
timeout 120 python scripts/r03_probe.py project >> $O/project_forms.jsonl 2>> $O/project.err
timeout 120 python scripts/r03_probe.py project 2000000 1024 >> $O/project_forms.jsonl 2>> $O/project.err
timeout 120 python scripts/r03_probe.py project 10000000 128 >> $O/project_forms.jsonl 2>> $O/project.err
cat $O/project_forms.jsonl; tail -3 $O/project.err
timeout 300 python scripts/n34_probe.py > $O/n34.log 2>&1; head -4 $O/n34.log
timeout 400 python scripts/r03_probe.py loop > $O/loop_default.json 2> $O/loop_default.err; cat $O/loop_default.json
( time timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider --durations=12 ) > $O/pytest_all.log 2>&1
tail -30 $O/pytest_all.log
