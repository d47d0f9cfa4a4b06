#!/bin/bash
# Round 3, GPU call 3: the 128-row one-wave-per-SIMD projection against the 64-row form; tests; the 2-rank developer run.
set -u
R=$(pwd)
O=$R/gpurun_out/r03c
mkdir -p $O
export TMPDIR=/tmp
python -c 'import oracle; oracle.build()' > $O/oracle_build.log 2>&1
timeout 200 python scripts/r03_probe.py kernels > $O/kernels_default.json 2> $O/kernels_default.err; cat $O/kernels_default.json; tail -2 $O/kernels_default.err
CLEORA_PROJECT=split64 timeout 120 python scripts/r03_probe.py project >> $O/project_forms.jsonl 2>> $O/project.err
timeout 120 python scripts/r03_probe.py project >> $O/project_forms.jsonl 2>> $O/project.err
timeout 120 python scripts/r03_probe.py project 2000000 1024 >> $O/project_forms.jsonl 2>> $O/project.err
timeout 120 python scripts/r03_probe.py project 10000000 128 >> $O/project_forms.jsonl 2>> $O/project.err
cat $O/project_forms.jsonl; tail -3 $O/project.err
( time timeout 900 python -m pytest tests/test_gpu_whiten.py tests/test_gpu_parity_at_scale.py tests/test_gpu_edge_scale.py::test_whitening_at_c2_size_against_numpy_oracle tests/test_gpu_variants.py -m gpu -q --no-header -p no:cacheprovider --durations=8 ) > $O/pytest_new.log 2>&1
tail -22 $O/pytest_new.log
timeout 400 python scripts/r03_probe.py loop > $O/loop_default.json 2> $O/loop_default.err; cat $O/loop_default.json
cd /tmp
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq -o pmc -- python $R/scripts/r03_probe.py project > $O/pmc_sq.log 2>&1
cd $R
c=$(find $O/pmc_sq -name "*counter_collection.csv" | head -1)
[ -n "$c" ] && { head -1 "$c" > $O/pmc_sq.csv; grep -E "project_split" "$c" >> $O/pmc_sq.csv; }
rm -rf $O/pmc_sq
( time timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --share-gpu --backend gloo --config C2 --steps 2 --warmup 1 ) > $O/share2.log 2>&1
grep -v "^$" $O/share2.log | tail -4 | cut -c1-2500
