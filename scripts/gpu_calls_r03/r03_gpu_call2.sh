#!/bin/bash
# Round 3, GPU call 2: the software-pipelined split projection, its profiling variants and counters, C5 at size, parity at scale.
set -u
R=$(pwd)
O=$R/gpurun_out/r03b
mkdir -p $O
export TMPDIR=/tmp
python -c 'import oracle; oracle.build()' > $O/oracle_build.log 2>&1
timeout 200 python scripts/r03_probe.py kernels > $O/kernels_default.json 2> $O/kernels_default.err; cat $O/kernels_default.json
for dbg in 1 2 3 4 7; do
  CLEORA_PROJECT_DEBUG=$dbg timeout 120 python scripts/r03_probe.py project >> $O/project_debug.jsonl 2>> $O/project_debug.err
done
cat $O/project_debug.jsonl
timeout 120 python scripts/r03_probe.py project 2000000 1024 >> $O/project_shapes.jsonl 2>> $O/project_debug.err
CLEORA_PROJECT=f32 timeout 120 python scripts/r03_probe.py project 2000000 1024 >> $O/project_shapes.jsonl 2>> $O/project_debug.err
timeout 120 python scripts/r03_probe.py project 10000000 128 >> $O/project_shapes.jsonl 2>> $O/project_debug.err
cat $O/project_shapes.jsonl
( time timeout 600 python -m pytest tests/test_gpu_whiten.py tests/test_gpu_parity_at_scale.py -m gpu -q --no-header -p no:cacheprovider ) > $O/pytest_new.log 2>&1
tail -8 $O/pytest_new.log
cd /tmp
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq -o pmc -- python $R/scripts/r03_probe.py project > $O/pmc_sq.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_inst -o pmc -- python $R/scripts/r03_probe.py project > $O/pmc_inst.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o pmc -- python $R/scripts/r03_probe.py project > $O/pmc_fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_write -o pmc -- python $R/scripts/r03_probe.py project > $O/pmc_write.log 2>&1
cd $R
for f in pmc_sq pmc_inst pmc_fetch pmc_write; do
  c=$(find $O/$f -name "*counter_collection.csv" | head -1)
  [ -n "$c" ] && { head -1 "$c" > $O/$f.csv; grep -E "project_split|pack_transform" "$c" >> $O/$f.csv; }
  rm -rf $O/$f
done
( time timeout 900 python bench.py --config C5 --steps 10 --warmup 2 --whiten-iters 3 ) > $O/bench_c5.log 2>&1
tail -2 $O/bench_c5.log | cut -c1-3000
( time timeout 600 python bench.py ) > $O/bench_c3.log 2>&1
tail -1 $O/bench_c3.log | cut -c1-3000
( time timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --share-gpu --backend gloo --config C2 --steps 2 --warmup 1 ) > $O/share2.log 2>&1
tail -3 $O/share2.log | cut -c1-2500
