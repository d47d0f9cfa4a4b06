#!/bin/bash
# Round-2 GPU batch 2: full test suite, bench, whitening-stage A/B, MFMA ceilings, PMC pass, 4-rank shared-GPU repeat.
O=gpurun_out/r02b; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 1200 python -m pytest tests -m gpu -q --maxfail=10 ) > $O/pytest.log 2>&1; tail -25 $O/pytest.log
( time timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench1.log 2>&1; tail -c 3500 $O/bench1.log
timeout 300 python scripts/whiten_stage_probe.py > $O/stage_new.log 2>&1; cat $O/stage_new.log
CLEORA_PROJECT_TILED=1 timeout 300 python scripts/whiten_stage_probe.py > $O/stage_tiled.log 2>&1; cat $O/stage_tiled.log
hipcc --offload-arch=gfx950 -O3 scripts/mfma_peak.hip -o /tmp/mfma_peak 2>/dev/null && timeout 200 /tmp/mfma_peak > $O/mfma_peak.log 2>&1; cat $O/mfma_peak.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -iE "^\s*(Name|Counter).*(MFMA|SQ_WAIT|SQ_BUSY|SQ_WAVE_CYCLES|SQ_ACTIVE_INST|LDS_BANK|LDS_IDX|SQ_INSTS_VALU|SQ_INST_CYCLES)" | head -60 > $GRAFT_REPO_ROOT/$O/counters.txt 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc1 -o pmc -- python $GRAFT_REPO_ROOT/scripts/whiten_stage_probe.py 1 > $GRAFT_REPO_ROOT/$O/pmc1.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, collections, glob
for f in glob.glob("gpurun_out/r02b/pmc1/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in acc.items():
        if "gram" in k or "project" in k:
            print(k, {n: round(sum(v) / len(v)) for n, v in c.items()})
PY
find $O/pmc1 -type f ! -name "*counter_collection.csv" -delete 2>/dev/null
( time timeout 170 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 4 --share-gpu --backend gloo --partition column --steps 2 --warmup 1 --watchdog 150 ) > $O/share4.log 2>&1; tail -c 800 $O/share4.log
