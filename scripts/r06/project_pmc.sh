#!/bin/bash
# Round 6: issue / wait / instruction-mix counters of the projection kernels at the C3 shape (scripts/r06/whiten_kernels_probe.py).
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/gpurun_out/prof_r06_proj
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o whiten -- python "$root/scripts/r06/whiten_kernels_probe.py" > "$out/stats.log" 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$out/p1" -o pmc -- python "$root/scripts/r06/whiten_kernels_probe.py" > "$out/p1.log" 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d "$out/p2" -o pmc -- python "$root/scripts/r06/whiten_kernels_probe.py" > "$out/p2.log" 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --output-format csv -d "$out/p3" -o pmc -- python "$root/scripts/r06/whiten_kernels_probe.py" > "$out/p3.log" 2>&1
find "$out" -type f ! -name "*_kernel_stats.csv" ! -name "pmc_counter_collection.csv" ! -name "*.log" -delete
python3 - "$out" <<'P'
import csv, sys, collections, json, os, re
out = sys.argv[1]
res = collections.defaultdict(lambda: collections.defaultdict(list))
for p in ("p1", "p2", "p3"):
    for dp, _, fs in os.walk(os.path.join(out, p)):
        for f in fs:
            if f.endswith("pmc_counter_collection.csv"):
                for r in csv.DictReader(open(os.path.join(dp, f))):
                    k = r["Kernel_Name"]
                    m = re.search(r"(\w+_kernel(?:IL\w+)?(?:<[^>]*>)?)", k)
                    if m and ("project" in k or "gram16" in k):
                        res[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
summary = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in res.items()}
json.dump(summary, open(os.path.join(out, "summary.json"), "w"), indent=1)
for k, cs in summary.items():
    print(k, {c: f"{v:.3g}" for c, v in cs.items()})
P
find "$out" -name "pmc_counter_collection.csv" -delete
du -sh "$out"
