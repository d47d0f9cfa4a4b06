#!/bin/bash
# kernel durations of one rank's blocks (plan_and_block_probe.py C3 --blocks): hub_products / hub_chain / hub_inorder / spmm_rows
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/gpurun_out/prof_r06
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/chain" -o chain -- python "$root/scripts/r06/plan_and_block_probe.py" ${1:-C3} --blocks > "$out/chain.log" 2>&1
f=$(find "$out/chain" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" "$out/chain_kernel_stats.csv" && head -12 "$f" | cut -c1-200
t=$(find "$out/chain" -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python - "$t" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
d = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"].split("(")[0][:60]
    d[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:12]:
    v2 = sorted(v)
    print(f"{k:60s} n={len(v):5d} median={v2[len(v2)//2]:9.1f} us max={v2[-1]:9.1f} us")
PY
find "$out/chain" -type f -delete
