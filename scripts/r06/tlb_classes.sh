#!/bin/bash
# Round 6: address-translation / L2 counters of the SpMM per dispatch while scripts/r06/placement_probe.py walks through plain and
# VMM-mapped iterates (which fall into the fast and the slow "placement class"): does the slow class show more UTCL1 misses?
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/gpurun_out/prof_r06
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
TLB="TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCC_HIT_sum TCC_MISS_sum GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE"
timeout 900 rocprofv3 --pmc $TLB --output-format csv -d "$out/tlb_classes" -o pmc -- python "$root/scripts/r06/placement_probe.py" ${1:-C3} vmm > "$out/tlb_classes.log" 2>&1
tail -1 "$out/tlb_classes.log" | cut -c1-2000
f=$(find "$out/tlb_classes" -name "*counter_collection.csv" | head -1)
if [ -n "$f" ]; then head -1 "$f" > "$out/tlb_classes_spmm.csv"; grep spmm_rows_kernel "$f" >> "$out/tlb_classes_spmm.csv"; fi
find "$out/tlb_classes" -type f -delete
du -sh "$out"
