#!/bin/bash
# Profiling recipe for one round (run on the GPU box through gpurun):
#   [LITE=1] scripts/profile_round.sh r01   ->  (LITE: without the raw-request, hit-rate and config-2 passes)
#   scripts/profile_round.sh r01   ->  gpurun_out/prof_r01/{stats,wstats,fetch,write,fetch_nohot,hit,hit_nohot}
# then, back in the container, scripts/summarize_profile.py r01 condenses it into profiles/.
# Counters are collected in passes of their own with nothing but --pmc (no trace domains).
tag=${1:-r01}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o bench -- python "$root/bench.py" --no-cpu-baseline --whiten-iters 0 > "$out/bench_stats.log" 2>&1
# whitening kernels: cleora_project_dev (plain and loop form) + the statistics in both forms (f64 Gram; the split-bf16 Gram of the loop's intermediate iterations)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/wstats" -o whiten -- python "$root/scripts/r04/kernel_probe.py" > "$out/whiten_stats.log" 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/fetch" -o pmc -- python "$root/scripts/pmc_probe.py" > "$out/fetch.log" 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/write" -o pmc -- python "$root/scripts/pmc_probe.py" > "$out/write.log" 2>&1
# cross-check of FETCH_SIZE against the raw L2 -> fabric request counters it derives from (and how many of them go to DRAM
# rather than to a peer / the host); round 1 also ran the policy-off passes (fetch_nohot, hit_nohot: profiles/r01_pmc.json)
if [ -z "${LITE:-}" ]; then
timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum --output-format csv -d "$out/ea" -o pmc -- python "$root/scripts/pmc_probe.py" > "$out/ea.log" 2>&1
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$out/hit" -o pmc -- python "$root/scripts/pmc_probe.py" > "$out/hit.log" 2>&1
fi
# whitening kernels: MFMA pipe occupancy (SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD, GRBM_GUI_ACTIVE per XCD)
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$out/wpmc" -o pmc -- python "$root/scripts/r04/kernel_probe.py" > "$out/wpmc.log" 2>&1
# the LDS port of the two split-bf16 kernels (reads + stage writes against the matrix pipe's cycles); optional passes, a
# counter this rocprofv3 does not know only loses its own pass
( rocprofv3 -L 2>/dev/null | grep -i -E "^\s*(Name|counter)?.*(LDS|MFMA)" | head -60 ) > "$out/counters_lds_mfma.txt" 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d "$out/wlds" -o pmc -- python "$root/scripts/r04/kernel_probe.py" > "$out/wlds.log" 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d "$out/winst" -o pmc -- python "$root/scripts/r04/kernel_probe.py" > "$out/winst.log" 2>&1
if [ -z "${LITE:-}" ]; then
# BASELINE config 2 (bipartite 1M / 20M, d = 256): kernel time and HBM bytes of the same kernel
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/c2_stats" -o c2 -- python "$root/scripts/pmc_probe.py" --graph c2 --iters 40 > "$out/c2_stats.log" 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/c2_fetch" -o pmc -- python "$root/scripts/pmc_probe.py" --graph c2 > "$out/c2_fetch.log" 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/c2_write" -o pmc -- python "$root/scripts/pmc_probe.py" --graph c2 > "$out/c2_write.log" 2>&1
fi
timeout 600 python "$root/bench.py" > "$out/bench_plain.log" 2>&1
tail -1 "$out/bench_plain.log"
for f in fetch write ea hit wpmc wlds winst c2_stats c2_fetch c2_write; do tail -1 "$out/$f.log" | cut -c1-200; done
# keep the merge under the gpurun_out size limit: only the summaries that summarize_profile.py reads
find "$out" -type f ! -name "*_kernel_stats.csv" ! -name "pmc_counter_collection.csv" ! -name "*.log" -delete
find "$out" -type f -size +8M -exec sh -c 'grep cleora "$1" > "$1.tmp"; head -1 "$1" | cat - "$1.tmp" > "$1.f"; mv "$1.f" "$1"; rm "$1.tmp"' _ {} \;
du -sh "$out"
