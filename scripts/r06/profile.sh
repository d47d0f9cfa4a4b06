#!/bin/bash
# Round-6 profiling recipe (run on the GPU box through gpurun): kernel-trace stats of the bench command, the HBM-traffic passes
# (FETCH_SIZE / WRITE_SIZE, calibrated on kernels that move a known number of bytes: scripts/pmc_probe.py), and — VERDICT round 4
# next #6 — the address-translation and L2 counters of the SpMM at config 3 against config 4's size.
#   scripts/r06/profile.sh stats|traffic|whiten|tlb_c3|tlb_c4s
part=${1:-stats}
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/gpurun_out/prof_r06
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
TLB="TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE"
case $part in
stats)
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o bench -- python "$root/bench.py" --no-cpu-baseline --whiten-iters 0 --no-end-to-end > "$out/bench_stats.log" 2>&1
  tail -1 "$out/bench_stats.log" | cut -c1-400 ;;
traffic)
  timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/fetch" -o pmc -- python "$root/scripts/pmc_probe.py" > "$out/fetch.log" 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/write" -o pmc -- python "$root/scripts/pmc_probe.py" > "$out/write.log" 2>&1
  tail -1 "$out/fetch.log"; tail -1 "$out/write.log" ;;
whiten)
  # the whitening kernels at the C3 shape: kernel stats and the MFMA-pipe occupancy counters (condensed by summarize_profile.py r05)
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/wstats" -o whiten -- python "$root/scripts/r06/whiten_kernels_probe.py" > "$out/whiten_stats.log" 2>&1
  timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$out/wpmc" -o pmc -- python "$root/scripts/r06/whiten_kernels_probe.py" > "$out/wpmc.log" 2>&1
  tail -1 "$out/whiten_stats.log" | cut -c1-300; tail -1 "$out/wpmc.log" | cut -c1-200 ;;
tlb_c3)
  ( rocprofv3 -L 2>/dev/null | grep -i -E "UTCL|TLB" | head -40 ) > "$out/counters_tlb.txt" 2>&1
  timeout 400 rocprofv3 --pmc $TLB --output-format csv -d "$out/tlb_c3" -o pmc -- python "$root/scripts/pmc_probe.py" --iters 5 > "$out/tlb_c3.log" 2>&1
  tail -2 "$out/tlb_c3.log" | cut -c1-300 ;;
tlb_c4s)
  timeout 900 rocprofv3 --pmc $TLB --output-format csv -d "$out/tlb_c4s" -o pmc -- python "$root/scripts/pmc_probe.py" --nodes 111000000 --pairs 800000000 --iters 5 > "$out/tlb_c4s.log" 2>&1
  tail -2 "$out/tlb_c4s.log" | cut -c1-300 ;;
esac
# keep the merge under the gpurun_out size limit: only the summaries
find "$out" -type f ! -name "*_kernel_stats.csv" ! -name "pmc_counter_collection.csv" ! -name "*.log" ! -name "*.txt" -delete
find "$out" -type f -size +8M -exec sh -c 'grep cleora "$1" > "$1.tmp"; head -1 "$1" | cat - "$1.tmp" > "$1.f"; mv "$1.f" "$1"; rm "$1.tmp"' _ {} \;
du -sh "$out"
