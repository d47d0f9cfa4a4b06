#!/usr/bin/env python3
"""Condense a gpurun_out/prof_<tag>/ rocprofv3 capture into the tracked files under profiles/.

  profiles/<tag>_kernel_stats.csv   `rocprofv3 --kernel-trace --stats` summary of `python bench.py`
                                    (our kernels + the top torch kernels of the graph generator)
  profiles/<tag>_pmc.json           FETCH_SIZE / WRITE_SIZE per kernel from the two --pmc passes over
                                    scripts/pmc_probe.py, with the calibration used
  profiles/hbm_traffic.json         bytes per launch of the dominant kernel (read by bench.py)

Counter handling follows MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE are collected in
separate passes (TCC slots), both are in KiB, and on gfx950 FETCH_SIZE reports half of the
bytes of a 16-byte-per-lane coalesced read.  The factor is not assumed: the probe runs
the stand-alone L2 pass (reads exactly n*d*4 B with the same dwordx4 accesses as the SpMM gathers) and
init_kernel (writes exactly n*d*4 B), and the corrections are the ratios measured on those.
"""
import collections
import csv
import json
import os
import re
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)


def short(name):
    m = re.search(r"(\w+_kernel(?:<[^>]*>)?)", name)
    return m.group(1) if m else name[:60]


# 1. kernel stats
rows = list(csv.DictReader(open(os.path.join(src, "stats", "bench_kernel_stats.csv"))))
with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for r in [r for i, r in enumerate(rows) if i < 10 or "cleora" in r["Name"]]:
        name = r["Name"] if "cleora" in r["Name"] else r["Name"][:90]
        w.writerow([name, r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                    r["MinNs"], r["MaxNs"], r["StdDev"]])

# 1b. whitening kernels (rocprofv3 --kernel-trace --stats of scripts/whiten_probe.py), if captured
wpath = os.path.join(src, "wstats", "whiten_kernel_stats.csv")
if os.path.exists(wpath):
    wrows = [r for r in csv.DictReader(open(wpath)) if "cleora" in r["Name"]]
    with open(os.path.join(dst, f"{tag}_whiten_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in wrows:
            w.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                        r["MinNs"], r["MaxNs"], r["StdDev"]])

# 2. PMC
probe = open(os.path.join(src, "fetch.log")).read()
m = re.search(r"PMC_PROBE n=(\d+) nnz=(\d+) d=(\d+)", probe)
n, nnz, d = (int(v) for v in m.groups())
counters = collections.defaultdict(lambda: collections.defaultdict(list))
for kind in ("fetch", "write"):
    for r in csv.DictReader(open(os.path.join(src, kind, "pmc_counter_collection.csv"))):
        if "cleora" in r["Kernel_Name"]:
            counters[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
known = n * d * 4
# the stand-alone L2 pass (l2_exact16_kernel, formerly rowops_kernel) reads exactly n*d*4 B; init_kernel writes exactly that
l2_key = [k for k in counters if k.startswith("l2_exact16_kernel") or k.startswith("rowops_kernel")][0]
init_key = [k for k in counters if k.startswith("init_kernel")][0]
fetch_corr = known / (counters[l2_key]["FETCH_SIZE"][0] * 1024)
write_corr = known / (counters[init_key]["WRITE_SIZE"][0] * 1024)
out = {"tag": tag, "n": n, "nnz": nnz, "d": d,
       "calibration": {"known_bytes": known, "read_kernel": l2_key, "write_kernel": init_key,
                       "rowops_FETCH_SIZE_KiB": counters[l2_key]["FETCH_SIZE"][0],
                       "init_WRITE_SIZE_KiB": counters[init_key]["WRITE_SIZE"][0],
                       "fetch_correction": fetch_corr, "write_correction": write_corr},
       "kernels": {}}
for k, c in counters.items():
    f = sum(c["FETCH_SIZE"]) / max(len(c["FETCH_SIZE"]), 1)
    w_ = sum(c["WRITE_SIZE"]) / max(len(c["WRITE_SIZE"]), 1)
    out["kernels"][k] = {"launches": len(c["FETCH_SIZE"]), "FETCH_SIZE_KiB_avg": f, "WRITE_SIZE_KiB_avg": w_,
                         "hbm_bytes_per_launch": f * 1024 * fetch_corr + w_ * 1024 * write_corr}

# 2b. optional comparison passes: the gather cache policy switched off (--hot 0), and L2 hit rates
def spmm_counters(sub):
    path = os.path.join(src, sub, "pmc_counter_collection.csv")
    acc = collections.defaultdict(list)
    if os.path.exists(path):
        for r in csv.DictReader(open(path)):
            if "spmm_rows_kernel" in r["Kernel_Name"] and (sub.endswith("nohot") or "true, true" in r["Kernel_Name"]):
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}

policy = {}
nohot = spmm_counters("fetch_nohot")
if nohot:
    policy["fetch_bytes_per_launch_policy_off"] = nohot["FETCH_SIZE"] * 1024 * fetch_corr
    policy["fetch_bytes_per_launch_policy_on"] = [v for k, v in out["kernels"].items() if k.startswith("spmm_rows_kernel<64, 1, 4, true, true")][0]["FETCH_SIZE_KiB_avg"] * 1024 * fetch_corr
for sub, key in (("hit", "policy_on"), ("hit_nohot", "policy_off")):
    c = spmm_counters(sub)
    if c:
        policy[f"l2_hit_rate_{key}"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
        policy[f"l2_requests_{key}"] = c["TCC_HIT_sum"] + c["TCC_MISS_sum"]
if policy:
    out["gather_cache_policy"] = policy
ea = spmm_counters("ea")
if ea:
    # FETCH_SIZE is derived from these; 64-B requests unless flagged 32B.  "_DRAM" = destined for the memory controller (as
    # opposed to a peer GPU or the host) — still counted on the L2 side of the Infinity Cache.
    rd, rd32, dram = ea.get("TCC_EA0_RDREQ_sum", 0.0), ea.get("TCC_EA0_RDREQ_32B_sum", 0.0), ea.get("TCC_EA0_RDREQ_DRAM_sum", 0.0)
    out["ea_read_requests"] = {"kernel": "spmm_rows_kernel<64, 1, 4, true, true, ...>", "TCC_EA0_RDREQ": rd, "TCC_EA0_RDREQ_32B": rd32,
                               "TCC_EA0_RDREQ_DRAM": dram, "bytes_at_64B": (rd - rd32) * 64 + rd32 * 32,
                               "bytes_at_64B_x_fetch_correction": ((rd - rd32) * 64 + rd32 * 32) * fetch_corr,
                               "dram_fraction_of_requests": dram / rd if rd else None}
json.dump(out, open(os.path.join(dst, f"{tag}_pmc.json"), "w"), indent=1)
dom = sorted((k for k in out["kernels"] if k.startswith("spmm_rows_kernel")), key=lambda k: out["kernels"][k]["launches"])[-1]
# stamp: bench.py only trusts this figure for the build of the kernel it was measured on (run this script right after
# the profile, before touching the kernel sources)
import hashlib
_h = hashlib.sha256()
for rel in ("cleora_amd/csrc/spmm.hip", "cleora_amd/csrc/row_epilogue.h", "cleora_amd/csrc/hot.hip", "cleora_amd/csrc/common.h"):   # = bench.py KERNEL_SOURCES
    _h.update(open(os.path.join(os.path.dirname(dst), rel), "rb").read())
# one SpMM = the main launch + the in-order hub launch beside it + the hub rows' epilogue (spmm.hip): their bytes together
hub = {k: v["hbm_bytes_per_launch"] for k, v in out["kernels"].items() if k.startswith("hub_inorder_kernel") or k.startswith("hub_epilogue_kernel") or k.startswith("hub_chain_kernel")}
json.dump({"n": n, "nnz": nnz, "d": d, "kernel": dom, "source": f"{tag}_pmc.json",
           "kernel_source_sha16": _h.hexdigest()[:16],
           "main_kernel_bytes_per_launch": out["kernels"][dom]["hbm_bytes_per_launch"], "hub_kernels_bytes_per_launch": hub,
           "bytes_per_launch": out["kernels"][dom]["hbm_bytes_per_launch"] + sum(hub.values())},
          open(os.path.join(dst, "hbm_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))


# 3. whitening kernels: MFMA pipe occupancy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 XCDs * 1024 SIMDs)
wp = os.path.join(src, "wpmc", "pmc_counter_collection.csv")
if os.path.exists(wp):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(wp)):
        if "cleora" in r["Kernel_Name"] and any(k in r["Kernel_Name"] for k in ("gram_kernel", "gram32_kernel", "gram16_kernel", "project_kernel",
                                                                                "project_rows_kernel", "project_split_kernel", "project_f16_kernel")):
            acc[(short(r["Kernel_Name"]), r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    wout = {"formula": "mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024)  [8 XCDs, 1024 SIMDs]", "kernels": []}
    for (name, grid), c in sorted(acc.items()):
        busy = sum(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(c["SQ_VALU_MFMA_BUSY_CYCLES"])
        act = sum(c["GRBM_GUI_ACTIVE"]) / len(c["GRBM_GUI_ACTIVE"])
        rec = {"kernel": name, "grid_size": int(grid), "launches": len(c["GRBM_GUI_ACTIVE"]),
               "SQ_VALU_MFMA_BUSY_CYCLES": busy, "GRBM_GUI_ACTIVE": act, "mfma_busy": busy / (act / 8 * 1024)}
        for extra in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_BUSY_CYCLES"):
            if extra in c:
                rec[extra] = sum(c[extra]) / len(c[extra])
        if "SQ_WAVE_CYCLES" in rec:
            rec["wait_inst_share_of_wave_cycles"] = rec.get("SQ_WAIT_INST_ANY", 0.0) / rec["SQ_WAVE_CYCLES"]
            rec["waves_per_simd"] = rec["SQ_WAVE_CYCLES"] * 4 / (act / 8 * 1024)
        wout["kernels"].append(rec)
    json.dump(wout, open(os.path.join(dst, f"{tag}_whiten_pmc.json"), "w"), indent=1)

# 4. BASELINE config 2 on one GPU: kernel time (rocprofv3 stats) and HBM bytes (PMC) of the same SpMM kernel
c2s = os.path.join(src, "c2_stats", "c2_kernel_stats.csv")
if os.path.exists(c2s):
    m2 = re.search(r"PMC_PROBE n=(\d+) nnz=(\d+) d=(\d+)", open(os.path.join(src, "c2_stats.log")).read())
    n2, nnz2, d2 = (int(v) for v in m2.groups())
    st = [r for r in csv.DictReader(open(c2s)) if "spmm_rows_kernel" in r["Name"]][0]
    byt = {}
    for kind, counter in (("c2_fetch", "FETCH_SIZE"), ("c2_write", "WRITE_SIZE")):
        vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(os.path.join(src, kind, "pmc_counter_collection.csv")))
                if "spmm_rows_kernel" in r["Kernel_Name"] and r["Counter_Name"] == counter]
        byt[counter] = sum(vals) / len(vals)
    alg = nnz2 * 8 + (n2 + 1) * 8 + nnz2 * d2 * 4 + n2 * d2 * 4
    pmc_bytes = byt["FETCH_SIZE"] * 1024 * fetch_corr + byt["WRITE_SIZE"] * 1024 * write_corr
    avg_ms = float(st["AverageNs"]) / 1e6
    json.dump({"config": "BASELINE config 2: bipartite 500k x 500k, 10M pairs, d=256, left Markov, SpMM + fused L2",
               "n": n2, "nnz": nnz2, "d": d2, "kernel": short(st["Name"]), "launches": int(st["Calls"]), "avg_launch_ms": avg_ms,
               "algorithmic_bytes_per_launch": alg, "algorithmic_GBps": alg / avg_ms / 1e6,
               "pmc_bytes_per_launch": pmc_bytes, "pmc_GBps": pmc_bytes / avg_ms / 1e6,
               "note": "X is 1.0 GB: part of it stays in L2 / Infinity Cache, so the fabric-side PMC bytes are below the gather model; "
                       "the PMC counters include Infinity-Cache hits"},
              open(os.path.join(dst, f"{tag}_c2.json"), "w"), indent=1)
