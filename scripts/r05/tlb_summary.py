"""Address-translation and L2 counters of the SpMM launch (rocprofv3 --pmc pass of scripts/r05/profile.sh tlb_*): per-launch averages."""
import collections
import csv
import json
import sys

out = {}
for tag in sys.argv[1:]:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f"gpurun_out/prof_r05/{tag}/pmc_counter_collection.csv")):
        for k in ("spmm_rows_kernel", "hub_inorder_kernel"):
            if k in r["Kernel_Name"]:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out[tag] = {k: {c: sum(v[2:]) / max(len(v[2:]), 1) for c, v in cs.items()} | {"launches_averaged": len(next(iter(cs.values()))) - 2} for k, cs in acc.items()}
    for k, cs in out[tag].items():
        if "TCP_UTCL1_TRANSLATION_MISS_sum" in cs:
            cs["utcl1_miss_rate"] = cs["TCP_UTCL1_TRANSLATION_MISS_sum"] / max(cs["TCP_UTCL1_TRANSLATION_MISS_sum"] + cs["TCP_UTCL1_TRANSLATION_HIT_sum"], 1)
        if "TCC_HIT_sum" in cs:
            cs["l2_hit_rate"] = cs["TCC_HIT_sum"] / max(cs["TCC_HIT_sum"] + cs["TCC_MISS_sum"], 1)
print(json.dumps(out, indent=1))
