"""Condenses a rocprofv3 --kernel-trace csv to (kernel, start, end, queue) rows of this library's kernels: who overlaps whom."""
import csv
import re
import sys

src, dst = sys.argv[1], sys.argv[2]
rows = [r for r in csv.DictReader(open(src)) if "cleora" in r["Kernel_Name"]]
t0 = min(int(r["Start_Timestamp"]) for r in rows)


def short(n):
    m = re.search(r"(\w+_kernel)", n)
    return m.group(1) if m else n[:40]


with open(dst, "w") as out:
    out.write("kernel,start_us,end_us,dur_us,queue,stream\n")
    for r in sorted(rows, key=lambda r: int(r["Start_Timestamp"])):
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        out.write(f"{short(r['Kernel_Name'])},{s / 1e3:.1f},{e / 1e3:.1f},{(e - s) / 1e3:.1f},{r.get('Queue_Id', '')},{r.get('Stream_Id', '')}\n")
print(len(rows), "kernels")
