#!/usr/bin/env python3
"""Registers, scratch, occupancy and LDS of every kernel, from hipcc -Rpass-analysis=kernel-resource-usage (runs in the
build container, no GPU): python scripts/kernel_resources.py r04 -> profiles/r04_kernel_resources.txt"""
import os, re, subprocess, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "cleora_amd", "csrc")
lines = ["# hipcc --offload-arch=gfx950 -O3 -Rpass-analysis=kernel-resource-usage over cleora_amd/csrc/*.hip "
         "(build flags of build.sh): file, kernel, VGPRs, AGPRs, scratch bytes/lane, waves/SIMD, LDS bytes/block"]
for f in ("spmm", "rowops", "whiten", "project_f16", "eigh", "hot", "attention", "similarity", "abi", "peer", "sharded", "colsharded", "multi", "stager"):
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math"]
    if f == "whiten":
        cmd += ["-mllvm", "-amdgpu-mfma-vgpr-form"]
    cmd += ["-c", f + ".hip", "-o", "/tmp/kernel_resources.o", "-Rpass-analysis=kernel-resource-usage"]
    log = subprocess.run(cmd, cwd=src, capture_output=True, text=True).stderr
    cur, rows = None, {}
    for line in log.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            rows[cur] = {}
            continue
        m = re.search(r"remark:\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)", line)
        if m and cur:
            rows[cur][m.group(1)] = m.group(2)
    for k, v in rows.items():
        name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"cleora::\(anonymous namespace\)::", "", name)
        name = re.sub(r"\(.*", "", name)
        lines.append(f"{f + '.hip':15s} {name:62s} VGPR {v.get('VGPRs', '?'):>3}  AGPR {v.get('AGPRs', '?'):>3}  scratch {v.get('ScratchSize [bytes/lane]', '?'):>4}  "
                     f"occ {v.get('Occupancy [waves/SIMD]', '?')}  LDS {v.get('LDS Size [bytes/block]', '?')}")
out = os.path.join(root, "profiles", f"{tag}_kernel_resources.txt")
open(out, "w").write("\n".join(lines) + "\n")
print(out, len(lines) - 1, "kernels")
