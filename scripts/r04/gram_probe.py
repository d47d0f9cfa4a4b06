#!/usr/bin/env python3
"""Developer probe: milliseconds of the intermediate statistics through an alternative build of the library (profiling builds of
gram16_raw_kernel: -DG16_PROBE bit 0 = no MFMAs, bit 1 = no staging, bit 2 = no LDS-DMA).  gram_probe.py LIB [n d]"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cleora_amd import _hip
if sys.argv[1] != "-":
    _hip.LIB_PATH = os.path.abspath(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
d = int(sys.argv[3]) if len(sys.argv) > 3 else 256
dev = torch.device("cuda:0")
L = _hip.lib()
s = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device=dev); g.manual_seed(1)
x = torch.randn((n, d), device=dev, generator=g) * torch.linspace(0.3, 2.0, d, device=dev) + 0.05
x = (x / x.norm(dim=1, keepdim=True)).contiguous()
ws = torch.empty(L.cleora_whiten_workspace(n, d), dtype=torch.uint8, device=dev)
m64 = torch.empty(d, dtype=torch.float64, device=dev)
gi = torch.empty((d, d), dtype=torch.float64, device=dev)
def run():
    _hip.check(L.cleora_whiten_stats_dev(x.data_ptr(), d, n, d, ws.data_ptr(), 1, m64.data_ptr(), gi.data_ptr(), s))
run(); torch.cuda.synchronize()
best = 1e9
for _ in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        run()
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 5)
print(json.dumps({"lib": os.path.basename(_hip.LIB_PATH), "n": n, "d": d, "stats_intermediate_ms": round(best, 4)}), flush=True)
