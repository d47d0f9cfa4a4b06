#!/usr/bin/env python3
"""Developer probe: full (x buffer, y buffer) matrix of SpMM times over several allocations."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cleora_amd import _hip, synth
dev = torch.device("cuda:0")
n_max, d = 10_000_000, 256
names, bufs = [], []
for i in range(2):
    names.append(f"e{i}"); bufs.append(torch.empty((n_max, d), dtype=torch.float32, device=dev))
g = synth.power_law_graph(10_000_000, 95_000_000, 2, dev)
n, nnz = g["n"], g["nnz"]
L = _hip.lib()
graph = _hip.Graph.from_device(n, n, nnz, g["rowptr"].data_ptr(), g["col"].data_ptr(), g["val_left"].data_ptr(), None, 0, 0, 0, keepalive=g)
for i in range(3):
    names.append(f"l{i}"); bufs.append(torch.empty((n_max, d), dtype=torch.float32, device=dev))
s = torch.cuda.current_stream().cuda_stream
src = torch.randn((n, d), device=dev); src /= src.norm(dim=1, keepdim=True)
for nm, b in zip(names, bufs):
    b[:n].copy_(src)
    print(nm, hex(b.data_ptr()), flush=True)
print("col", hex(g["col"].data_ptr()), "val", hex(g["val_left"].data_ptr()), "rowptr", hex(g["rowptr"].data_ptr()))
def run(xp, yp):
    _hip.check(L.cleora_propagate_dev(graph.handle, 0, xp, d, d, yp, d, _hip.F_L2NORM, 0.0, None, None, None, s))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        _hip.check(L.cleora_propagate_dev(graph.handle, 0, xp, d, d, yp, d, _hip.F_L2NORM, 0.0, None, None, None, s))
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 3
print("rows: x, cols: y   ", "  ".join(f"{nm:>6s}" for nm in names))
for i, xb in enumerate(bufs):
    row = []
    for j, yb in enumerate(bufs):
        if i == j:
            row.append("   -  "); continue
        t = run(xb.data_ptr(), yb.data_ptr())
        xb[:n].copy_(src)  # y was overwritten when it is later used as x
        yb[:n].copy_(src)
        row.append(f"{t:6.2f}")
    print(f"x={names[i]:3s}             ", "  ".join(row), flush=True)
