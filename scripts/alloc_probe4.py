#!/usr/bin/env python3
"""Developer probe (run under rocprofv3 --pmc): one SpMM launch per ordered (x, y) allocation pair."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cleora_amd import _hip, synth
dev = torch.device("cuda:0")
n_max, d = 10_000_000, 256
bufs = [torch.empty((n_max, d), dtype=torch.float32, device=dev) for _ in range(2)]
g = synth.power_law_graph(10_000_000, 95_000_000, 2, dev)
n, nnz = g["n"], g["nnz"]
L = _hip.lib()
graph = _hip.Graph.from_device(n, n, nnz, g["rowptr"].data_ptr(), g["col"].data_ptr(), g["val_left"].data_ptr(), None, 0, 0, 0, keepalive=g)
bufs += [torch.empty((n_max, d), dtype=torch.float32, device=dev) for _ in range(2)]
s = torch.cuda.current_stream().cuda_stream
src = torch.randn((n, d), device=dev); src /= src.norm(dim=1, keepdim=True)
for b in bufs: b[:n].copy_(src)
torch.cuda.synchronize()
k = 0
for i, xb in enumerate(bufs):
    for j, yb in enumerate(bufs):
        if i == j: continue
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _hip.check(L.cleora_propagate_dev(graph.handle, 0, xb.data_ptr(), d, d, yb.data_ptr(), d, _hip.F_L2NORM, 0.0, None, None, None, s))
        e1.record(); torch.cuda.synchronize()
        print(f"PAIR {k} x={i} y={j} ms={e0.elapsed_time(e1):.2f}", flush=True)
        k += 1
        yb[:n].copy_(src); torch.cuda.synchronize()
