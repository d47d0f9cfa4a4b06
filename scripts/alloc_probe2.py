#!/usr/bin/env python3
"""Developer probe: sensitivity of the SpMM time to the relative placement of X and Y."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cleora_amd import _hip, synth
dev = torch.device("cuda:0")
n_max, d, extra = 10_000_000, 256, 4_200_000
bufs = [torch.empty((n_max + extra, d), dtype=torch.float32, device=dev) for _ in range(2)]
g = synth.power_law_graph(10_000_000, 95_000_000, 2, dev)
n, nnz = g["n"], g["nnz"]
L = _hip.lib()
graph = _hip.Graph.from_device(n, n, nnz, g["rowptr"].data_ptr(), g["col"].data_ptr(), g["val_left"].data_ptr(), None, 0, 0, 0, keepalive=g)
s = torch.cuda.current_stream().cuda_stream
src = torch.randn((n, d), device=dev); src /= src.norm(dim=1, keepdim=True)
bufs[0][:n].copy_(src)
print("x base", hex(bufs[0].data_ptr()), "y base", hex(bufs[1].data_ptr()), "delta MiB", (bufs[1].data_ptr() - bufs[0].data_ptr()) / 2**20, flush=True)
def run(xp, yp, label):
    for _ in range(2):
        _hip.check(L.cleora_propagate_dev(graph.handle, 0, xp, d, d, yp, d, _hip.F_L2NORM, 0.0, None, None, None, s))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
        _hip.check(L.cleora_propagate_dev(graph.handle, 0, xp, d, d, yp, d, _hip.F_L2NORM, 0.0, None, None, None, s))
    e1.record(); torch.cuda.synchronize()
    print(f"{label:36s} {e0.elapsed_time(e1)/4:.2f} ms", flush=True)
for rows in (0, 1, 4, 16, 64, 256, 1024, 2048, 4096, 16384, 65536, 262144, 1048576, 2097152, 4194304):
    run(bufs[0].data_ptr(), bufs[1].data_ptr() + rows * d * 4, f"y offset {rows} rows ({rows*d*4/2**20:.3f} MiB)")
# also x offset inside its own allocation
for rows in (1024, 1048576):
    bufs[0][rows:rows + n].copy_(src)
    run(bufs[0].data_ptr() + rows * d * 4, bufs[1].data_ptr(), f"x offset {rows} rows")
