#!/usr/bin/env python3
"""Developer probe: what the exact-order (sequential) sum of squares costs inside the fused SpMM at C3."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cleora_amd import _hip, synth
L = _hip.lib()
dev = torch.device("cuda:0")
g = synth.power_law_graph(10_000_000, 95_000_000, 2, dev)
n, nnz, d = g["n"], g["nnz"], 256
graph = _hip.Graph.from_device(n, n, nnz, g["rowptr"].data_ptr(), g["col"].data_ptr(), g["val_left"].data_ptr(), None, 0, 0, 0, keepalive=g)
x = torch.randn((n, d), device=dev); x /= x.norm(dim=1, keepdim=True)
ys = [torch.empty_like(x) for _ in range(3)]
s = torch.cuda.current_stream().cuda_stream
def run(flags, y, reps=6):
    for _ in range(3):
        _hip.check(L.cleora_propagate_dev(graph.handle, 0, x.data_ptr(), d, d, y.data_ptr(), d, flags, 0.0, None, None, None, s))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        _hip.check(L.cleora_propagate_dev(graph.handle, 0, x.data_ptr(), d, d, y.data_ptr(), d, flags, 0.0, None, None, None, s))
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for y in ys:
    print({name: round(run(f, y), 3) for name, f in (("no epilogue", 0), ("L2 exact", _hip.F_L2NORM), ("L2 fast", _hip.F_L2NORM | _hip.F_FASTNORM),
                                                     ("no epilogue again", 0), ("L2 exact again", _hip.F_L2NORM))}, flush=True)
