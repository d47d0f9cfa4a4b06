#!/usr/bin/env python3
"""Developer probe: gather cache policy at C2 scale (X = 1 GB, a quarter of it fits the Infinity Cache)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cleora_amd import _hip, synth
L = _hip.lib()
dev = torch.device("cuda:0")
s = torch.cuda.current_stream().cuda_stream
def timed(fn, reps=20):
    for _ in range(4): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for name, g in (("C2 bipartite 1M/20M", synth.bipartite_graph(500_000, 500_000, 10_000_000, 1, dev)),
                ("power-law 2M/40M", synth.power_law_graph(2_000_000, 19_000_000, 2, dev)),
                ("power-law 4M/80M", synth.power_law_graph(4_000_000, 38_000_000, 2, dev))):
    n, nnz, d = g["n"], g["nnz"], 256
    graph = _hip.Graph.from_device(n, n, nnz, g["rowptr"].data_ptr(), g["col"].data_ptr(), g["val_left"].data_ptr(), None, 0, 0, 0, keepalive=g)
    x = torch.randn((n, d), device=dev); x /= x.norm(dim=1, keepdim=True)
    y = torch.empty_like(x)
    out = {}
    for hot in (0, 32 << 20, 64 << 20, 128 << 20, 256 << 20, 512 << 20, 0):
        graph.set_hot_cache(hot)
        ms = timed(lambda: _hip.check(L.cleora_propagate_dev(graph.handle, 0, x.data_ptr(), d, d, y.data_ptr(), d, _hip.F_L2NORM, 0.0, None, None, None, s)))
        out[f"{hot >> 20}MiB" + ("'" if f"{hot >> 20}MiB" in out else "")] = round(ms, 3)
    print(name, f"X={n * d * 4 / 2**30:.2f} GiB", out, flush=True)
    del graph, x, y, g
