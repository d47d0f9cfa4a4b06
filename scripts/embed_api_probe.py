#!/usr/bin/env python3
"""Developer probe: per-iteration cost of the host-pointer cleora_embed at C3 (slope of total time)."""
import ctypes, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cleora_amd import _hip, synth
dev = torch.device("cuda:0")
g = synth.power_law_graph(10_000_000, 95_000_000, 2, dev)
n, nnz, d = g["n"], g["nnz"], 256
L = _hip.lib()
graph = _hip.Graph.from_device(n, n, nnz, g["rowptr"].data_ptr(), g["col"].data_ptr(), g["val_left"].data_ptr(), None, 0, 0, 0, keepalive=g)
hashes = synth.entity_hashes(n, 0, dev).cpu().numpy().view(np.uint64)
out = np.empty((n, d), np.float32)
it = ctypes.c_uint64(0)
for rep in range(2):
    ts = {}
    for iters in (20, 60):
        t0 = time.perf_counter()
        _hip.check(L.cleora_embed(graph.handle, _hip.ptr(hashes), None, 0, d, iters, 0, 0.0, 0.0, 0, _hip.ptr(out), ctypes.byref(it)))
        ts[iters] = time.perf_counter() - t0
    print(f"rep {rep}: 20 it {ts[20]:.3f} s, 60 it {ts[60]:.3f} s -> {(ts[60]-ts[20])/40*1e3:.2f} ms/iter steady state; norms ok: {abs(float(np.linalg.norm(out[12345]))-1)<1e-5}", flush=True)
