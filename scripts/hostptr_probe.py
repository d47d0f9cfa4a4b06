#!/usr/bin/env python3
"""Developer probe (round 2): the host-pointer drop-in call — SparseMatrix.left_markov_propagate(numpy) -> numpy, what
the reference's unmodified embed() does 40 times (src/lib.rs:29-47) — end to end, pageable memory both ways.
Effective GB/s = (bytes in + bytes out) / time.  Round 1: 18.7 GB/s (54.7 ms at 0.5M x 256)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cleora_amd import _hip, synth
L = _hip.lib()
dev = torch.device("cuda:0")
for nodes, pairs in ((500_000, 4_500_000), (10_000_000, 95_000_000)):
    g = synth.power_law_graph(nodes, pairs, 2, dev)
    n, nnz, d = g["n"], g["nnz"], 256
    graph = _hip.Graph.from_device(n, n, nnz, g["rowptr"].data_ptr(), g["col"].data_ptr(), g["val_left"].data_ptr(), None, 0, keepalive=g)
    x = np.random.default_rng(0).standard_normal((n, d), dtype=np.float32)
    y = np.empty_like(x)
    for rep in range(4):
        t0 = time.perf_counter()
        _hip.check(L.cleora_propagate(graph.handle, 0, _hip.ptr(x), d, _hip.ptr(y)))
        el = time.perf_counter() - t0
        print(f"n={n} d={d}: call {rep}: {el * 1e3:.1f} ms  = {2 * x.nbytes / el / 1e9:.1f} GB/s effective", flush=True)
    # the pieces
    buf = _hip.DevArray((n, d), np.float32)
    for name, fn in (("h2d", lambda: L.cleora_memcpy_h2d(buf.ptr, _hip.ptr(x), x.nbytes, None)),
                     ("d2h", lambda: L.cleora_memcpy_d2h(_hip.ptr(y), buf.ptr, x.nbytes, None))):
        fn(); t0 = time.perf_counter(); fn(); el = time.perf_counter() - t0
        print(f"   {name}: {el * 1e3:.1f} ms = {x.nbytes / el / 1e9:.1f} GB/s", flush=True)
    t = torch.from_numpy(x)
    tt = torch.empty((n, d), device=dev)
    tt.copy_(t); torch.cuda.synchronize(); t0 = time.perf_counter(); tt.copy_(t); torch.cuda.synchronize(); el = time.perf_counter() - t0
    print(f"   torch pageable h2d for comparison: {el * 1e3:.1f} ms = {x.nbytes / el / 1e9:.1f} GB/s", flush=True)
    del graph, g, buf, tt
