#!/usr/bin/env python3
"""Developer probe: the default embed() loop (whiten=True) device-resident, per-iteration cost."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cleora_amd import _hip, synth, sharded

dev = torch.device("cuda:0")
for (nodes, pairs, d, iters) in ((1_000_000, 10_000_000, 256, 10), (10_000_000, 95_000_000, 256, 5), (2_000_000, 40_000_000, 1024, 4)):
    g = synth.power_law_graph(nodes, pairs, 5, dev)
    n = g["n"]
    sg = sharded.ShardedGraph(n, g["rowptr"], g["col"], g["val_left"], None, 0, 1, 1, sharded.HipBackend(dev))
    x0 = torch.zeros((sg.n_pad, d), dtype=torch.float32, device=dev)
    hashes = synth.entity_hashes(n, 0, dev)
    _hip.check(_hip.lib().cleora_init_dev(hashes.data_ptr(), n, d, 0, x0.data_ptr(), d, torch.cuda.current_stream().cuda_stream))
    sharded.embed_sharded(sg, 0, x0, 1, whiten=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    x, _ = sharded.embed_sharded(sg, 0, x0, iters, whiten=True)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    xs, _ = sharded.embed_sharded(sg, 0, x0, iters, whiten=False)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    cov = torch.cov(x[:n].T.double())
    print(f"n={n} nnz={g['nnz']} d={d}: whitened loop {1e3*(t1-t0)/iters:.1f} ms/iter, plain loop {1e3*(t2-t1)/iters:.1f} ms/iter; "
          f"|cov-I|max={float((cov - torch.eye(d, device=dev, dtype=torch.float64)).abs().max()):.2e}", flush=True)
    del g, sg, x0, x, xs
    torch.cuda.empty_cache()
