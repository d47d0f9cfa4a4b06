#!/usr/bin/env python3
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cleora_amd import _hip, synth, sharded
dev = torch.device("cuda:0")
g = synth.power_law_graph(10_000_000, 95_000_000, 5, dev)
n, d = g["n"], 256
sg = sharded.ShardedGraph(n, g["rowptr"], g["col"], g["val_left"], None, 0, 1, 1, sharded.HipBackend(dev))
x = torch.zeros((sg.n_pad, d), dtype=torch.float32, device=dev)
hashes = synth.entity_hashes(n, 0, dev)
_hip.check(_hip.lib().cleora_init_dev(hashes.data_ptr(), n, d, 0, x.data_ptr(), d, torch.cuda.current_stream().cuda_stream))
y = torch.zeros_like(x); out = torch.zeros_like(x)
import cProfile, pstats
for it in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sg.propagate(0, x, y, gather=False)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    if it == 4:
        pr = cProfile.Profile(); pr.enable()
    sg.whiten(y, out)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    if it == 4:
        pr.disable(); pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
    print(f"iter {it}: propagate {1e3*(t1-t0):.1f} ms, whiten {1e3*(t2-t1):.1f} ms", flush=True)
    x, out = out, x
