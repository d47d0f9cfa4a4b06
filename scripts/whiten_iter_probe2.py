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
def tp(label):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sg.propagate(0, x, y, gather=False)
    torch.cuda.synchronize(); print(f"   {label}: propagate {1e3*(time.perf_counter()-t0):.1f} ms", flush=True)
for it in range(4):
    tp(f"iter {it} first")
    sg.whiten(y, out); torch.cuda.synchronize()
    x, out = out, x
    ax = x[:n].abs()
    mn = float(torch.where(ax > 0, ax, torch.full_like(ax, 1e30)).min())
    print(f" after whiten {it}: max|x|={float(ax.max()):.3e} min nonzero={mn:.3e} "
          f"denormal frac={float(((ax>0)&(ax<1.17e-38)).sum())/ax.numel():.6f}", flush=True)
    del ax
    for r in range(3):
        tp(f"  repeat {r} on whitened x")
    time.sleep(0.2)
    tp("  after 200 ms idle")
