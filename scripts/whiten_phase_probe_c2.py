#!/usr/bin/env python3
"""Developer probe: phase-by-phase timing of one whitened iteration at the C3 shape."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cleora_amd import _hip, synth, sharded
from cleora_amd.embed import eigh_descending
dev = torch.device("cuda:0")
g = synth.bipartite_graph(500_000, 500_000, 10_000_000, 1, dev)
n, d = g["n"], 256
be = sharded.HipBackend(dev)
sg = sharded.ShardedGraph(n, g["rowptr"], g["col"], g["val_left"], None, 0, 1, 1, be)
x = torch.zeros((sg.n_pad, d), dtype=torch.float32, device=dev)
hashes = synth.entity_hashes(n, 0, dev)
_hip.check(_hip.lib().cleora_init_dev(hashes.data_ptr(), n, d, 0, x.data_ptr(), d, torch.cuda.current_stream().cuda_stream))
y = torch.zeros_like(x); out = torch.zeros_like(x)
def T(label, fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    print(f"  {label:28s} {1e3*(time.perf_counter()-t0):8.2f} ms", flush=True); return r
for it in range(4):
    print("iteration", it)
    T("propagate+L2", lambda: sg.propagate(0, x, y, gather=False))
    yv = y[:n]
    cs = T("colsum", lambda: be.colsum(yv))
    mean = cs / float(n)
    gram = T("gram", lambda: be.gram(yv, mean))
    cov = T("gram -> host", lambda: gram.cpu().numpy() * (1.0 / (n - 1)))
    w, v = T("eigh host", lambda: eigh_descending(cov, "host"))
    def mk():
        scale = 1.0 / np.sqrt(np.maximum(w, 1e-10))
        return torch.from_numpy(np.ascontiguousarray((v * scale).astype(np.float32))).to(dev)
    tr = T("transform -> device", mk)
    mean32 = mean.to(torch.float32)
    T("project", lambda: be.project(yv, mean32, tr, out[:n]))
    T("whole sg.whiten()", lambda: sg.whiten(y, out))
    x, out = out, x
