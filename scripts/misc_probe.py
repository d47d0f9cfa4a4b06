#!/usr/bin/env python3
"""Developer probe: host vs device eigh, PCIe-inclusive propagate, C++ builder throughput."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cleora_amd import _hip, _host, synth
from cleora_amd.pycleora import SparseMatrix

for d in (256, 1024):
    a = np.random.default_rng(0).standard_normal((4 * d, d))
    cov = a.T @ a / (4 * d)
    t0 = time.perf_counter(); w, v = np.linalg.eigh(cov); t1 = time.perf_counter()
    ct = torch.from_numpy(cov).cuda()
    torch.linalg.eigh(ct); torch.cuda.synchronize()
    t2 = time.perf_counter(); wt, vt = torch.linalg.eigh(ct); torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f"eigh d={d}: numpy host {1e3*(t1-t0):.1f} ms, torch device {1e3*(t3-t2):.1f} ms, max|dw|={np.abs(w - wt.cpu().numpy()).max():.2e}", flush=True)

# C++ builder throughput: 2M reflexive 2-token lines
rng = np.random.default_rng(1)
a = rng.integers(0, 500_000, 2_000_000); b = rng.integers(0, 500_000, 2_000_000)
lines = [f"{x} {y}" for x, y in zip(a, b)]
t0 = time.perf_counter()
g = SparseMatrix.from_iterator(iter(lines), "complex::reflexive::node")
t1 = time.perf_counter()
print(f"builder: {len(lines)} lines -> n={g.num_entities} nnz={g.num_edges} in {t1-t0:.2f} s ({len(lines)/(t1-t0)/1e6:.2f} M lines/s)", flush=True)

# PCIe-inclusive propagate through the drop-in (numpy in, numpy out), d=256
x = g.initialize_deterministically(256)
g.left_markov_propagate(x)
t0 = time.perf_counter()
for _ in range(3):
    y = g.left_markov_propagate(x)
t1 = time.perf_counter()
nb = x.nbytes
print(f"drop-in propagate n={g.num_entities} d=256: {(t1-t0)/3*1e3:.1f} ms per call incl. H2D+D2H of {nb/1e6:.0f} MB each way "
      f"=> {2*nb/((t1-t0)/3)/1e9:.1f} GB/s effective PCIe", flush=True)
t0 = time.perf_counter(); e = g.embed_fast(256, 40); t1 = time.perf_counter()
print(f"embed_fast(256, 40) device-resident: {t1-t0:.3f} s total ({(t1-t0)/40*1e3:.2f} ms/iter incl. one upload/download)")
