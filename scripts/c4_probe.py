#!/usr/bin/env python3
"""Developer probe: BASELINE config 4 scale on ONE GPU — |V| = 111M, nnz = 3.3e9, d = 256: two iterates of
113.7 GB + 27 GB of CSR = 254 GB of the 288 GB.  Synthetic stand-in (uniform degree 30, random columns);
the point is capacity + 64-bit offsets + the gather rate at 11x the C3 footprint."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cleora_amd import _hip
dev = torch.device("cuda:0")
n, deg, d = int(sys.argv[1]) if len(sys.argv) > 1 else 111_000_000, 30, 256
nnz = n * deg
gen = torch.Generator(device=dev); gen.manual_seed(4)
col = torch.randint(0, n, (nnz,), generator=gen, device=dev, dtype=torch.int32)
val = torch.full((nnz,), 1.0 / deg, dtype=torch.float32, device=dev)
rowptr = torch.arange(0, nnz + 1, deg, dtype=torch.int64, device=dev)
g = _hip.Graph.from_device(n, n, nnz, rowptr.data_ptr(), col.data_ptr(), val.data_ptr(), None, 0, keepalive=(rowptr, col, val))
x = torch.empty((n, d), dtype=torch.float32, device=dev)
y = torch.empty((n, d), dtype=torch.float32, device=dev)
L = _hip.lib(); s = torch.cuda.current_stream().cuda_stream
hashes = torch.arange(n, device=dev, dtype=torch.int64) * 7919
_hip.check(L.cleora_init_dev(hashes.data_ptr(), n, d, 0, x.data_ptr(), d, s))
for _ in range(1):
    _hip.check(L.cleora_propagate_dev(g.handle, 0, x.data_ptr(), d, d, y.data_ptr(), d, _hip.F_L2NORM, 0.0, None, None, None, s)); x, y = y, x
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
iters = 4
for _ in range(iters):
    _hip.check(L.cleora_propagate_dev(g.handle, 0, x.data_ptr(), d, d, y.data_ptr(), d, _hip.F_L2NORM, 0.0, None, None, None, s)); x, y = y, x
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
# correctness of the last launch on sampled rows (x holds the result, y the previous iterate)
import time
bad = 0
for r in (0, 5, n // 3, n // 2, n - 2, n - 1):
    b0 = r * deg
    acc = (y[col[b0:b0 + deg].long()].double() * (1.0 / deg)).sum(0)
    want = acc / acc.norm().clamp(min=1e-10)
    bad = max(bad, float((x[r].double() - want).abs().max()))
t0 = time.perf_counter()
_hip.check(L.cleora_propagate_dev(g.handle, 0, x.data_ptr(), d, d, y.data_ptr(), d, _hip.F_L2NORM, 0.0, None, None, None, s))
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) * 1e3
print("max sampled row error", bad, "single launch wall ms", round(wall, 1), "unique cols in first 1M edges", int(torch.unique(col[:1_000_000]).numel()))
b = nnz * 8 + (n + 1) * 8 + nnz * d * 4 + n * d * 4
print(json.dumps({"n": n, "nnz": nnz, "d": d, "ms_per_iter": round(ms, 1), "it_per_s": round(1e3 / ms, 3), "gather_model_GBps": round(b / ms / 1e6),
                  "hbm_in_use_GiB": round(torch.cuda.memory_allocated() / 2**30, 1), "finite": bool(torch.isfinite(x[:1000]).all())}))
