#!/usr/bin/env python3
"""Developer probe: time colsum / centred Gram / projection at scale (GPU box)."""
import sys, os, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cleora_amd import _hip
L = _hip.lib()
for n, d in ((10_000_000, 256), (2_000_000, 1024)):
    x = torch.randn((n, d), device="cuda", dtype=torch.float32)
    out = torch.empty_like(x)
    s = torch.cuda.current_stream().cuda_stream
    cws = torch.empty(L.cleora_colsum_workspace(n, d), dtype=torch.float64, device="cuda")
    cs = torch.empty(d, dtype=torch.float64, device="cuda")
    gws = torch.empty(L.cleora_gram_workspace(n, d), dtype=torch.float64, device="cuda")
    gram = torch.empty((d, d), dtype=torch.float64, device="cuda")
    mean32 = torch.zeros(d, dtype=torch.float32, device="cuda")
    t = torch.randn((d, d), device="cuda", dtype=torch.float32)
    def timeit(fn, reps=3):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    ms = timeit(lambda: _hip.check(L.cleora_colsum_dev(x.data_ptr(), d, n, d, cws.data_ptr(), cs.data_ptr(), s)))
    print(f"n={n} d={d} colsum {ms:.2f} ms  {n*d*4/ms/1e6:.0f} GB/s")
    mean = (cs / n).contiguous()
    ms = timeit(lambda: _hip.check(L.cleora_centered_gram_dev(x.data_ptr(), d, n, d, mean.data_ptr(), gws.data_ptr(), gram.data_ptr(), s)))
    print(f"n={n} d={d} gram   {ms:.2f} ms  {2*n*d*d/ms/1e9:.1f} TFLOP/s(full) {n*d*(d+128)/ms/1e9:.1f} TFLOP/s(upper-tri tiles)")
    ref = (x[:100000].double() - mean).T @ (x[:100000].double() - mean)
    ms = timeit(lambda: _hip.check(L.cleora_project_dev(x.data_ptr(), d, n, d, mean32.data_ptr(), t.data_ptr(), d, out.data_ptr(), d, s)))
    print(f"n={n} d={d} project {ms:.2f} ms  {2*n*d*d/ms/1e9:.1f} TFLOP/s")
    chk = (out[:1000] - x[:1000] @ t).abs().max().item()
    print("project check", chk)
    del x, out, gws
