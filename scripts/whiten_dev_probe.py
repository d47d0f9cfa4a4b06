#!/usr/bin/env python3
"""Developer probe: cleora_whiten_dev (the device-only whitening chain) at the BASELINE shapes, and the
rocSOLVER dsyevd share of it (cleora_whiten_transform_dev alone)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cleora_amd import _hip
dev = torch.device("cuda:0")
L = _hip.lib()
s = torch.cuda.current_stream().cuda_stream
def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for n, d in ((1_000_000, 256), (9_999_997, 256), (2_000_000, 1024), (34, 128)):
    x = torch.randn((n, d), device=dev) * torch.linspace(0.5, 2.0, d, device=dev)
    y = torch.empty_like(x)
    ws = torch.empty(L.cleora_whiten_workspace(n, d), dtype=torch.uint8, device=dev)
    t_all = timed(lambda: _hip.check(L.cleora_whiten_dev(x.data_ptr(), d, n, d, 0, y.data_ptr(), d, ws.data_ptr(), None, s)), 5)
    gram = (torch.randn((d, d), device=dev, dtype=torch.float64))
    gram = gram @ gram.T
    t = torch.empty((d, d), dtype=torch.float32, device=dev)
    ews = torch.empty(L.cleora_eigh_workspace(d), dtype=torch.uint8, device=dev)
    t_eig = timed(lambda: _hip.check(L.cleora_whiten_transform_dev(gram.data_ptr(), n, d, d, t.data_ptr(), None, ews.data_ptr(), s)), 5)
    t_torch = timed(lambda: torch.linalg.eigh(gram), 5)
    print(f"n={n} d={d}: cleora_whiten_dev {t_all:.2f} ms (transform incl. dsyevd {t_eig:.2f} ms; torch.linalg.eigh {t_torch:.2f} ms)", flush=True)
    del x, y, ws
