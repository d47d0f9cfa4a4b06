#!/usr/bin/env python3
"""Developer probe: the stand-alone row pass (rowops_kernel) at the widths the column partition uses."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cleora_amd import _hip
L = _hip.lib()
dev = torch.device("cuda:0")
n = 10_000_000
s = torch.cuda.current_stream().cuda_stream
def timed(fn, reps=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for d in (32, 64, 128, 256):
    x = torch.randn((n, d), device=dev); y = torch.empty_like(x)
    sumsq = x.pow(2).sum(1).float().contiguous()
    gb = 2 * n * d * 4 / 1e9
    for name, flags in (("L2NORM exact", _hip.F_L2NORM), ("L2NORM fast", _hip.F_L2NORM | _hip.F_FASTNORM), ("SCALE", _hip.F_SCALE)):
        for inplace in (False, True):
            out = x if inplace else y
            t = timed(lambda: _hip.check(L.cleora_rowops_dev(x.data_ptr(), d, n, d, out.data_ptr(), d, flags, 0.0, None, None, sumsq.data_ptr(), s)))
            print(f"d={d:4d} {name:13s} inplace={inplace}: {t:.3f} ms = {gb / t * 1e3:.0f} GB/s", flush=True)
    t = timed(lambda: y.copy_(x))
    print(f"d={d:4d} torch copy_: {t:.3f} ms = {gb / t * 1e3:.0f} GB/s", flush=True)
    del x, y
