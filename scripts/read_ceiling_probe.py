#!/usr/bin/env python3
"""Developer probe: pure-READ streaming rate of this box (the guide's 6.29 TB/s ceiling is a COPY, half writes).
cosine_kernel reads X exactly once (n*d*4 B) with the same one-wavefront-per-1-KiB-row access as the SpMM's
gathers, but in address order; torch.sum is an independent read-only reduction; copy_ is the guide's case."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cleora_amd import _hip
L = _hip.lib()
dev = torch.device("cuda:0")
n, d = 10_000_000, 256
x = torch.randn((n, d), device=dev)
y = torch.empty_like(x)
q = torch.randn(d, device=dev)
sc = torch.empty(n, device=dev)
s = torch.cuda.current_stream().cuda_stream
def timed(fn, reps=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
gb = n * d * 4 / 1e9
t = timed(lambda: _hip.check(L.cleora_cosine_scores_dev(x.data_ptr(), d, n, d, q.data_ptr(), sc.data_ptr(), s)))
print(f"cosine_kernel (read X once, 1 KiB rows in order): {t:.3f} ms = {gb / t * 1e3:.0f} GB/s")
t = timed(lambda: x.sum())
print(f"torch.sum (read only): {t:.3f} ms = {gb / t * 1e3:.0f} GB/s")
t = timed(lambda: y.copy_(x))
print(f"torch copy_ (read + write): {t:.3f} ms = {2 * gb / t * 1e3:.0f} GB/s")
t = timed(lambda: _hip.check(L.cleora_rowops_dev(x.data_ptr(), d, n, d, y.data_ptr(), d, _hip.F_L2NORM, 0.0, None, None, None, s)))
print(f"rowops L2 (read + write): {t:.3f} ms = {2 * gb / t * 1e3:.0f} GB/s")
# random-order 1 KiB row reads with NO reuse: a permutation gather (what the SpMM's cold gathers look like)
perm = torch.randperm(n, device=dev)
t = timed(lambda: torch.index_select(x, 0, perm[: n // 4]), 5)
print(f"torch index_select of n/4 random rows (read + write): {t:.3f} ms = {2 * gb / 4 / t * 1e3:.0f} GB/s")
