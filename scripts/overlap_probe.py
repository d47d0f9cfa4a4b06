#!/usr/bin/env python3
"""Developer probe: does the MFMA-bound Gram kernel overlap with the HBM-bound SpMM when they run on two
streams?  (Gram of the previous row block next to the SpMM of the next one is the pipelining candidate for
the whitened loop.)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cleora_amd import _hip, synth
L = _hip.lib()
dev = torch.device("cuda:0")
g = synth.power_law_graph(10_000_000, 95_000_000, 2, dev)
n, nnz, d = g["n"], g["nnz"], 256
graph = _hip.Graph.from_device(n, n, nnz, g["rowptr"].data_ptr(), g["col"].data_ptr(), g["val_left"].data_ptr(), None, 0, 0, 0, keepalive=g)
x = torch.randn((n, d), device=dev); x /= x.norm(dim=1, keepdim=True)
y = torch.empty_like(x)
z = torch.randn((n, d), device=dev)                       # the matrix whose Gram is taken (previous block's output)
mean = torch.zeros(d, dtype=torch.float64, device=dev)
gws = torch.empty(L.cleora_gram_workspace(n, d), dtype=torch.float64, device=dev)
gram = torch.empty((d, d), dtype=torch.float64, device=dev)
cws = torch.empty(L.cleora_colsum_workspace(n, d), dtype=torch.float64, device=dev)
cs = torch.empty(d, dtype=torch.float64, device=dev)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
def spmm(s): _hip.check(L.cleora_propagate_dev(graph.handle, 0, x.data_ptr(), d, d, y.data_ptr(), d, _hip.F_L2NORM, 0.0, None, None, None, s.cuda_stream))
def gramf(s):
    _hip.check(L.cleora_colsum_dev(z.data_ptr(), d, n, d, cws.data_ptr(), cs.data_ptr(), s.cuda_stream))
    _hip.check(L.cleora_centered_gram_dev(z.data_ptr(), d, n, d, mean.data_ptr(), gws.data_ptr(), gram.data_ptr(), s.cuda_stream))
def wall(fn, reps=5):
    fn(); fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
import time
def wall_host(fn, reps=5):
    fn(); fn(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / reps
t_s = wall_host(lambda: spmm(sa)); t_g = wall_host(lambda: gramf(sb))
def both():
    spmm(sa); gramf(sb)
t_b = wall_host(both)
print(f"SpMM alone {t_s:.2f} ms, colsum+Gram alone {t_g:.2f} ms, sum {t_s + t_g:.2f} ms, both streams concurrently {t_b:.2f} ms")
def both2():
    gramf(sb); spmm(sa)
print(f"launch order reversed: {wall_host(both2):.2f} ms")
