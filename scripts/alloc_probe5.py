#!/usr/bin/env python3
"""Developer probe: how often do later allocations (with spacer allocations in between, or with the
contiguous flag) fall into a different placement class than the first iterate buffer?"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cleora_amd import _hip, synth
dev = torch.device("cuda:0")
g = synth.power_law_graph(10_000_000, 95_000_000, 2, dev)
n, nnz, d = g["n"], g["nnz"], 256
L = _hip.lib()
graph = _hip.Graph.from_device(n, n, nnz, g["rowptr"].data_ptr(), g["col"].data_ptr(), g["val_left"].data_ptr(), None, 0, 0, 0, keepalive=g)
s = torch.cuda.current_stream().cuda_stream
x0 = torch.randn((n, d), device=dev); x0 /= x0.norm(dim=1, keepdim=True)
def t(xp, yp):
    _hip.check(L.cleora_propagate_dev(graph.handle, 0, xp, d, d, yp, d, _hip.F_L2NORM, 0.0, None, None, None, s))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _hip.check(L.cleora_propagate_dev(graph.handle, 0, xp, d, d, yp, d, _hip.F_L2NORM, 0.0, None, None, None, s))
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)
keep = []
for k in range(8):
    spacer = torch.empty(int((0.6 + 0.83 * k) * 2**30), dtype=torch.uint8, device=dev) if k else None
    c = torch.empty((n, d), dtype=torch.float32, device=dev)
    keep += [spacer, c]
    print(f"cand {k} (torch, spacer {0 if spacer is None else spacer.numel()>>20} MiB) va={hex(c.data_ptr())}: x0->c {t(x0.data_ptr(), c.data_ptr()):.2f} ms", flush=True)
rt = None
for m in open("/proc/self/maps"):
    if "libamdhip64" in m:
        rt = ctypes.CDLL(m.split()[-1]); break
for k in range(3):
    p = ctypes.c_void_p()
    rc = rt.hipExtMallocWithFlags(ctypes.byref(p), ctypes.c_size_t(n * d * 4), ctypes.c_uint(0x4))
    print(f"contig {k} rc={rc} va={hex(p.value or 0)}: x0->c {t(x0.data_ptr(), p.value):.2f} ms", flush=True)
for k in range(3):
    p = ctypes.c_void_p()
    rc = rt.hipMalloc(ctypes.byref(p), ctypes.c_size_t(n * d * 4 + (k + 1) * (3 << 20)))
    print(f"hipMalloc {k} rc={rc} va={hex(p.value or 0)}: x0->c {t(x0.data_ptr(), p.value):.2f} ms", flush=True)
