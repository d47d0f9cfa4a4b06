#!/usr/bin/env python3
"""Developer probe: is the allocation-pair sensitivity a page-table fragment (TLB reach) effect?  Buffers from
hipExtMallocWithFlags(hipDeviceMallocContiguous) are physically contiguous (largest fragments); plain hipMalloc
buffers are whatever the VRAM manager had.  All ordered (x, y) pairs inside and across the two groups."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cleora_amd import _hip, synth
dev = torch.device("cuda:0")
g = synth.power_law_graph(10_000_000, 95_000_000, 2, dev)
n, nnz, d = g["n"], g["nnz"], 256
L = _hip.lib()
graph = _hip.Graph.from_device(n, n, nnz, g["rowptr"].data_ptr(), g["col"].data_ptr(), g["val_left"].data_ptr(), None, 0, 0, 0, keepalive=g)
graph.set_hot_cache(768 << 20)
s = torch.cuda.current_stream().cuda_stream
rt = None
for m in open("/proc/self/maps"):
    if "libamdhip64" in m:
        rt = ctypes.CDLL(m.split()[-1]); break
bytes_ = n * d * 4
def alloc(contig):
    p = ctypes.c_void_p()
    rc = rt.hipExtMallocWithFlags(ctypes.byref(p), ctypes.c_size_t(bytes_), ctypes.c_uint(0x4)) if contig else rt.hipMalloc(ctypes.byref(p), ctypes.c_size_t(bytes_))
    assert rc == 0, rc
    return p.value
bufs = {}
for k in range(3):
    bufs[f"m{k}"] = alloc(False)
    bufs[f"c{k}"] = alloc(True)
src = torch.randn((n, d), device=dev); src /= src.norm(dim=1, keepdim=True)
for name, p in bufs.items():
    rt.hipMemcpy(ctypes.c_void_p(p), ctypes.c_void_p(src.data_ptr()), ctypes.c_size_t(bytes_), ctypes.c_int(3))
def t(xp, yp):
    for _ in range(2):
        _hip.check(L.cleora_propagate_dev(graph.handle, 0, xp, d, d, yp, d, 0, 0.0, None, None, None, s))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(2):
        _hip.check(L.cleora_propagate_dev(graph.handle, 0, xp, d, d, yp, d, 0, 0.0, None, None, None, s))
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 2
names = list(bufs)
print("va:", {k: hex(v) for k, v in bufs.items()})
print("x \\ y   " + "  ".join(f"{b:>6s}" for b in names))
for a in names:
    row = []
    for b in names:
        if a == b:
            row.append("   -  ")
        else:
            xp, yp = bufs[a], bufs[b]
            rt.hipMemcpy(ctypes.c_void_p(xp), ctypes.c_void_p(src.data_ptr()), ctypes.c_size_t(bytes_), ctypes.c_int(3))
            row.append(f"{t(xp, yp):6.2f}")
    print(f"{a:7s} " + "  ".join(row), flush=True)
