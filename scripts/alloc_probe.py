#!/usr/bin/env python3
"""Developer probe: does the physical placement of X (allocation order / contiguous flag) change the
gather rate?  Same kernel, same data, different allocations of the gathered matrix."""
import ctypes, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cleora_amd import _hip, synth, sharded
dev = torch.device("cuda:0")
n_max, d = 10_000_000, 256
early = [torch.empty((n_max, d), dtype=torch.float32, device=dev) for _ in range(2)]   # before any churn
g = synth.power_law_graph(10_000_000, 95_000_000, 2, dev)
n, nnz = g["n"], g["nnz"]
L = _hip.lib()
graph = _hip.Graph.from_device(n, n, nnz, g["rowptr"].data_ptr(), g["col"].data_ptr(), g["val_left"].data_ptr(), None, 0, 0, 0, keepalive=g)
late = [torch.empty((n_max, d), dtype=torch.float32, device=dev) for _ in range(3)]
hip = ctypes.CDLL("libamdhip64.so.7") if False else None
import torch.cuda
rt = None
for m in open("/proc/self/maps"):
    if "libamdhip64" in m:
        rt = ctypes.CDLL(m.split()[-1]); break
contig = []
for _ in range(2):
    p = ctypes.c_void_p()
    rc = rt.hipExtMallocWithFlags(ctypes.byref(p), ctypes.c_size_t(n_max * d * 4), ctypes.c_uint(0x4))
    print("hipExtMallocWithFlags(contiguous) rc =", rc, hex(p.value or 0), flush=True)
    contig.append(p.value if rc == 0 else None)
s = torch.cuda.current_stream().cuda_stream
src = torch.randn((n, d), device=dev); src /= src.norm(dim=1, keepdim=True)
def run(xp, yp, label):
    for _ in range(2):
        _hip.check(L.cleora_propagate_dev(graph.handle, 0, xp, d, d, yp, d, _hip.F_L2NORM, 0.0, None, None, None, s))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        _hip.check(L.cleora_propagate_dev(graph.handle, 0, xp, d, d, yp, d, _hip.F_L2NORM, 0.0, None, None, None, s))
    e1.record(); torch.cuda.synchronize()
    print(f"{label:40s} {e0.elapsed_time(e1)/5:.2f} ms", flush=True)
def fill(ptr_or_tensor):
    if isinstance(ptr_or_tensor, torch.Tensor):
        ptr_or_tensor[:n].copy_(src); return ptr_or_tensor.data_ptr()
    _hip.check(L.cleora_memcpy_d2d(ptr_or_tensor, src.data_ptr(), n * d * 4, s)); torch.cuda.synchronize(); return ptr_or_tensor
for rnd in range(2):
    run(fill(early[0]), early[1].data_ptr(), "x=early0 y=early1")
    run(fill(early[1]), early[0].data_ptr(), "x=early1 y=early0")
    run(fill(late[0]), late[1].data_ptr(), "x=late0 y=late1")
    run(fill(late[1]), late[2].data_ptr(), "x=late1 y=late2")
    run(fill(late[2]), late[0].data_ptr(), "x=late2 y=late0")
    run(fill(early[0]), late[0].data_ptr(), "x=early0 y=late0")
    run(fill(late[0]), early[0].data_ptr(), "x=late0 y=early0")
    if contig[0] and contig[1]:
        run(fill(contig[0]), contig[1], "x=contig0 y=contig1")
        run(fill(contig[1]), contig[0], "x=contig1 y=contig0")
