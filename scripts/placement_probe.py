#!/usr/bin/env python3
"""Developer probe (round 2): what decides the "placement class" of an iterate buffer?

Hypothesis from round 1: the class is a band of PHYSICAL addresses (HBM stack-ID rank); a pair of buffers in the
same band is ~12 % slower for the SpMM (X gathers + Y stores) than a pair in different bands.
 (1) spacer sweep: A is allocated first; for S in 0..200 GB a spacer of S bytes is allocated, then B, then the
     spacer is freed; time SpMM(A -> B).  If the class is a physical band, t(S) is a step function of S.
 (2) is a plain streaming copy A -> B (read stream + write stream, no gathers) sensitive to the same pairing?
     If it is, the library can classify a pair in ~3 ms without a graph.
 (3) three buffers allocated ~96 GB apart: all three pairs fast?
raw hipMalloc / hipFree through the runtime torch loaded, so torch's caching allocator does not interfere."""
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
nbytes = n * d * 4
GB = 1 << 30
def alloc(b):
    p = ctypes.c_void_p()
    rc = rt.hipMalloc(ctypes.byref(p), ctypes.c_size_t(b))
    return p.value if rc == 0 else None
def free(p):
    rt.hipFree(ctypes.c_void_p(p))
src = torch.randn((n, d), device=dev); src /= src.norm(dim=1, keepdim=True)
def fill(p):
    rt.hipMemcpy(ctypes.c_void_p(p), ctypes.c_void_p(src.data_ptr()), ctypes.c_size_t(nbytes), ctypes.c_int(3))
def view(p):
    # a torch view of raw memory for the copy-kernel test
    class Holder: pass
    h = Holder()
    h.__cuda_array_interface__ = {"shape": (n * d,), "typestr": "<f4", "data": (p, False), "version": 2}
    return torch.as_tensor(h, device=dev)
def t_spmm(xp, yp, reps=2):
    for _ in range(2):
        _hip.check(L.cleora_propagate_dev(graph.handle, 0, xp, d, d, yp, d, _hip.F_L2NORM, 0.0, None, None, None, s))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        _hip.check(L.cleora_propagate_dev(graph.handle, 0, xp, d, d, yp, d, _hip.F_L2NORM, 0.0, None, None, None, s))
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
def t_copy(xp, yp, reps=5):
    a, b = view(xp), view(yp)
    torch.mul(a, 1.0, out=b); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        torch.mul(a, 1.0, out=b)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
free_b, total_b = torch.cuda.mem_get_info()
print(f"free {free_b / GB:.1f} GB of {total_b / GB:.1f} GB", flush=True)
A = alloc(nbytes); fill(A)
print("(1) spacer sweep   S_GB   va(B)-va(A)_GB   spmm A->B ms   spmm B->A ms   copy A->B ms", flush=True)
for S in (0, 10, 20, 30, 40, 50, 60, 70, 80, 90, 100, 110, 120, 140, 160, 180, 200, 220):
    if (S + 12) * GB > torch.cuda.mem_get_info()[0]:
        break
    sp = alloc(S * GB) if S else None
    B = alloc(nbytes)
    if sp: free(sp)
    if B is None:
        print(f"  {S:4d}  alloc failed"); continue
    fill(A); ab = t_spmm(A, B); fill(B); ba = t_spmm(B, A); fill(A)
    print(f"  {S:4d}   {(B - A) / GB:8.1f}   {ab:7.2f}   {ba:7.2f}   {t_copy(A, B):6.3f}", flush=True)
    free(B)
print("(3) three buffers ~96 GB apart", flush=True)
sp1 = alloc(86 * GB); B = alloc(nbytes); sp2 = alloc(86 * GB); C = alloc(nbytes)
for p_ in (sp1, sp2):
    if p_: free(p_)
if B and C:
    for name, (x, y) in {"A->B": (A, B), "B->C": (B, C), "C->A": (C, A), "A->C": (A, C)}.items():
        fill(x)
        print(f"  {name}: spmm {t_spmm(x, y):6.2f} ms   copy {t_copy(x, y):6.3f} ms", flush=True)
