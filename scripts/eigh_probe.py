#!/usr/bin/env python3
"""Developer probe: rocSOLVER symmetric eigensolvers on the d x d covariance (dsyevd vs dsyevj vs dsyevdj)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cleora_amd import _hip
_hip.lib()
R = ctypes.CDLL(os.environ["CLEORA_ROCSOLVER"])
B = ctypes.CDLL(os.path.join(os.path.dirname(os.environ["CLEORA_ROCSOLVER"]), "librocblas.so"))
vp, ci, cd = ctypes.c_void_p, ctypes.c_int, ctypes.c_double
h = vp()
B.rocblas_create_handle.argtypes = [ctypes.POINTER(vp)]
assert B.rocblas_create_handle(ctypes.byref(h)) == 0
B.rocblas_set_stream.argtypes = [vp, vp]
B.rocblas_set_stream(h, torch.cuda.current_stream().cuda_stream)
R.rocsolver_dsyevd.argtypes = [vp, ci, ci, ci, vp, ci, vp, vp, vp]
R.rocsolver_dsyevj.argtypes = [vp, ci, ci, ci, ci, vp, ci, cd, vp, ci, vp, vp, vp]
R.rocsolver_dsyevdj.argtypes = [vp, ci, ci, ci, vp, ci, vp, vp]
dev = torch.device("cuda:0")
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for d in (64, 128, 256, 512, 1024):
    g = torch.randn((d, d), device=dev, dtype=torch.float64); cov = g @ g.T / d
    a = torch.empty_like(cov); w = torch.empty(d, device=dev, dtype=torch.float64); e = torch.empty_like(w)
    info = torch.zeros(1, device=dev, dtype=torch.int32); res = torch.zeros(1, device=dev, dtype=torch.float64); ns = torch.zeros(1, device=dev, dtype=torch.int32)
    def f_d():
        a.copy_(cov); assert R.rocsolver_dsyevd(h, 211, 121, d, a.data_ptr(), d, w.data_ptr(), e.data_ptr(), info.data_ptr()) == 0
    def f_j():
        a.copy_(cov); assert R.rocsolver_dsyevj(h, 252, 211, 121, d, a.data_ptr(), d, 0.0, res.data_ptr(), 100, ns.data_ptr(), w.data_ptr(), info.data_ptr()) == 0
    def f_dj():
        a.copy_(cov); assert R.rocsolver_dsyevdj(h, 211, 121, d, a.data_ptr(), d, w.data_ptr(), info.data_ptr()) == 0
    ref = torch.linalg.eigvalsh(cov)
    out = {}
    for name, f in (("dsyevd", f_d), ("dsyevj", f_j), ("dsyevdj", f_dj)):
        try:
            ms = timed(f); err = float((w - ref).abs().max() / ref.abs().max())
            v = a.T  # column-major eigenvectors
            orth = float((v.T @ v - torch.eye(d, device=dev, dtype=torch.float64)).abs().max())
            out[name] = f"{ms:.2f} ms (eig err {err:.1e}, orth {orth:.1e}, sweeps {int(ns)})"
        except Exception as ex:
            out[name] = f"failed {ex}"
    print(d, out, flush=True)
