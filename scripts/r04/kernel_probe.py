#!/usr/bin/env python3
"""Round-4 developer probe: milliseconds of the whitened loop's kernels at a shape (HIP events on torch's stream), and the
accuracy of the intermediate statistics against the f64 form.  python scripts/r04/kernel_probe.py [n d] -> one JSON line."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cleora_amd import _hip
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda:0")
L = _hip.lib()
s = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device=dev); g.manual_seed(1)
x = torch.randn((n, d), device=dev, generator=g) * torch.linspace(0.3, 2.0, d, device=dev) + 0.05
x = (x / x.norm(dim=1, keepdim=True)).contiguous()
ws = torch.empty(L.cleora_whiten_workspace(n, d), dtype=torch.uint8, device=dev)
m64 = torch.empty(d, dtype=torch.float64, device=dev)
g64 = torch.empty((d, d), dtype=torch.float64, device=dev)
gi = torch.empty((d, d), dtype=torch.float64, device=dev)

def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

out = {"n": n, "d": d, "env": {k: v for k, v in os.environ.items() if k.startswith("CLEORA_X")}}
out["stats_f64_ms"] = timed(lambda: _hip.check(L.cleora_whiten_stats_dev(x.data_ptr(), d, n, d, ws.data_ptr(), 0, m64.data_ptr(), g64.data_ptr(), s)), 3)
out["stats_intermediate_ms"] = timed(lambda: _hip.check(L.cleora_whiten_stats_dev(x.data_ptr(), d, n, d, ws.data_ptr(), 1, m64.data_ptr(), gi.data_ptr(), s)))
out["gram_rel_frobenius_vs_f64"] = float((gi - g64).norm() / g64.norm())
dg, di = torch.diagonal(g64), torch.diagonal(gi)
out["gram_mean_rel_diag_bias"] = float(((di - dg) / dg).mean())
out["gram_max_rel_diag_err"] = float(((di - dg) / dg).abs().max())
mean32 = m64.float()
t = (torch.randn((d, d), device=dev, generator=g) / d ** 0.5).contiguous()
y = torch.empty_like(x)
out["project_plain_ms"] = timed(lambda: _hip.check(L.cleora_project_dev(x.data_ptr(), d, n, d, mean32.data_ptr(), t.data_ptr(), d, y.data_ptr(), d, s)))
rs = torch.ones(n, device=dev)
import ctypes
done = ctypes.c_int(0)
out["project_loop_form_ms"] = timed(lambda: _hip.check(L.cleora_project_general_dev(x.data_ptr(), d, n, d, mean32.data_ptr(), t.data_ptr(), d, y.data_ptr(), d,
                                                                                    rs.data_ptr(), None, 0, 1.0, 0.0, 1, ctypes.byref(done), s)))
out["project_norm_done"] = done.value
out["project_norm_only_ms"] = timed(lambda: _hip.check(L.cleora_project_general_dev(x.data_ptr(), d, n, d, mean32.data_ptr(), t.data_ptr(), d, y.data_ptr(), d,
                                                                                    None, None, 0, 1.0, 0.0, 1, ctypes.byref(done), s)))
out["project_scale_only_ms"] = timed(lambda: _hip.check(L.cleora_project_general_dev(x.data_ptr(), d, n, d, mean32.data_ptr(), t.data_ptr(), d, y.data_ptr(), d,
                                                                                     rs.data_ptr(), None, 0, 1.0, 0.0, 0, ctypes.byref(done), s)))
x2 = torch.roll(x, 1, 0)
out["project_loop_form_blend_ms"] = timed(lambda: _hip.check(L.cleora_project_general_dev(x.data_ptr(), d, n, d, mean32.data_ptr(), t.data_ptr(), d, y.data_ptr(), d,
                                                                                          rs.data_ptr(), x2.data_ptr(), d, 0.8, 0.2, 1, ctypes.byref(done), s)))
# the wide launch (128-row tiles) against a narrow one (64-row tiles, one barrier per k-step) on the same rows: bit for bit;
# and against an f64 product
m = 8192
for label, scale, norm in (("plain", None, 0), ("loop_form", rs, 1)):
    _hip.check(L.cleora_project_general_dev(x.data_ptr(), d, n, d, mean32.data_ptr(), t.data_ptr(), d, y.data_ptr(), d,
                                            scale.data_ptr() if scale is not None else None, None, 0, 1.0, 0.0, norm, ctypes.byref(done), s))
    small = torch.empty((m, d), device=dev)
    eq = True
    for r0 in (0, (n // 2) // 128 * 128, n - m):
        _hip.check(L.cleora_project_general_dev(x[r0:].data_ptr(), d, m, d, mean32.data_ptr(), t.data_ptr(), d, small.data_ptr(), d,
                                                scale[r0:].data_ptr() if scale is not None else None, None, 0, 1.0, 0.0, norm, ctypes.byref(done), s))
        torch.cuda.synchronize()
        eq = eq and bool(torch.equal(small, y[r0:r0 + m]))
    out[f"project_{label}_wide_bit_equal_to_narrow"] = eq
    want = (x[:m].double() - mean32.double()) @ t.double()
    if norm:
        want = want / want.norm(dim=1, keepdim=True)
    out[f"project_{label}_max_err_vs_f64"] = float((y[:m].double() - want).abs().max() / want.abs().max())
print(json.dumps(out), flush=True)
