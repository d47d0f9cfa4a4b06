#!/usr/bin/env python3
"""Round-6 workload for the whitening-kernel profiles (rocprofv3 --kernel-trace --stats / --pmc): at the C3 shape, five launches each of
the intermediate statistics (gram16_kernel), the f64 statistics (gram_kernel), the six-product bf16 projection (project_split_kernel,
plain and loop form) and the bounded projection of the loop's intermediate iterations (project_f16_kernel)."""
import ctypes, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cleora_amd import _hip
n = int(sys.argv[1]) if len(sys.argv) > 1 else 9_999_997
d = 256
dev = torch.device("cuda:0")
L = _hip.lib()
s = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device=dev); g.manual_seed(1)
x = torch.randn((n, d), device=dev, generator=g) * torch.linspace(0.3, 2.0, d, device=dev) + 0.05
x = (x / x.norm(dim=1, keepdim=True)).contiguous()
ws = torch.empty(L.cleora_whiten_workspace(n, d), dtype=torch.uint8, device=dev)
m64 = torch.empty(d, dtype=torch.float64, device=dev)
g64 = torch.empty((d, d), dtype=torch.float64, device=dev)
mean32 = x[: 100_000].mean(0).contiguous()
t = (torch.randn((d, d), device=dev, generator=g) / d ** 0.5).contiguous()
rowscale = torch.ones(n, device=dev)
y = torch.empty_like(x)
nd, fm = ctypes.c_int(0), ctypes.c_int(0)


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


out = {"n": n, "d": d}
out["stats_f64_ms"] = timed(lambda: _hip.check(L.cleora_whiten_stats_dev(x.data_ptr(), d, n, d, ws.data_ptr(), 0, m64.data_ptr(), g64.data_ptr(), s)), 3)
out["stats_intermediate_ms"] = timed(lambda: _hip.check(L.cleora_whiten_stats_dev(x.data_ptr(), d, n, d, ws.data_ptr(), 1, m64.data_ptr(), g64.data_ptr(), s)))
out["project_bf16x6_plain_ms"] = timed(lambda: _hip.check(L.cleora_project_dev(x.data_ptr(), d, n, d, mean32.data_ptr(), t.data_ptr(), d, y.data_ptr(), d, s)))
out["project_bf16x6_loop_form_ms"] = timed(lambda: _hip.check(L.cleora_project_general_dev(x.data_ptr(), d, n, d, mean32.data_ptr(), t.data_ptr(), d, y.data_ptr(), d, rowscale.data_ptr(), None, 0, 1.0, 0.0, 1, ctypes.byref(nd), s)))
out["project_f16_loop_form_ms"] = timed(lambda: _hip.check(L.cleora_project_bounded_dev(x.data_ptr(), d, n, d, mean32.data_ptr(), t.data_ptr(), d, y.data_ptr(), d, rowscale.data_ptr(), None, 1, ctypes.byref(nd), ctypes.byref(fm), s)))
print(json.dumps(out))
