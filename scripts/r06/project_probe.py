"""Round 6: the bounded-operand projection (csrc/project_f16.hip: three f16 MFMAs per product, transform in registers) against the
six-product bf16 form (project_split_kernel) at the C3 shape — accuracy against an f64 product on a row sample, and time.

    python scripts/r06/project_probe.py [n_rows]
"""
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cleora_amd import _hip  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 9_999_997
d = 256
dev = torch.device("cuda:0")
L = _hip.lib()
S = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device=dev)
g.manual_seed(3)
# Z = A Y -like rows: inside the ball of radius B_r (B_r = sum |a| of the row: 1 for a left Markov matrix, anything otherwise)
bound = torch.where(torch.rand(n, generator=g, device=dev) < 0.9, torch.ones(n, device=dev), 10 ** (torch.rand(n, generator=g, device=dev) * 4 - 1))
x = torch.randn((n, d), generator=g, device=dev)
x /= x.norm(dim=1, keepdim=True)
x *= (bound * (0.2 + 0.8 * torch.rand(n, generator=g, device=dev)))[:, None]
rowscale = (bound * (2 * torch.rand(n, generator=g, device=dev) - 1)).contiguous()       # |s| <= B
mean = (torch.randn(d, generator=g, device=dev) * 0.05).clamp_(-1, 1).contiguous()
t = torch.randn((d, d), generator=g, device=dev) * (10 ** (torch.rand(d, generator=g, device=dev) * 5 - 2))[None, :]
t = t.contiguous()
out = torch.empty((n, d), device=dev)
res = {"n": n, "d": d}
sample = torch.randperm(n, generator=g, device=dev)[:50_000]


def reference(norm, scaled):
    o = x[sample].double() - (rowscale[sample].double()[:, None] if scaled else 1.0) * mean.double()[None, :]
    o = o @ t.double()
    if norm == 1:
        o = o / o.norm(dim=1, keepdim=True).clamp_min(1e-10)
    elif norm == 2:
        o = o / o.abs().sum(dim=1, keepdim=True).clamp_min(1e-10)
    return o


def call(form, norm, scaled):
    nd, fm = ctypes.c_int(0), ctypes.c_int(-1)
    rs = rowscale.data_ptr() if scaled else None
    if form == "f16":
        _hip.check(L.cleora_project_bounded_dev(x.data_ptr(), d, n, d, mean.data_ptr(), t.data_ptr(), d, out.data_ptr(), d, rs, bound.data_ptr(), norm,
                                                ctypes.byref(nd), ctypes.byref(fm), S))
        assert fm.value == 1
    else:
        _hip.check(L.cleora_project_general_dev(x.data_ptr(), d, n, d, mean.data_ptr(), t.data_ptr(), d, out.data_ptr(), d, rs, None, 0, 1.0, 0.0, norm,
                                                ctypes.byref(nd), S))
    assert norm == 0 or nd.value == 1


FORMS = ("f16", "bf16x6")


def select(form):
    return form


for norm in (1, 0, 2):
    for scaled in (True, False):
        ref = reference(norm, scaled)
        key = f"norm{norm}_{'scaled' if scaled else 'plain'}"
        res[key] = {}
        for form in FORMS:
            out.zero_()
            call(select(form), norm, scaled)
            torch.cuda.synchronize()
            got = out[sample].double()
            err = (got - ref).norm(dim=1) / ref.norm(dim=1).clamp_min(1e-300)
            el = ((got - ref).abs() / ref.abs().amax(dim=1, keepdim=True).clamp_min(1e-300)).max()
            res[key][form] = {"max_row_rel_err": float(err.max()), "mean_row_rel_err": float(err.mean()), "max_elem_err_rel_to_row_max": float(el),
                              "finite": bool(torch.isfinite(out).all())}
print(json.dumps(res), flush=True)
# time
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
res["ms"] = {}
for rep in range(2):
    for form in FORMS:
        f = select(form)
        for _ in range(2):
            call(f, 1, True)
        ev[0].record()
        for _ in range(10):
            call(f, 1, True)
        ev[1].record()
        torch.cuda.synchronize()
        res["ms"].setdefault(form, []).append(round(ev[0].elapsed_time(ev[1]) / 10, 3))
res["hbm_floor_ms_at_6.4TBps"] = round(2 * n * d * 4 / 6.4e12 * 1e3, 3)
print(json.dumps(res), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "r06_project_probe.json"), "w"), indent=1)
