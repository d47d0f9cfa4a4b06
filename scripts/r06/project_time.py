"""Round 6: milliseconds of the bounded-operand projection (csrc/project_f16.hip) at the C3 shape — for A/B runs of build flags:
    python scripts/r06/project_time.py [label]"""
import ctypes, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cleora_amd import _hip
n, d = 9_999_997, 256
dev = torch.device("cuda:0"); L = _hip.lib(); S = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device=dev); g.manual_seed(3)
x = torch.randn((n, d), generator=g, device=dev); x /= x.norm(dim=1, keepdim=True)
bound = torch.ones(n, device=dev); rs = (2 * torch.rand(n, generator=g, device=dev) - 1).contiguous()
mean = (torch.randn(d, generator=g, device=dev) * 0.05).contiguous(); t = torch.randn((d, d), generator=g, device=dev).contiguous()
out = torch.empty((n, d), device=dev)
nd, fm = ctypes.c_int(0), ctypes.c_int(-1)
def call():
    _hip.check(L.cleora_project_bounded_dev(x.data_ptr(), d, n, d, mean.data_ptr(), t.data_ptr(), d, out.data_ptr(), d, rs.data_ptr(), bound.data_ptr(), 1, ctypes.byref(nd), ctypes.byref(fm), S))
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ms = []
for rep in range(3):
    call(); call()
    ev[0].record()
    for _ in range(10): call()
    ev[1].record(); torch.cuda.synchronize()
    ms.append(round(ev[0].elapsed_time(ev[1]) / 10, 3))
ref = (x[:4096].double() - rs[:4096].double()[:, None] * mean.double()[None, :]) @ t.double()
ref /= ref.norm(dim=1, keepdim=True)
err = float(((out[:4096].double() - ref).norm(dim=1) / ref.norm(dim=1)).max())
print(json.dumps({"label": sys.argv[1] if len(sys.argv) > 1 else "", "form": fm.value, "ms": ms, "max_row_err_first_4096": err}))
