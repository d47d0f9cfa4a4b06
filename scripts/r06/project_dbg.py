"""Round 6 diagnostic: the transposed f16 projection with its loads / its stores taken out (development builds only)."""
import ctypes, sys, os, json
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
res = {}
for rep in range(2):
    for name, form in (("transposed", 0), ("staged", 1), ("transposed_no_reloads", 2), ("transposed_no_stores", 3)):
        L.cleora_dev_project_form(form)
        call(); call()
        ev[0].record()
        for _ in range(10): call()
        ev[1].record(); torch.cuda.synchronize()
        res.setdefault(name, []).append(round(ev[0].elapsed_time(ev[1]) / 10, 3))
L.cleora_dev_project_form(4)
call(); call(); torch.cuda.synchronize()
tm = out[0, :32].cpu().tolist()
t2 = out[0, 32:64].cpu().tolist()
res["finish_parts_partials_barrier_epilogue"] = [t2[4 * w:4 * w + 3] for w in range(8)]
res["cycles_wave_w_products_finish_total_tiles (s_memtime, 100 MHz)"] = [tm[4 * w:4 * w + 4] for w in range(8)]
L.cleora_dev_project_form(0)
print(json.dumps(res))
