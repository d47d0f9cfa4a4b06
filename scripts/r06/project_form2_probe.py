import ctypes, json, os, sys, torch
sys.path.insert(0, "/root/repo")
from cleora_amd import _hip
n, d = 9_999_997, 256
dev = torch.device("cuda:0"); L = _hip.lib(); S = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device=dev); g.manual_seed(3)
x = torch.randn((n, d), generator=g, device=dev); x /= x.norm(dim=1, keepdim=True)
bound = torch.ones(n, device=dev); rs = (2 * torch.rand(n, generator=g, device=dev) - 1).contiguous()
mean = (torch.randn(d, generator=g, device=dev) * 0.05).contiguous(); t = torch.randn((d, d), generator=g, device=dev).contiguous()
buf = torch.empty(n * d + 4, device=dev)
nd, fm = ctypes.c_int(0), ctypes.c_int(-1)
res = {}
for name, off in (("aligned (form 1: project_f16_kernel)", 0), ("out shifted by one float (form 2: bounded split mode)", 4)):
    op = buf.data_ptr() + off
    def call():
        _hip.check(L.cleora_project_bounded_dev(x.data_ptr(), d, n, d, mean.data_ptr(), t.data_ptr(), d, op, d, rs.data_ptr(), bound.data_ptr(), 1, ctypes.byref(nd), ctypes.byref(fm), S))
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ms = []
    for rep in range(2):
        call(); call()
        ev[0].record()
        for _ in range(10): call()
        ev[1].record(); torch.cuda.synchronize()
        ms.append(round(ev[0].elapsed_time(ev[1]) / 10, 3))
    res[name] = {"form": fm.value, "norm_done": nd.value, "ms": ms}
print(json.dumps(res))
