#!/usr/bin/env python3
"""Developer probe (round 2): per-stage time of cleora_whiten_dev (statistics | Gram | eigensolver | projection) at the
BASELINE shapes, from the library's own HIP-event timing (cleora_whiten_set_timing).
CLEORA_PROJECT_TILED=1 selects the first (128x128x32 LDS-tiled) projection kernel for A/B."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cleora_amd import _hip
dev = torch.device("cuda:0")
L = _hip.lib()
shapes = [(10_000_000, 256), (1_000_000, 256), (4_000_000, 512), (2_000_000, 1024), (10_000_000, 128)]
if len(sys.argv) > 1:
    shapes = shapes[: int(sys.argv[1])]
s = torch.cuda.current_stream().cuda_stream
for n, d in shapes:
    x = torch.randn((n, d), device=dev)
    x = (x / x.norm(dim=1, keepdim=True)).contiguous()
    y = torch.empty_like(x)
    ws = torch.empty(L.cleora_whiten_workspace(n, d), dtype=torch.uint8, device=dev)
    for _ in range(2):
        _hip.check(L.cleora_whiten_dev(x.data_ptr(), d, n, d, d, y.data_ptr(), d, ws.data_ptr(), None, s))
    torch.cuda.synchronize()
    _hip.check(L.cleora_whiten_set_timing(1))
    reps = 5
    for _ in range(reps):
        _hip.check(L.cleora_whiten_dev(x.data_ptr(), d, n, d, d, y.data_ptr(), d, ws.data_ptr(), None, s))
    ms, c = (ctypes.c_double * 4)(), ctypes.c_uint64(0)
    _hip.check(L.cleora_whiten_get_timing(ctypes.byref(ms), ctypes.byref(c)))
    _hip.check(L.cleora_whiten_set_timing(0))
    st, gr, eg, pr = (ms[i] / c.value for i in range(4))
    cov = torch.cov(y[: min(n, 1_000_000)].double().T)
    err = float((cov - torch.eye(d, device=dev, dtype=torch.float64)).abs().max())
    print(f"n={n} d={d}: stats {st:.3f}  gram {gr:.3f}  eigh {eg:.3f}  project {pr:.3f} ms   total {st+gr+eg+pr:.3f}   "
          f"project {2.0*n*d*d/pr/1e9:.1f} TF/s   |cov-I|max {err:.2e}   tiled={os.environ.get('CLEORA_PROJECT_TILED', '0')}", flush=True)
    del x, y, ws
