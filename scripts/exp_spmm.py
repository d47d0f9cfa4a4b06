#!/usr/bin/env python3
"""Driver for scripts/exp_spmm.hip (experiments only)."""
import ctypes, os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cleora_amd import synth
L = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libexp.so"))
vp = ctypes.c_void_p
L.exp_launch.argtypes = [ctypes.c_int, vp, vp, vp, vp, vp, ctypes.c_uint64, ctypes.c_uint32, vp, vp]
dev = torch.device("cuda:0")
g = synth.power_law_graph(10_000_000, 95_000_000, 2, dev)
n, nnz, d = g["n"], g["nnz"], 256
deg = torch.diff(g["rowptr"])
main = deg <= 1024
nnz_main = int(deg[main].sum())
bytes_main = nnz_main * 8 + (n + 1) * 8 + nnz_main * d * 4 + n * d * 4
x = torch.randn((n, d), device=dev, dtype=torch.float32)
x /= x.norm(dim=1, keepdim=True)
y = torch.zeros_like(x)
y0 = None
s = torch.cuda.current_stream().cuda_stream
order_lpt = torch.argsort(deg, descending=True).to(torch.int32)
names = {0: "base U8 blk256", 1: "nt col/val/Y", 2: "nt all", 3: "blk64", 4: "U16", 5: "U4", 6: "rolling U8",
         7: "lds R32", 8: "lds R64", 9: "blk128"}
def run(v, order=None, reps=5):
    op = order.data_ptr() if order is not None else None
    for _ in range(2):
        rc = L.exp_launch(v, g["rowptr"].data_ptr(), g["col"].data_ptr(), g["val_left"].data_ptr(), x.data_ptr(), y.data_ptr(), n, 1024, op, s)
        assert rc == 0, rc
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        L.exp_launch(v, g["rowptr"].data_ptr(), g["col"].data_ptr(), g["val_left"].data_ptr(), x.data_ptr(), y.data_ptr(), n, 1024, op, s)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for rnd in range(2):
    for v in sorted(names):
        y.zero_()
        ms = run(v)
        if y0 is None: y0 = y.clone()
        same = bool(torch.equal(y, y0))
        print(json.dumps({"round": rnd, "variant": v, "name": names[v], "ms": round(ms, 3), "GBps": round(bytes_main / ms / 1e6, 1), "bit_equal_to_base": same}), flush=True)
ms = run(0, order_lpt)
print(json.dumps({"variant": "0+LPT order", "ms": round(ms, 3), "GBps": round(bytes_main / ms / 1e6, 1), "same": bool(torch.equal(y, y0))}))
ms = run(3, order_lpt)
print(json.dumps({"variant": "3+LPT order", "ms": round(ms, 3), "GBps": round(bytes_main / ms / 1e6, 1)}))
