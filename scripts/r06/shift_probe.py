"""Round 6: is the 10-12 % "placement class" of an iterate pair a function of the buffers' physical ALIGNMENT against the
workgroup -> XCD round robin?  Block b of the SpMM runs on XCD b % 8 and writes four consecutive 1 KiB rows: XCD k writes the 4 KiB
chunks k, k + 8, ... of Y.  If memory interleaves across the HBM stacks in 4 KiB units, the offset of Y's base decides whether every
XCD writes near or far.  The probe times y <- A x (fixed x, no ping-pong) with y's base shifted by multiples of 1 KiB inside one
allocation, then x's base, on two different allocation pairs.

    python scripts/r06/shift_probe.py [C3]
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from cleora_amd import _hip  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
dev = torch.device("cuda:0")
L = _hip.lib()
S = torch.cuda.current_stream().cuda_stream


class A:
    config, nodes, pairs, hyperedges, products, dim = cfg, 0, 0, 0, 0, 0


g, hashes, label, c = bench.make_workload(A, dev, 0, 1, False)
n, nnz, d = g["n"], g["nnz"], c["dim"]
graph = _hip.Graph.from_device(n, n, nnz, g["rowptr"].data_ptr(), g["col"].data_ptr(), g["val_left"].data_ptr(), None, 0,
                               keepalive=(g["rowptr"], g["col"], g["val_left"]))
SLACK = 64 * 1024 // 4          # floats


def launch_ms(xp, yp, iters=8, warm=3):
    for _ in range(warm):
        _hip.check(L.cleora_propagate_dev(graph.handle, 0, xp, d, d, yp, d, _hip.F_L2NORM, 0.0, None, None, None, S))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        _hip.check(L.cleora_propagate_dev(graph.handle, 0, xp, d, d, yp, d, _hip.F_L2NORM, 0.0, None, None, None, S))
    torch.cuda.synchronize()
    return round((time.perf_counter() - t0) / iters * 1e3, 3)


out = {"config": cfg, "n": n, "nnz": nnz, "d": d, "pairs": []}
junk = []
for pair in range(3):
    xb = torch.empty(n * d + SLACK, dtype=torch.float32, device=dev)
    yb = torch.empty(n * d + SLACK, dtype=torch.float32, device=dev)
    _hip.check(L.cleora_init_dev(hashes.data_ptr(), n, d, 0, xb.data_ptr(), d, S))
    for _ in range(4):          # arm the gather cache policy
        _hip.check(L.cleora_propagate_dev(graph.handle, 0, xb.data_ptr(), d, d, yb.data_ptr(), d, _hip.F_L2NORM, 0.0, None, None, None, S))
    rec = {"x": hex(xb.data_ptr()), "y": hex(yb.data_ptr())}
    rec["y_shift_KiB"] = {str(k): launch_ms(xb.data_ptr(), yb.data_ptr() + k * 1024) for k in list(range(0, 33)) + [36, 40, 48, 64 - 1]}
    # x shifted: E_0 is re-initialised at the shifted base each time
    xs = {}
    for k in (0, 1, 2, 4, 8, 12, 16, 24, 32):
        _hip.check(L.cleora_init_dev(hashes.data_ptr(), n, d, 0, xb.data_ptr() + k * 1024, d, S))
        xs[str(k)] = launch_ms(xb.data_ptr() + k * 1024, yb.data_ptr())
    rec["x_shift_KiB"] = xs
    out["pairs"].append(rec)
    print(json.dumps(rec), flush=True)
    # churn the allocator so that the next pair gets other physical pages
    junk.append(xb)
    del yb
    torch.cuda.empty_cache()
    junk.append(torch.empty((1 << 30) * (pair + 1) // 4, dtype=torch.float32, device=dev))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"r06_shift_probe_{cfg}.json"), "w"), indent=1)
