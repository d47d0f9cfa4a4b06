"""A/B on one box: the long-row threshold (rows scheduled first in the main launch) at a BASELINE workload, interleaved."""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
import bench
from cleora_amd import _hip

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
dev = torch.device("cuda:0")
L = _hip.lib()


class A:
    config, nodes, pairs, hyperedges, products, dim = cfg, 0, 0, 0, 0, 0


g, hashes, label, c = bench.make_workload(A, dev, 0, 1, False)
n, nnz, d = g["n"], g["nnz"], c["dim"]
graphs = {t: _hip.Graph.from_device(n, n, nnz, g["rowptr"].data_ptr(), g["col"].data_ptr(), g["val_left"].data_ptr(), None, 0, t, 0, keepalive=g)
          for t in (128, 256, 512, 1024, 2048)}
(xa, xb), ms = _hip.DevArray.iterates(graphs[1024], n, d, 2)
a, b = torch.as_tensor(xa, device=dev), torch.as_tensor(xb, device=dev)
s = torch.cuda.current_stream().cuda_stream
_hip.check(L.cleora_init_dev(hashes.data_ptr(), n, d, 0, a.data_ptr(), d, s))


def run(graph, iters):
    global a, b
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        _hip.check(L.cleora_propagate_dev(graph.handle, 0, a.data_ptr(), d, d, b.data_ptr(), d, _hip.F_L2NORM, 0.0, None, None, None, s))
        a, b = b, a
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


out = {"config": cfg, "placement_ms": ms, "ms_per_iteration": {}, "long_rows": {}}
for t, gr in graphs.items():
    run(gr, 4)
    out["long_rows"][str(t)] = [int(gr.info().n_hub_rows), int(gr.info().n_inorder_rows), int(gr.info().hub_inorder_min)]
out["span_parts_ms"] = {}                     # [fork of the hub launch | main kernel | wait for the hub launch] per SpMM, HIP events
for rep in range(3):
    for t, gr in graphs.items():
        gr.set_timing(True)
        out["ms_per_iteration"].setdefault(str(t), []).append(round(run(gr, 10), 3))
        ms3, calls = gr.get_timing()
        gr.set_timing(False)
        out["span_parts_ms"].setdefault(str(t), []).append([round(v / max(calls, 1), 3) for v in ms3])
print(json.dumps(out))
