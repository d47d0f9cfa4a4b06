"""A/B on one box: the SpMM + fused L2 iteration at a BASELINE workload with the in-order hub launch in its 4-lane and 2-lane shapes
(cleora_graph_set_hub_lanes) and with the segmented hub sum, interleaved so that box and placement are the same for all."""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
import bench
from cleora_amd import _hip, synth

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
dev = torch.device("cuda:0")
L = _hip.lib()


class A:
    config, nodes, pairs, hyperedges, products, dim = cfg, 0, 0, 0, 0, 0


g, hashes, label, c = bench.make_workload(A, dev, 0, 1, False)
n, nnz, d = g["n"], g["nnz"], c["dim"]
graph = _hip.Graph.from_device(n, n, nnz, g["rowptr"].data_ptr(), g["col"].data_ptr(), g["val_left"].data_ptr(), None, 0, keepalive=g)
(xa, xb), ms = _hip.DevArray.iterates(graph, n, d, 2)
a, b = torch.as_tensor(xa, device=dev), torch.as_tensor(xb, device=dev)
s = torch.cuda.current_stream().cuda_stream
_hip.check(L.cleora_init_dev(hashes.data_ptr(), n, d, 0, a.data_ptr(), d, s))


def run(flags, iters):
    global a, b
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        _hip.check(L.cleora_propagate_dev(graph.handle, 0, a.data_ptr(), d, d, b.data_ptr(), d, _hip.F_L2NORM | flags, 0.0, None, None, None, s))
        a, b = b, a
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


run(0, 6)
out = {"config": cfg, "n": n, "nnz": nnz, "d": d, "placement_ms": ms, "ms_per_iteration": {"lanes4": [], "lanes2": [], "segments": []}}
out["ms_per_iteration"] = {}
variants = (("min1024_lanes4", 1024, 4, 0), ("min8192_lanes4", 8192, 4, 0), ("min32768_lanes4", 32768, 4, 0), ("min8192_lanes2", 8192, 2, 0),
            ("segments", 8192, 0, _hip.F_HUB_SEGMENTS))
for rep in range(3):
    for name, min_edges, lanes, flags in variants:
        graph.set_hub_inorder_min(min_edges)
        graph.set_hub_lanes(lanes)
        run(flags, 2)
        out["ms_per_iteration"].setdefault(name, []).append(round(run(flags, 10), 3))
out["n_inorder_rows"] = {}
for m in (1024, 8192, 32768):
    graph.set_hub_inorder_min(m)
    out["n_inorder_rows"][str(m)] = int(graph.info().n_inorder_rows)
print(json.dumps(out))
