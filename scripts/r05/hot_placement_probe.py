"""Does the PLACEMENT of a graph handle's private hot-column copy (col with the gather cache policy's marks: 4 B per edge) move the
SpMM between the fast and the slow class, like the iterates' placement does?  One graph, one pair of iterates; the copy is dropped,
a spacer of growing size is allocated, the copy is rebuilt (somewhere else), the iteration is timed again."""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
import bench
from cleora_amd import _hip

dev = torch.device("cuda:0")
L = _hip.lib()


class A:
    config, nodes, pairs, hyperedges, products, dim = "C3", 0, 0, 0, 0, 0


g, hashes, label, c = bench.make_workload(A, dev, 0, 1, False)
n, nnz, d = g["n"], g["nnz"], c["dim"]
graph = _hip.Graph.from_device(n, n, nnz, g["rowptr"].data_ptr(), g["col"].data_ptr(), g["val_left"].data_ptr(), None, 0, keepalive=g)
(xa, xb), ms = _hip.DevArray.iterates(graph, n, d, 2)
a, b = torch.as_tensor(xa, device=dev), torch.as_tensor(xb, device=dev)
s = torch.cuda.current_stream().cuda_stream
_hip.check(L.cleora_init_dev(hashes.data_ptr(), n, d, 0, a.data_ptr(), d, s))


def run(iters):
    global a, b
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        _hip.check(L.cleora_propagate_dev(graph.handle, 0, a.data_ptr(), d, d, b.data_ptr(), d, _hip.F_L2NORM, 0.0, None, None, None, s))
        a, b = b, a
    torch.cuda.synchronize()
    return round((time.perf_counter() - t0) / iters * 1e3, 3)


run(4)
out = {"placement_ms": ms, "first": run(10), "policy_off": None, "after_replacing_the_copy": []}
graph.set_hot_cache(0)
run(2)
out["policy_off"] = run(10)
spacers = []
for k in range(8):
    graph.set_hot_cache(0)
    run(1)
    spacers.append(torch.empty((96 + 160 * k) << 20, dtype=torch.uint8, device=dev))      # moves the next allocation
    graph.set_hot_cache(256 << 20)
    run(3)
    out["after_replacing_the_copy"].append(run(10))
print(json.dumps(out))
