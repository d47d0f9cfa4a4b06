"""Round 6: 300 launches of the fused SpMM at the C3 shape, each between its own pair of events on the launch stream (the span covers the
chain / hub kernels' join): how often does a launch run long because the side-stream kernels started late?
    python scripts/r06/step_histogram.py"""
import importlib.util, json, os, sys, types
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
from cleora_amd import _hip
args = types.SimpleNamespace(config="C3", nodes=0, pairs=0, hyperedges=0, products=0, dim=0, seed=2)
dev = torch.device("cuda:0")
g, hashes, _, cfg = bench.make_workload(args, dev, 0, 1, False)
n, nnz, d = g["n"], g["nnz"], cfg["dim"]
L = _hip.lib(); S = torch.cuda.current_stream().cuda_stream
gr = _hip.Graph.from_device(n, n, nnz, g["rowptr"].data_ptr(), g["col"].data_ptr(), g["val_left"].data_ptr(), None, 0, keepalive=(g["rowptr"], g["col"], g["val_left"]))
x = torch.empty((n, d), device=dev); y = torch.empty((n, d), device=dev)
_hip.check(L.cleora_init_dev(hashes.data_ptr(), n, d, 0, x.data_ptr(), d, S))
N = 60
evs = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
for _ in range(4):
    _hip.check(L.cleora_propagate_dev(gr.handle, _hip.LEFT, x.data_ptr(), d, d, y.data_ptr(), d, _hip.F_L2NORM, 0.0, None, None, None, S)); x, y = y, x
torch.cuda.synchronize()
evs[0].record()
for i in range(N):
    _hip.check(L.cleora_propagate_dev(gr.handle, _hip.LEFT, x.data_ptr(), d, d, y.data_ptr(), d, _hip.F_L2NORM, 0.0, None, None, None, S)); x, y = y, x
    evs[i + 1].record()
torch.cuda.synchronize()
ms = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(N))
med = ms[N // 2]
print(json.dumps({"launches": N, "median_ms": round(med, 3), "min": round(ms[0], 3), "p90": round(ms[int(N * 0.9)], 3), "p99": round(ms[int(N * 0.99)], 3), "max": round(ms[-1], 3),
                  "launches_more_than_3_ms_over_the_median": sum(1 for v in ms if v > med + 3.0), "mean": round(sum(ms) / N, 3)}))
