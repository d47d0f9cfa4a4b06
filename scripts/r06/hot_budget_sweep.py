"""Round 6: the gather cache policy's hot-set budget again, now that the SpMM's own stores no longer pass through the caches (C3, one pair of
buffers, one process): ms per launch against cleora_graph_set_hot_cache(bytes).
    python scripts/r06/hot_budget_sweep.py"""
import importlib.util, json, os, sys, types
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
from cleora_amd import _hip
args = types.SimpleNamespace(config=sys.argv[1] if len(sys.argv) > 1 else "C3", nodes=0, pairs=0, hyperedges=0, products=0, dim=0, seed=2)
dev = torch.device("cuda:0")
g, hashes, _, cfg = bench.make_workload(args, dev, 0, 1, False)
n, nnz, d = g["n"], g["nnz"], cfg["dim"]
L = _hip.lib(); S = torch.cuda.current_stream().cuda_stream
gr = _hip.Graph.from_device(n, n, nnz, g["rowptr"].data_ptr(), g["col"].data_ptr(), g["val_left"].data_ptr(), None, 0, keepalive=(g["rowptr"], g["col"], g["val_left"]))
x = torch.empty((n, d), device=dev)
ys = [torch.empty((n, d), device=dev) for _ in range(4)]
y = ys[0]
_hip.check(L.cleora_init_dev(hashes.data_ptr(), n, d, 0, x.data_ptr(), d, S))
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
def run(reps):
    for _ in range(reps):
        _hip.check(L.cleora_propagate_dev(gr.handle, _hip.LEFT, x.data_ptr(), d, d, y.data_ptr(), d, _hip.F_L2NORM, 0.0, None, None, None, S))
res = {"config": args.config, "n": n, "d": d, "pairs": []}
MB = 1 << 20
for k, yk in enumerate(ys):                       # several destination buffers: both placement classes, with luck
    y = yk
    row = {"y": hex(y.data_ptr()), "ms_per_launch": {}}
    for rep in range(3):
        for budget in (0, 128 * MB, 192 * MB, 256 * MB, 384 * MB, 512 * MB, 768 * MB, 1536 * MB):
            if budget == 0 and rep:
                continue
            gr.set_hot_cache(budget)
            run(4); torch.cuda.synchronize()
            ev[0].record(); run(8); ev[1].record(); torch.cuda.synchronize()
            row["ms_per_launch"].setdefault(f"{budget // MB} MiB", []).append(round(ev[0].elapsed_time(ev[1]) / 8, 3))
    res["pairs"].append(row)
    print(json.dumps(row), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", f"r06_hot_budget_sweep_{args.config.lower()}.json"), "w"), indent=1)
