"""Round 6: the placement classes of the fused SpMM against the cache policy of its Y stores — several builds of the library (default /
nt / sc1 / sc0 sc1 / sc0 / sc1 nt stores: -DCLEORA_Y_POLICY=k, built by store_policy_probe.sh) loaded into ONE process and timed on the
SAME (X, Y) pairs, K candidate Y buffers each.
    python scripts/r06/store_policy_probe.py lib_default.so lib_p1.so ..."""
import ctypes, importlib.util, json, os, sys, types
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
from cleora_amd import _hip
args = types.SimpleNamespace(config="C3", nodes=0, pairs=0, hyperedges=0, products=0, dim=0, seed=2)
dev = torch.device("cuda:0")
g, hashes, _, cfg = bench.make_workload(args, dev, 0, 1, False)
n, nnz, d = g["n"], g["nnz"], cfg["dim"]
S = torch.cuda.current_stream().cuda_stream
libs = {"default": _hip.lib()}
for path in sys.argv[1:]:
    L = ctypes.CDLL(os.path.abspath(path))
    for name, (res, a) in _hip.SIGNATURES.items():
        fn = getattr(L, name); fn.restype = res; fn.argtypes = a
    libs[os.path.basename(path).replace("libcleora_hip_", "").replace(".so", "")] = L
graphs = {}
for name, L in libs.items():
    h = ctypes.c_void_p()
    rc = L.cleora_graph_create_dev(0, n, n, nnz, g["rowptr"].data_ptr(), g["col"].data_ptr(), g["val_left"].data_ptr(), None, 0, 0, ctypes.byref(h))
    assert rc == 0, (name, rc)
    graphs[name] = h
x = torch.empty((n, d), device=dev)
_hip.check(libs["default"].cleora_init_dev(hashes.data_ptr(), n, d, 0, x.data_ptr(), d, S))
x /= x.norm(dim=1, keepdim=True)
K = 5
ys = [torch.empty((n, d), device=dev) for _ in range(K)]
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
res = {"n": n, "nnz": nnz, "d": d, "x": hex(x.data_ptr()), "pairs": []}
def run(L, h, src, dst, reps):
    for _ in range(reps):
        rc = L.cleora_propagate_dev(h, _hip.LEFT, src.data_ptr(), d, d, dst.data_ptr(), d, _hip.F_L2NORM, 0.0, None, None, None, S)
        assert rc == 0
for k, y in enumerate(ys):
    row = {"y": hex(y.data_ptr())}
    for name, L in libs.items():
        run(L, graphs[name], x, y, 4)                 # (the gather cache policy arms on the third launch)
        torch.cuda.synchronize()
        ev[0].record(); run(L, graphs[name], x, y, 6); ev[1].record(); torch.cuda.synchronize()
        row[name] = round(ev[0].elapsed_time(ev[1]) / 6, 3)
    # and the default build once more at the end (drift within the pair?)
    L = libs["default"]; ev[0].record(); run(L, graphs["default"], x, y, 6); ev[1].record(); torch.cuda.synchronize()
    row["default_again"] = round(ev[0].elapsed_time(ev[1]) / 6, 3)
    res["pairs"].append(row)
    print(json.dumps(row), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "r06_store_policy_probe.json"), "w"), indent=1)
