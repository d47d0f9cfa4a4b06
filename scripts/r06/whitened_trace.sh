#!/bin/bash
# Round 6: who runs when in the default (whitened) loop — rocprofv3 --kernel-trace of a 6-iteration cleora_embed_dev + CLEORA_F_WHITEN call
# at a BASELINE shape (C5: d = 1024 through the host builder; C3 otherwise), condensed to (kernel, start, end) rows.
#   scripts/r06/whitened_trace.sh C5
cfg=${1:-C5}
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/gpurun_out/prof_r06
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
cat > /tmp/whitened_trace.py <<PY
import ctypes, sys, types, torch
sys.path.insert(0, "$root")
import importlib.util
spec = importlib.util.spec_from_file_location("bench", "$root/bench.py"); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
from cleora_amd import _hip
args = types.SimpleNamespace(config="$cfg", nodes=0, pairs=0, hyperedges=0, products=0, dim=0, seed=2)
dev = torch.device("cuda:0")
g, hashes, _, cfg = bench.make_workload(args, dev, 0, 1, False)
n, nnz, d = g["n"], g["nnz"], cfg["dim"]
L = _hip.lib()
gr = _hip.Graph.from_device(n, n, nnz, g["rowptr"].data_ptr(), g["col"].data_ptr(), g["val_left"].data_ptr(), None, 0, keepalive=(g["rowptr"], g["col"], g["val_left"]))
x = torch.empty((n, d), dtype=torch.float32, device=dev)
_hip.check(L.cleora_init_dev(hashes.data_ptr(), n, d, 0, x.data_ptr(), d, torch.cuda.current_stream().cuda_stream))
_hip.check(L.cleora_embed_dev(gr.handle, x.data_ptr(), _hip.LEFT, d, 2, 0.0, 0.0, _hip.F_WHITEN, None))
torch.cuda.synchronize()
_hip.check(L.cleora_init_dev(hashes.data_ptr(), n, d, 0, x.data_ptr(), d, torch.cuda.current_stream().cuda_stream))
_hip.check(L.cleora_embed_dev(gr.handle, x.data_ptr(), _hip.LEFT, d, 6, 0.0, 0.0, _hip.F_WHITEN, None))
torch.cuda.synchronize()
print("loop ms", L.cleora_last_embed_loop_ms())
PY
timeout 900 rocprofv3 --kernel-trace --output-format csv -d "$out/wtrace" -o wt -- python /tmp/whitened_trace.py > "$out/wtrace.log" 2>&1
tail -3 "$out/wtrace.log" | cut -c1-300
t=$(find "$out/wtrace" -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python "$root/scripts/r05/trace_short.py" "$t" "$out/whitened_loop_trace_$cfg.csv" && tail -120 "$out/whitened_loop_trace_$cfg.csv"
find "$out/wtrace" -type f -delete
