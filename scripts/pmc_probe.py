#!/usr/bin/env python3
"""Workload for rocprofv3 passes: on the bench graph run init -> L2 rowops -> 3 x propagate.
init / rowops move a KNOWN number of bytes with the same 16-byte-per-lane row accesses as the
SpMM, which calibrates FETCH_SIZE / WRITE_SIZE (MI355X_MICROARCH.md §HBM) before they are
read for the SpMM kernel."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cleora_amd import _hip, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--nodes", type=int, default=10_000_000)
ap.add_argument("--pairs", type=int, default=95_000_000)
ap.add_argument("--dim", type=int, default=256)
ap.add_argument("--iters", type=int, default=7)   # the automatic gather cache policy arms on the third launch
ap.add_argument("--graph", default="c3", choices=["c3", "c2"], help="c3: power-law generator; c2: BASELINE config 2, bipartite 500k x 500k, 10M pairs")
ap.add_argument("--hot", type=int, default=-1, help="cleora_graph_set_hot_cache: -1 automatic, 0 off, >0 bytes")
args = ap.parse_args()
dev = torch.device("cuda:0")
L = _hip.lib()
g = synth.bipartite_graph(500_000, 500_000, 10_000_000, 1, dev) if args.graph == "c2" else synth.power_law_graph(args.nodes, args.pairs, 2, dev)
n, nnz, d = g["n"], g["nnz"], args.dim
graph = _hip.Graph.from_device(n, n, nnz, g["rowptr"].data_ptr(), g["col"].data_ptr(),
                               g["val_left"].data_ptr(), None, 0, 0, 0, keepalive=g)
graph.set_hot_cache(args.hot)
hashes = synth.entity_hashes(n, 0, dev)
x = torch.empty((n, d), dtype=torch.float32, device=dev)
y = torch.empty_like(x)
s = torch.cuda.current_stream().cuda_stream
_hip.check(L.cleora_init_dev(hashes.data_ptr(), n, d, 0, x.data_ptr(), d, s))
_hip.check(L.cleora_rowops_dev(x.data_ptr(), d, n, d, y.data_ptr(), d, _hip.F_L2NORM, 0.0, None, None, None, s))
x, y = y, x
for _ in range(args.iters):
    _hip.check(L.cleora_propagate_dev(graph.handle, 0, x.data_ptr(), d, d, y.data_ptr(), d,
                                      _hip.F_L2NORM, 0.0, None, None, None, s))
    x, y = y, x
torch.cuda.synchronize()
info = graph.info()
print(f"PMC_PROBE n={n} nnz={nnz} d={d} hub_rows={info.n_hub_rows} hub_segments={info.n_hub_segments}")
