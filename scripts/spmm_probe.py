#!/usr/bin/env python3
"""Developer probe: time the fused SpMM+L2 kernel on synthetic graphs (GPU box only)."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cleora_amd import _hip, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graph", default="powerlaw", choices=["powerlaw", "bipartite"])
    ap.add_argument("--nodes", type=int, default=10_000_000)
    ap.add_argument("--pairs", type=int, default=95_000_000)
    ap.add_argument("--dim", type=int, default=256)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--hub", type=int, default=0)
    ap.add_argument("--seg", type=int, default=0)
    ap.add_argument("--flags", type=int, default=_hip.F_L2NORM)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    L = _hip.lib()
    t0 = time.time()
    if args.graph == "powerlaw":
        g = synth.power_law_graph(args.nodes, args.pairs, 2, dev)
    else:
        g = synth.bipartite_graph(args.nodes // 2, args.nodes // 2, args.pairs, 1, dev)
    torch.cuda.synchronize()
    n, nnz, d = g["n"], g["nnz"], args.dim
    deg = torch.diff(g["rowptr"])
    print(f"graph built in {time.time()-t0:.1f}s: n={n} nnz={nnz} maxdeg={int(deg.max())} "
          f"mean={float(deg.float().mean()):.1f}", flush=True)
    graph = _hip.Graph.from_device(n, n, nnz, g["rowptr"].data_ptr(), g["col"].data_ptr(),
                                   g["val_left"].data_ptr(), g["val_sym"].data_ptr(), 0,
                                   args.hub, args.seg, keepalive=g)
    info = graph.info()
    print(f"hub rows={info.n_hub_rows} segments={info.n_hub_segments} thr={info.hub_threshold}")
    hashes = synth.entity_hashes(n, 0, dev)
    x = torch.empty((n, d), dtype=torch.float32, device=dev)
    y = torch.empty_like(x)
    stream = torch.cuda.current_stream().cuda_stream
    _hip.check(L.cleora_init_dev(hashes.data_ptr(), n, d, 0, x.data_ptr(), d, stream))
    for flags in (args.flags, args.flags | _hip.F_FASTNORM, 0):
        for _ in range(2):
            _hip.check(L.cleora_propagate_dev(graph.handle, 0, x.data_ptr(), d, d, y.data_ptr(), d,
                                              flags, 0.0, None, None, None, stream))
            x, y = y, x
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            _hip.check(L.cleora_propagate_dev(graph.handle, 0, x.data_ptr(), d, d, y.data_ptr(), d,
                                              flags, 0.0, None, None, None, stream))
            x, y = y, x
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.iters
        bytes_iter = nnz * 8 + (n + 1) * 8 + nnz * d * 4 + n * d * 4
        print(json.dumps({"flags": flags, "ms_per_iter": round(ms, 3),
                          "gather_model_GBps": round(bytes_iter / ms / 1e6, 1),
                          "frac_8TBps": round(bytes_iter / ms / 1e6 / 8000, 3),
                          "edge_dim_per_s": nnz * d / ms * 1e3}), flush=True)
    # property: left Markov matrix is row-stochastic => A @ const = const
    ones = torch.full((n, d), 0.25, dtype=torch.float32, device=dev)
    _hip.check(L.cleora_propagate_dev(graph.handle, 0, ones.data_ptr(), d, d, y.data_ptr(), d, 0, 0.0,
                                      None, None, None, stream))
    torch.cuda.synchronize()
    print("row-stochastic check max|A*c - c| =", float((y - 0.25).abs().max()))


if __name__ == "__main__":
    main()
