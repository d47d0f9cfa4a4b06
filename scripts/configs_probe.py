#!/usr/bin/env python3
"""Developer probe: per-iteration times for BASELINE configs 2 and 5 (1 GPU), device resident."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cleora_amd import _hip, synth, sharded
dev = torch.device("cuda:0")
L = _hip.lib()
def run(name, g, d, whiten, iters):
    n, nnz = g["n"], g["nnz"]
    sg = sharded.ShardedGraph(n, g["rowptr"], g["col"], g["val_left"], None, 0, 1, 1, sharded.HipBackend(dev))
    x0 = torch.zeros((sg.n_pad, d), dtype=torch.float32, device=dev)
    hashes = synth.entity_hashes(n, 0, dev)
    _hip.check(L.cleora_init_dev(hashes.data_ptr(), n, d, 0, x0.data_ptr(), d, torch.cuda.current_stream().cuda_stream))
    sharded.embed_sharded(sg, 0, x0, 2, whiten=whiten); torch.cuda.synchronize()
    t0 = time.perf_counter()
    sharded.embed_sharded(sg, 0, x0, iters, whiten=whiten); torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / iters
    b = nnz * 8 + (n + 1) * 8 + nnz * d * 4 + n * d * 4
    print(f"{name}: n={n} nnz={nnz} d={d} whiten={whiten}: {ms:.2f} ms/iter, {1e3/ms:.1f} it/s, "
          f"{nnz*d/ms/1e6:.0f} G edge*dim/s, gather-model {b/ms/1e6:.0f} GB/s", flush=True)
g2 = synth.bipartite_graph(500_000, 500_000, 10_000_000, 1, dev)
run("C2 bipartite", g2, 256, False, 40)
run("C2 bipartite", g2, 256, True, 10)
del g2; torch.cuda.empty_cache()
# C5 flavour: hyperedges of arity ~2+Poisson(6) over 2M products expanded to cliques -> use the power-law generator with
# the same nnz budget (320M) as a structural stand-in for the clique expansion
g5 = synth.power_law_graph(2_000_000, 159_000_000, 5, dev)
run("C5 stand-in", g5, 1024, False, 5)
run("C5 stand-in", g5, 1024, True, 3)
