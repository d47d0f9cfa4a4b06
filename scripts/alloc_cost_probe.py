#!/usr/bin/env python3
"""Developer probe (round 2): what the iterate-placement search costs.  (1) cleora_alloc_iterates wall clock for a pair and a
triple at BASELINE config 3; (2) raw hipMalloc / hipFree of one 10 GB iterate; (3) the plain cleora_embed_dev loop, 40
iterations, with CLEORA_TUNE_TRACE=1 (per-trial timings on stderr)."""
import ctypes, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cleora_amd import _hip, synth
dev = torch.device("cuda:0")
L = _hip.lib()
g = synth.power_law_graph(10_000_000, 95_000_000, 2, dev)
n, nnz, d = g["n"], g["nnz"], 256
gr = _hip.Graph.from_device(n, n, nnz, g["rowptr"].data_ptr(), g["col"].data_ptr(), g["val_left"].data_ptr(), None, 0, keepalive=g)
torch.cuda.synchronize()
for count in (2, 3, 2):
    t = time.perf_counter()
    bufs = _hip.DevArray.iterates(gr, n, d, count)
    print(f"cleora_alloc_iterates(count={count}): {(time.perf_counter() - t) * 1e3:.0f} ms", flush=True)
    del bufs
for _ in range(3):
    t = time.perf_counter(); a = _hip.DevArray((n, d), "float32"); t1 = time.perf_counter(); del a; t2 = time.perf_counter()
    print(f"hipMalloc 10 GB {1e3 * (t1 - t):.1f} ms, hipFree {1e3 * (t2 - t1):.1f} ms", flush=True)
x0 = torch.empty((n, d), device=dev)
_hip.check(L.cleora_init_dev(synth.entity_hashes(n, 0, dev).data_ptr(), n, d, 0, x0.data_ptr(), d, torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
for rep in range(2):
    x = x0.clone()
    t = time.perf_counter()
    _hip.check(L.cleora_embed_dev(gr.handle, x.data_ptr(), 0, d, 40, 0.0, 0.0, 0, None))
    print(f"plain cleora_embed_dev, 40 iterations: loop timer {L.cleora_last_embed_loop_ms() / 40:.2f} ms/iter, call wall {(time.perf_counter() - t) * 1e3 / 40:.2f} ms/iter", flush=True)
