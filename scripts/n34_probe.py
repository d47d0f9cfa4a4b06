#!/usr/bin/env python3
"""Developer probe (round 2): the N3 / N4 kernels at BASELINE config 2's size (|V| = 1M, nnz = 20M, d = 256):
edge attention weights (cleora_edge_attention_dev) and the device top-k of predict_links / find_most_similar
(cleora_topk_cosine_dev), HIP-event times."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cleora_amd import _hip, synth
dev = torch.device("cuda:0")
L = _hip.lib()
g = synth.bipartite_graph(500_000, 500_000, 10_000_000, 1, dev)
n, nnz, d = g["n"], g["nnz"], 256
gr = _hip.Graph.from_device(n, n, nnz, g["rowptr"].data_ptr(), g["col"].data_ptr(), g["val_left"].data_ptr(), None, 0, keepalive=g)
x = torch.randn((n, d), device=dev)
s = torch.cuda.current_stream().cuda_stream
vals = torch.empty(nnz, device=dev)
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
t = timed(lambda: _hip.check(L.cleora_edge_attention_dev(gr.handle, 0, x.data_ptr(), d, d, 1.0, vals.data_ptr(), s)))
print(f"edge attention, n={n} nnz={nnz} d={d}: {t:.3f} ms  ({(nnz * d * 4 + n * d * 8) / t / 1e6:.0f} GB/s of gathered rows + norm pass)", flush=True)
t2 = timed(lambda: _hip.check(L.cleora_propagate_dev(gr.handle, 0, x.data_ptr(), d, d, torch.empty_like(x).data_ptr(), d, _hip.F_L2NORM, 0.0, None, None, None, s)))
print(f"   (the SpMM over the same edges: {t2:.3f} ms)", flush=True)
y = torch.empty_like(x)
t3 = timed(lambda: (_hip.check(L.cleora_edge_attention_dev(gr.handle, 0, x.data_ptr(), d, d, 1.0, vals.data_ptr(), s)),
                    _hip.check(L.cleora_propagate_vals_dev(gr.handle, vals.data_ptr(), x.data_ptr(), d, d, y.data_ptr(), d, _hip.F_L2NORM, 0.0, None, None, None, s))))
t4 = timed(lambda: _hip.check(L.cleora_propagate_attention_dev(gr.handle, 0, x.data_ptr(), d, d, 1.0, y.data_ptr(), d, _hip.F_L2NORM, 0.0, None, None, s)))
print(f"one attention iteration: weights + SpMM as two kernels {t3:.3f} ms; fused, softmax online (cleora_propagate_attention_dev) {t4:.3f} ms "
      f"= {t4 / t2:.2f} x the plain SpMM", flush=True)
for nq, k in ((1, 10), (8, 10), (9, 10), (64, 10), (256, 10), (64, 100)):
    q = torch.randint(0, n, (nq,), device=dev, dtype=torch.int32)
    oi = torch.empty((nq, k), dtype=torch.int32, device=dev); os_ = torch.empty((nq, k), device=dev)
    ws = torch.empty(L.cleora_topk_workspace(n, k), dtype=torch.uint8, device=dev)
    for excl in (0, 1):
        t = timed(lambda: _hip.check(L.cleora_topk_cosine_dev(gr.handle if excl else None, x.data_ptr(), d, n, d, q.data_ptr(), nq, k, 1, excl,
                                                              oi.data_ptr(), os_.data_ptr(), ws.data_ptr(), s)))
        print(f"top-{k} cosine for {nq} queries, exclude_existing={excl}: {t:.3f} ms  ({t / nq:.3f} ms per query; X = {n * d * 4 / 1e9:.2f} GB, read once per 8 queries on the vector units, per 64 on the matrix cores)", flush=True)
