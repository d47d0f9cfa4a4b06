#!/usr/bin/env python3
"""Developer probe (round 2): ms per iteration of the whitened default loop through cleora_embed_dev on BASELINE config 3 (or 2
with --c2): overlapped (SpMM(t+1) beside Gram/eigh(t)) against the reference's sequential order, from the library's
own loop timer.  Environment knobs read by the library: CLEORA_GRAM_CO_BLOCKS (Gram blocks per CU while co-running)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cleora_amd import _hip, synth
dev = torch.device("cuda:0")
L = _hip.lib()
c2 = "--c2" in sys.argv
g = synth.bipartite_graph(500_000, 500_000, 10_000_000, 1, dev) if c2 else synth.power_law_graph(10_000_000, 95_000_000, 2, dev)
n, nnz, d = g["n"], g["nnz"], 256
gr = _hip.Graph.from_device(n, n, nnz, g["rowptr"].data_ptr(), g["col"].data_ptr(), g["val_left"].data_ptr(), None, 0, keepalive=g)
x0 = torch.empty((n, d), device=dev)
_hip.check(L.cleora_init_dev(synth.entity_hashes(n, 0, dev).data_ptr(), n, d, 0, x0.data_ptr(), d, torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
iters = 10
# the plain loop (embed_fast: SpMM + fused L2, the partner buffer searched on its own iterations) for comparison with bench.py's step
x = x0.clone()
_hip.check(L.cleora_embed_dev(gr.handle, x.data_ptr(), 0, d, 40, 0.0, 0.0, 0, None))
print(f"{'C2' if c2 else 'C3'} plain embed loop, 40 iterations incl. the placement search on the way: {L.cleora_last_embed_loop_ms() / 40:.2f} ms/iter", flush=True)
for label, thr in (("overlapped", 0.0), ("sequential+rmse", 1e-30), ("overlapped", 0.0)):
    x = x0.clone()
    _hip.check(L.cleora_embed_dev(gr.handle, x.data_ptr(), 0, d, iters, 0.0, thr, _hip.F_WHITEN, None))
    print(f"{'C2' if c2 else 'C3'} {label}: {L.cleora_last_embed_loop_ms() / iters:.2f} ms/iter  (Cholesky route: {os.environ.get('CLEORA_CHOLESKY', 'default')})", flush=True)
