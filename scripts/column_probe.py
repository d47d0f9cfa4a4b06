#!/usr/bin/env python3
"""Developer probe: what ONE rank of a P-way column-partitioned run executes per iteration
(SpMM + ROWSQ on d/P columns, then the SCALE pass), timed on a single GPU; the all-reduce of the
n-float row sums is the only part not exercised here."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cleora_amd import _hip, sharded, synth
dev = torch.device("cuda:0")
g = synth.power_law_graph(10_000_000, 95_000_000, 2, dev)
n, nnz, d = g["n"], g["nnz"], 256
be = sharded.HipBackend(dev)
hashes = synth.entity_hashes(n, 0, dev)
L = _hip.lib()
for P in (1, 2, 4, 8):
    cg = sharded.ColumnShardedGraph(n, g["rowptr"], g["col"], g["val_left"], None, d, 0, P, be)
    dl = cg.dl
    x = torch.empty((n, dl), dtype=torch.float32, device=dev)
    y = torch.empty_like(x)
    rowsq = torch.zeros(n, dtype=torch.float32, device=dev)
    _hip.check(L.cleora_init_dev(hashes.data_ptr(), n, dl, 0, x.data_ptr(), dl, torch.cuda.current_stream().cuda_stream))
    def step():
        global x, y
        if P == 1:
            be.propagate(cg.block, 0, x, y, _hip.F_L2NORM, 0.0, x)
        else:
            be.propagate(cg.block, 0, x, y, _hip.F_ROWSQ, 0.0, x, None, rowsq)
            rowsq.mul_(float(P))  # stand-in for the all-reduce (keeps the norms sane)
            be.rowops(y, y, _hip.F_SCALE, 0.0, None, None, rowsq)
        x, y = y, x
    for hot in (0, -1, 128 << 20, 256 << 20, 512 << 20):      # gather cache policy: off, automatic, forced budgets
        cg.block.set_hot_cache(hot)
        for _ in range(3): step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): step()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(json.dumps({"emulated_world": P, "cols_per_rank": dl, "hot_bytes": hot,
                          "ms_per_iter_without_allreduce": round(ms, 3)}), flush=True)
    del cg, x, y
