#!/usr/bin/env python3
"""Round-3 measurement probe (GPU box): kernel timings of the whitening pieces and of the loops under the library's A/B
switches (CLEORA_PROJECT=f32, CLEORA_GRAM=f64, CLEORA_SPMM_WAITS=compiler are read once per process, so every variant is its
own process).  Prints one JSON line per mode.

    python scripts/r03_probe.py kernels [n] [d]     projection / Gram / stats timings on random unit rows
    python scripts/r03_probe.py loop [nodes] [pairs] [d]   plain SpMM launch ms + whitened loop ms/iter on the C3 graph
"""
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cleora_amd import _hip, synth  # noqa: E402

ENV = {k: os.environ.get(k) for k in ("CLEORA_PROJECT", "CLEORA_GRAM", "CLEORA_SPMM_WAITS", "CLEORA_CHOLESKY") if os.environ.get(k)}


def timed(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def kernels(n, d):
    dev = torch.device("cuda:0")
    L = _hip.lib()
    s = torch.cuda.current_stream().cuda_stream
    x = torch.randn((n, d), device=dev)
    x = (x * torch.linspace(0.5, 2.0, d, device=dev) + 0.1)
    x = (x / x.norm(dim=1, keepdim=True)).contiguous()
    mean = x.mean(0).contiguous()
    t = (torch.randn((d, d), device=dev) * d ** 0.5).contiguous()
    out = torch.empty_like(x)
    res = {"mode": "kernels", "n": n, "d": d, "env": ENV}
    res["project_ms"] = timed(lambda: _hip.check(L.cleora_project_dev(x.data_ptr(), d, n, d, mean.data_ptr(), t.data_ptr(), d,
                                                                      out.data_ptr(), d, s)))
    res["project_tflops_f32_equiv"] = 2.0 * n * d * d / (res["project_ms"] * 1e-3) / 1e12
    res["project_gbps_x_plus_out"] = 2.0 * n * d * 4 / (res["project_ms"] * 1e-3) / 1e9
    # accuracy on a row sample against f64
    rows = torch.arange(0, n, max(1, n // 4096), device=dev)[:4096]
    ref = (x[rows].double() - mean.double()) @ t.double()
    res["project_max_err_rel"] = float((out[rows].double() - ref).abs().max() / ref.abs().max())
    ws = torch.empty(L.cleora_whiten_workspace(n, d), dtype=torch.uint8, device=dev)
    m64 = torch.empty(d, dtype=torch.float64, device=dev)
    g64 = torch.empty((d, d), dtype=torch.float64, device=dev)
    grams = {}
    for inter in (0, 1):
        res[f"stats_ms_intermediate{inter}"] = timed(lambda: _hip.check(L.cleora_whiten_stats_dev(
            x.data_ptr(), d, n, d, ws.data_ptr(), inter, m64.data_ptr(), g64.data_ptr(), s)))
        grams[inter] = g64.clone()
    res["gram_f32_vs_f64_rel_fro"] = float((grams[1] - grams[0]).norm() / grams[0].norm())
    sub = x[: min(n, 2_000_000)].double()
    # reference for the mean only (the f64 Gram is pinned by the tests)
    res["mean_err"] = float((m64 - x.double().mean(0)).abs().max())
    del sub
    print(json.dumps(res), flush=True)


def loop(nodes, pairs, d):
    dev = torch.device("cuda:0")
    L = _hip.lib()
    g = synth.power_law_graph(nodes, pairs, 2, dev) if nodes > 2_000_000 else synth.bipartite_graph(nodes // 2, nodes // 2, pairs, 1, dev)
    n, nnz = g["n"], g["nnz"]
    gr = _hip.Graph.from_device(n, n, nnz, g["rowptr"].data_ptr(), g["col"].data_ptr(), g["val_left"].data_ptr(), None, 0,
                                keepalive=g)
    hashes = synth.entity_hashes(n, 0, dev)
    s = torch.cuda.current_stream().cuda_stream
    (a_, b_), place = _hip.DevArray.iterates(gr, n, d, 2)
    a, b = torch.as_tensor(a_, device=dev), torch.as_tensor(b_, device=dev)
    _hip.check(L.cleora_init_dev(hashes.data_ptr(), n, d, 0, a.data_ptr(), d, s))
    res = {"mode": "loop", "n": n, "nnz": nnz, "d": d, "env": ENV, "placement_ms": [round(place[0], 3), round(place[1], 3)]}

    def it():
        nonlocal a, b
        _hip.check(L.cleora_propagate_dev(gr.handle, _hip.LEFT, a.data_ptr(), d, d, b.data_ptr(), d, _hip.F_L2NORM, 0.0, None, None, None, s))
        a, b = b, a
    for _ in range(4):
        it()
    torch.cuda.synchronize()
    gr.set_timing(True)
    t0 = time.perf_counter()
    for _ in range(20):
        it()
    torch.cuda.synchronize()
    res["plain_ms_per_iter_wall"] = (time.perf_counter() - t0) / 20 * 1e3
    ms, c = gr.get_timing()
    gr.set_timing(False)
    res["spmm_rows_kernel_ms"] = ms[1] / max(c, 1)
    res["hot_rows"] = int(gr.info().hot_rows)
    x0 = a.clone()
    del a, b, a_, b_
    torch.cuda.empty_cache()
    _hip.check(L.cleora_embed_dev(gr.handle, x0.data_ptr(), _hip.LEFT, d, 2, 0.0, 0.0, _hip.F_WHITEN, None))
    iters = 8
    _hip.check(L.cleora_embed_dev(gr.handle, x0.data_ptr(), _hip.LEFT, d, iters, 0.0, 0.0, _hip.F_WHITEN, None))
    res["whitened_ms_per_iter"] = L.cleora_last_embed_loop_ms() / iters
    cov = torch.cov(x0[: min(n, 1_000_000)].double().T)
    res["whitened_cov_minus_identity"] = float((cov - torch.eye(d, device=dev, dtype=torch.float64)).abs().max())
    print(json.dumps(res), flush=True)


def loop_masks(nodes, pairs, d):
    """The whitened loop under the CU-partition switches of embed_whitened_overlapped (read per call): one graph, one process."""
    dev = torch.device("cuda:0")
    L = _hip.lib()
    g = synth.power_law_graph(nodes, pairs, 2, dev)
    n, nnz = g["n"], g["nnz"]
    gr = _hip.Graph.from_device(n, n, nnz, g["rowptr"].data_ptr(), g["col"].data_ptr(), g["val_left"].data_ptr(), None, 0, keepalive=g)
    hashes = synth.entity_hashes(n, 0, dev)
    x0 = torch.empty((n, d), device=dev)
    s = torch.cuda.current_stream().cuda_stream
    _hip.check(L.cleora_init_dev(hashes.data_ptr(), n, d, 0, x0.data_ptr(), d, s))
    torch.cuda.synchronize()
    init = x0.clone()
    variants = [{}, {"CLEORA_GRAM_CUS": "32"}, {"CLEORA_GRAM_CUS": "64"}, {"CLEORA_GRAM_CUS": "64", "CLEORA_GRAM_CO_BLOCKS": "2"},
                {"CLEORA_GRAM_CUS": "96", "CLEORA_GRAM_CO_BLOCKS": "2"}, {"CLEORA_GRAM_CUS": "64", "CLEORA_SPMM_AVOID": "1"},
                {"CLEORA_GRAM_CUS": "64", "CLEORA_GRAM_CO_BLOCKS": "2", "CLEORA_SPMM_AVOID": "1"},
                {"CLEORA_GRAM_CUS": "48", "CLEORA_GRAM_CO_BLOCKS": "2", "CLEORA_SPMM_AVOID": "1"},
                {"CLEORA_GRAM_CUS": "128", "CLEORA_GRAM_CO_BLOCKS": "2"}, {}]
    keys = ("CLEORA_GRAM_CUS", "CLEORA_GRAM_CO_BLOCKS", "CLEORA_SPMM_AVOID")
    _hip.check(L.cleora_embed_dev(gr.handle, x0.data_ptr(), _hip.LEFT, d, 2, 0.0, 0.0, _hip.F_WHITEN, None))
    for v in variants:
        for k in keys:
            os.environ.pop(k, None)
        os.environ.update({k: val for k, val in v.items() if k in keys})
        x0.copy_(init)
        torch.cuda.synchronize()
        iters = 8
        rc = L.cleora_embed_dev(gr.handle, x0.data_ptr(), _hip.LEFT, d, iters, 0.0, 0.0, _hip.F_WHITEN, None)
        res = {"mode": "loop_masks", "n": n, "nnz": nnz, "d": d, "switches": v, "rc": rc}
        if rc == 0:
            res["whitened_ms_per_iter"] = L.cleora_last_embed_loop_ms() / iters
            cov = torch.cov(x0[:1_000_000].double().T)
            res["whitened_cov_minus_identity"] = float((cov - torch.eye(d, device=dev, dtype=torch.float64)).abs().max())
        else:
            res["error"] = _hip.last_error() if hasattr(_hip, "last_error") else "?"
        print(json.dumps(res), flush=True)


def project_only(n, d):
    """The projection alone (profiling: CLEORA_PROJECT_DEBUG variants, rocprofv3 --pmc passes)."""
    dev = torch.device("cuda:0")
    L = _hip.lib()
    s = torch.cuda.current_stream().cuda_stream
    x = torch.randn((n, d), device=dev)
    x = (x / x.norm(dim=1, keepdim=True)).contiguous()
    mean = x.mean(0).contiguous()
    t = (torch.randn((d, d), device=dev) * d ** 0.5).contiguous()
    out = torch.empty_like(x)
    ms = timed(lambda: _hip.check(L.cleora_project_dev(x.data_ptr(), d, n, d, mean.data_ptr(), t.data_ptr(), d, out.data_ptr(), d, s)),
               reps=4, warm=1)
    print(json.dumps({"mode": "project", "n": n, "d": d, "env": ENV, "debug": os.environ.get("CLEORA_PROJECT_DEBUG"), "project_ms": ms,
                      "f32_equiv_tflops": 2.0 * n * d * d / (ms * 1e-3) / 1e12}), flush=True)


if __name__ == "__main__":
    mode = sys.argv[1]
    args = [int(v) for v in sys.argv[2:]]
    if mode == "loop_masks":
        loop_masks(*(args or [10_000_000, 95_000_000, 256]))
    elif mode == "project":
        project_only(*(args + [10_000_000, 256][len(args):]))
    elif mode == "kernels":
        kernels(*(args + [10_000_000, 256][len(args):]))
    else:
        loop(*(args + [10_000_000, 95_000_000, 256][len(args):]))
