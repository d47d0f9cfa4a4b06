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
    rel = (grams[1] - grams[0]) / grams[0]
    dg = torch.diagonal(rel)
    res["gram_diag_rel_err_mean"] = float(dg.mean())
    res["gram_diag_rel_err_std"] = float(dg.std())
    big = grams[0].abs() > 0.2 * torch.diagonal(grams[0]).abs().mean()      # entries with a covariance that is not noise
    big &= ~torch.eye(d, dtype=torch.bool, device=dev)
    if int(big.sum()):
        res["gram_offdiag_big_entries"] = int(big.sum())
        res["gram_offdiag_big_rel_err_mean"] = float(rel[big].mean())
        res["gram_offdiag_big_rel_err_std"] = float(rel[big].std())
    res["gram_offdiag_abs_err_rms_over_diag_mean"] = float(((grams[1] - grams[0])[~torch.eye(d, dtype=torch.bool, device=dev)]).pow(2).mean().sqrt() / torch.diagonal(grams[0]).mean())
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
    g = synth.power_law_graph(nodes, pairs, 2, dev) if nodes > 2_000_000 else synth.bipartite_graph(nodes // 2, nodes // 2, pairs, 1, dev)
    n, nnz = g["n"], g["nnz"]
    gr = _hip.Graph.from_device(n, n, nnz, g["rowptr"].data_ptr(), g["col"].data_ptr(), g["val_left"].data_ptr(), None, 0, keepalive=g)
    hashes = synth.entity_hashes(n, 0, dev)
    x0 = torch.empty((n, d), device=dev)
    s = torch.cuda.current_stream().cuda_stream
    _hip.check(L.cleora_init_dev(hashes.data_ptr(), n, d, 0, x0.data_ptr(), d, s))
    torch.cuda.synchronize()
    init = x0.clone()
    variants = [{}, {"CLEORA_STATS_BEFORE_SPMM": "1"}, {"CLEORA_STATS_BEFORE_SPMM": "1", "CLEORA_GRAM_CO_BLOCKS": "2"}, {},
                {"CLEORA_STATS_BEFORE_SPMM": "1"}, {"CLEORA_GRAM": "f32"}, {"CLEORA_GRAM": "f32", "CLEORA_STATS_BEFORE_SPMM": "1"}, {}]
    keys = ("CLEORA_GRAM_CUS", "CLEORA_GRAM_CO_BLOCKS", "CLEORA_SPMM_AVOID", "CLEORA_STATS_BEFORE_SPMM", "CLEORA_GRAM")
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
            cov = torch.cov(x0.double().T) if n <= 2_000_000 else torch.cov(x0[:1_000_000].double().T)
            res["whitened_cov_minus_identity"] = float((cov - torch.eye(d, device=dev, dtype=torch.float64)).abs().max())
        else:
            res["error"] = _hip.last_error() if hasattr(_hip, "last_error") else "?"
        print(json.dumps(res), flush=True)


def spmm_masks(nodes, pairs, d):
    """The plain SpMM on streams confined to a part of the chip: does its time follow the CUs it holds?"""
    dev = torch.device("cuda:0")
    L = _hip.lib()
    g = synth.power_law_graph(nodes, pairs, 2, dev)
    n, nnz = g["n"], g["nnz"]
    gr = _hip.Graph.from_device(n, n, nnz, g["rowptr"].data_ptr(), g["col"].data_ptr(), g["val_left"].data_ptr(), None, 0, keepalive=g)
    hashes = synth.entity_hashes(n, 0, dev)
    (a_, b_), place = _hip.DevArray.iterates(gr, n, d, 2)
    a, b = torch.as_tensor(a_, device=dev), torch.as_tensor(b_, device=dev)
    s0 = torch.cuda.current_stream().cuda_stream
    _hip.check(L.cleora_init_dev(hashes.data_ptr(), n, d, 0, a.data_ptr(), d, s0))
    torch.cuda.synchronize()
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    patterns = {"all": lambda c: True, "first 192": lambda c: c < 192, "first 128": lambda c: c < 128, "first 64": lambda c: c < 64,
                "3 of every 4": lambda c: c % 4 != 3, "every other": lambda c: c % 2 == 0, "last 192": lambda c: c >= 64,
                "3 of every 4 groups of 8": lambda c: (c // 8) % 4 != 3, "3 of every 4 groups of 32": lambda c: (c // 32) % 4 != 3}
    order = list(patterns.items()) + [(k, patterns[k]) for k in ("all", "3 of every 4", "every other") * 3]
    for name, f in order:
        words = (cus + 31) // 32
        mask = (ctypes.c_uint32 * words)()
        on = 0
        for c in range(cus):
            if f(c):
                mask[c >> 5] |= 1 << (c & 31)
                on += 1
        st = ctypes.c_void_p()
        _hip.check(L.cleora_stream_create_cu_mask(ctypes.byref(st), mask, words))

        def it():
            nonlocal a, b
            _hip.check(L.cleora_propagate_dev(gr.handle, _hip.LEFT, a.data_ptr(), d, d, b.data_ptr(), d, _hip.F_L2NORM, 0.0, None, None, None, st))
            a, b = b, a
        for _ in range(2):
            it()
        _hip.check(L.cleora_stream_sync(st))
        t0 = time.perf_counter()
        for _ in range(6):
            it()
        _hip.check(L.cleora_stream_sync(st))
        ms = (time.perf_counter() - t0) / 6 * 1e3
        _hip.check(L.cleora_stream_destroy(st))
        print(json.dumps({"mode": "spmm_masks", "n": n, "nnz": nnz, "d": d, "mask": name, "cus": on, "spmm_ms": ms,
                          "ms_times_cus_over_256": ms * on / 256}), flush=True)


def overlap(nodes, pairs, d):
    """Do the SpMM and the statistics kernels overlap at all?  Each alone and both at once, on plain streams and on streams that
    own interleaved parts of the chip."""
    dev = torch.device("cuda:0")
    L = _hip.lib()
    g = synth.power_law_graph(nodes, pairs, 2, dev)
    n, nnz = g["n"], g["nnz"]
    gr = _hip.Graph.from_device(n, n, nnz, g["rowptr"].data_ptr(), g["col"].data_ptr(), g["val_left"].data_ptr(), None, 0, keepalive=g)
    hashes = synth.entity_hashes(n, 0, dev)
    (a_, b_), place = _hip.DevArray.iterates(gr, n, d, 2)
    a, b = torch.as_tensor(a_, device=dev), torch.as_tensor(b_, device=dev)
    s0 = torch.cuda.current_stream().cuda_stream
    _hip.check(L.cleora_init_dev(hashes.data_ptr(), n, d, 0, a.data_ptr(), d, s0))
    for _ in range(3):
        _hip.check(L.cleora_propagate_dev(gr.handle, _hip.LEFT, a.data_ptr(), d, d, b.data_ptr(), d, _hip.F_L2NORM, 0.0, None, None, None, s0))
        a, b = b, a
    torch.cuda.synchronize()
    ws = torch.empty(L.cleora_whiten_workspace(n, d), dtype=torch.uint8, device=dev)
    m64 = torch.empty(d, dtype=torch.float64, device=dev)
    g64 = torch.empty((d, d), dtype=torch.float64, device=dev)
    mean = torch.zeros(d, device=dev)
    tr = torch.eye(d, device=dev).contiguous()
    out = torch.empty_like(a)
    cus = torch.cuda.get_device_properties(dev).multi_processor_count

    def masked(f):
        if f is None:
            st = ctypes.c_void_p()
            _hip.check(L.cleora_stream_create(ctypes.byref(st)))
            return st
        words = (cus + 31) // 32
        mask = (ctypes.c_uint32 * words)()
        for c in range(cus):
            if f(c):
                mask[c >> 5] |= 1 << (c & 31)
        st = ctypes.c_void_p()
        _hip.check(L.cleora_stream_create_cu_mask(ctypes.byref(st), mask, words))
        return st

    def wall(fns, streams, reps=4):
        for f in fns:
            f()
        for st in streams:
            _hip.check(L.cleora_stream_sync(st))
        t0 = time.perf_counter()
        for _ in range(reps):
            for f in fns:
                f()
        for st in streams:
            _hip.check(L.cleora_stream_sync(st))
        return (time.perf_counter() - t0) / reps * 1e3

    layouts = {"plain streams": (None, None), "SpMM 3 of 4, other every 4th": (lambda c: c % 4 != 3, lambda c: c % 4 == 3),
               "SpMM even CUs, other odd CUs": (lambda c: c % 2 == 0, lambda c: c % 2 == 1)}
    for name, (fa, fb) in layouts.items():
        sa, sb = masked(fa), masked(fb)
        spmm = lambda: _hip.check(L.cleora_propagate_dev(gr.handle, _hip.LEFT, a.data_ptr(), d, d, b.data_ptr(), d, _hip.F_L2NORM, 0.0, None, None, None, sa))
        stats1 = lambda: _hip.check(L.cleora_whiten_stats_dev(a.data_ptr(), d, n, d, ws.data_ptr(), 1, m64.data_ptr(), g64.data_ptr(), sb))
        stats0 = lambda: _hip.check(L.cleora_whiten_stats_dev(a.data_ptr(), d, n, d, ws.data_ptr(), 0, m64.data_ptr(), g64.data_ptr(), sb))
        proj = lambda: _hip.check(L.cleora_project_dev(a.data_ptr(), d, n, d, mean.data_ptr(), tr.data_ptr(), d, out.data_ptr(), d, sb))
        copy = lambda: _hip.check(L.cleora_memcpy_d2d(out.data_ptr(), a.data_ptr(), n * d * 4, sb))
        res = {"mode": "overlap", "layout": name, "spmm_alone_ms": wall([spmm], [sa])}
        for tag, other in (("stats_f32", stats1), ("stats_f64", stats0), ("project", proj), ("copy_10GB", copy)):
            res[tag + "_alone_ms"] = wall([other], [sb])
            res[tag + "_then_spmm_both_ms"] = wall([other, spmm], [sa, sb])
            res["spmm_then_" + tag + "_both_ms"] = wall([spmm, other], [sa, sb])
        print(json.dumps(res), flush=True)
        _hip.check(L.cleora_stream_destroy(sa))
        _hip.check(L.cleora_stream_destroy(sb))


def side_load(nodes, pairs, d):
    """What a co-resident kernel costs the SpMM: synthetic side kernels (scripts/probes/side_load.hip) on a second stream —
    wave slots only (spin), the matrix cores from registers (no memory traffic), a streaming read — each sized to ~25 ms alone."""
    dev = torch.device("cuda:0")
    L = _hip.lib()
    side = ctypes.CDLL(os.path.join(ROOT, "scripts", "probes", "libside_load.so"))
    vp, u32, u64 = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64
    side.side_spin.argtypes = [u64, u32, u32, vp, vp]
    side.side_mfma.argtypes = [u32, u32, u32, vp, vp]
    side.side_mfma32.argtypes = [u32, u32, u32, vp, vp]
    side.side_read.argtypes = [vp, u64, u32, u32, vp, vp]
    g = synth.power_law_graph(nodes, pairs, 2, dev)
    n, nnz = g["n"], g["nnz"]
    gr = _hip.Graph.from_device(n, n, nnz, g["rowptr"].data_ptr(), g["col"].data_ptr(), g["val_left"].data_ptr(), None, 0, keepalive=g)
    hashes = synth.entity_hashes(n, 0, dev)
    (a_, b_), place = _hip.DevArray.iterates(gr, n, d, 2)
    a, b = torch.as_tensor(a_, device=dev), torch.as_tensor(b_, device=dev)
    s0 = torch.cuda.current_stream().cuda_stream
    _hip.check(L.cleora_init_dev(hashes.data_ptr(), n, d, 0, a.data_ptr(), d, s0))
    for _ in range(3):
        _hip.check(L.cleora_propagate_dev(gr.handle, _hip.LEFT, a.data_ptr(), d, d, b.data_ptr(), d, _hip.F_L2NORM, 0.0, None, None, None, s0))
        a, b = b, a
    torch.cuda.synchronize()
    sink = torch.zeros(64, device=dev)
    other = torch.randn((n, d), device=dev)          # what the read kernel streams: not the SpMM's operands
    sa, sb = ctypes.c_void_p(), ctypes.c_void_p()
    _hip.check(L.cleora_stream_create(ctypes.byref(sa)))
    _hip.check(L.cleora_stream_create(ctypes.byref(sb)))

    def wall(fns, reps=3):
        for f in fns:
            f()
        _hip.check(L.cleora_stream_sync(sa)); _hip.check(L.cleora_stream_sync(sb))
        t0 = time.perf_counter()
        for _ in range(reps):
            for f in fns:
                f()
        _hip.check(L.cleora_stream_sync(sa)); _hip.check(L.cleora_stream_sync(sb))
        return (time.perf_counter() - t0) / reps * 1e3

    side.side_valu.argtypes = [u32, u32, u32, vp, vp]
    cus = torch.cuda.get_device_properties(dev).multi_processor_count

    def masked(f):
        words = (cus + 31) // 32
        mask = (ctypes.c_uint32 * words)()
        for c in range(cus):
            if f(c):
                mask[c >> 5] |= 1 << (c & 31)
        st = ctypes.c_void_p()
        _hip.check(L.cleora_stream_create_cu_mask(ctypes.byref(st), mask, words))
        return st

    def run(sa_, sb_, launch_side, k, order):
        """own durations (events on each stream) and the wall clock of one concurrent pair"""
        ta, tb = torch.cuda.ExternalStream(sa_.value), torch.cuda.ExternalStream(sb_.value)
        spmm = lambda: _hip.check(L.cleora_propagate_dev(gr.handle, _hip.LEFT, a.data_ptr(), d, d, b.data_ptr(), d, _hip.F_L2NORM, 0.0, None, None, None, sa_))
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        out = []
        for rep in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for what in order:
                if what == "side":
                    ev[0].record(tb); launch_side(k, sb_); ev[1].record(tb)
                else:
                    ev[2].record(ta); spmm(); ev[3].record(ta)
            torch.cuda.synchronize()
            w = (time.perf_counter() - t0) * 1e3
            out = [w, ev[0].elapsed_time(ev[1]) if "side" in order else None, ev[2].elapsed_time(ev[3]) if "spmm" in order else None]
        return out

    plain = (sa, sb)
    ma, mb = masked(lambda c: c % 4 != 3), masked(lambda c: c % 4 == 3)
    cases = [("spin (sleeping), 256 x 256", lambda k, st: side.side_spin(k, 256, 256, sink.data_ptr(), st), 1_000_000),
             ("vector ALU fma, 256 x 256 (1 wave per SIMD)", lambda k, st: side.side_valu(k, 256, 256, sink.data_ptr(), st), 20_000),
             ("vector ALU fma, 64 x 256", lambda k, st: side.side_valu(k, 64, 256, sink.data_ptr(), st), 20_000),
             ("bf16 MFMA from registers, 256 x 256", lambda k, st: side.side_mfma(k, 256, 256, sink.data_ptr(), st), 20_000),
             ("bf16 MFMA from registers, 64 x 256", lambda k, st: side.side_mfma(k, 64, 256, sink.data_ptr(), st), 20_000),
             ("bf16 MFMA from registers, 64 x 64 (one wave per CU on 64 CUs)", lambda k, st: side.side_mfma(k, 64, 64, sink.data_ptr(), st), 20_000),
             ("streaming read of another 10 GB, 1024 blocks", lambda k, st: side.side_read(other.data_ptr(), n * d * 4, k, 1024, sink.data_ptr(), st), 1)]
    for name, launch, k0 in cases:
        for label, (xa, xb) in (("plain streams", plain), ("SpMM on 3 of 4 CUs, side on every 4th", (ma, mb))):
            t = run(xa, xb, launch, k0, ["side"])[1]
            k = max(1, int(k0 * 25.0 / max(t, 1e-3)))
            alone = run(xa, xb, launch, k, ["side"])
            sp = run(xa, xb, launch, k, ["spmm"])
            both = run(xa, xb, launch, k, ["side", "spmm"])
            print(json.dumps({"mode": "side_load", "side": name, "streams": label, "side_alone_ms": alone[1], "spmm_alone_ms": sp[2],
                              "both_wall_ms": both[0], "both_side_own_ms": both[1], "both_spmm_own_ms": both[2]}), flush=True)


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
    if mode == "side_load":
        side_load(*(args or [10_000_000, 95_000_000, 256]))
    elif mode == "overlap":
        overlap(*(args or [10_000_000, 95_000_000, 256]))
    elif mode == "spmm_masks":
        spmm_masks(*(args or [10_000_000, 95_000_000, 256]))
    elif mode == "loop_masks":
        loop_masks(*(args or [10_000_000, 95_000_000, 256]))
    elif mode == "project":
        project_only(*(args + [10_000_000, 256][len(args):]))
    elif mode == "kernels":
        kernels(*(args + [10_000_000, 256][len(args):]))
    else:
        loop(*(args + [10_000_000, 95_000_000, 256][len(args):]))
