"""Round 6 experiment: does PLACEMENT of the iterate's rows change what the SpMM costs?

  (a) row order: the unchanged kernel on the C3 graph as generated (hot rows scattered by the generator's permutation) against
      the same graph RELABELLED so that rows are placed by descending in-degree — all rows, or only the hot set (the 256 Ki / 1 Mi
      most referenced rows) moved to the front.  The relabelling keeps every row's edge ORDER, so Y'[pi(i)] == Y[i] bit for
      bit (checked).  This is the upper bound of what an internal degree-ordered placement inside the library could win,
      measured before building it.
  (b) mapping: the same launch on iterates from plain hipMalloc against iterates mapped through the HIP virtual-memory API with
      one physical chunk / 1 GiB chunks / 2 MiB chunks and 2 MiB / 1 GiB virtual alignment (scripts/r06/vmm_probe.hip).

    python scripts/r06/placement_probe.py [C3|C2|C4s] [rows|vmm|both]
"""
import ctypes
import json
import os
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from cleora_amd import _hip  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
what = sys.argv[2] if len(sys.argv) > 2 else "both"
dev = torch.device("cuda:0")
L = _hip.lib()
S = torch.cuda.current_stream().cuda_stream


class A:
    config, nodes, pairs, hyperedges, products, dim = cfg, 0, 0, 0, 0, 0


g, hashes, label, c = bench.make_workload(A, dev, 0, 1, False)
n, nnz, d = g["n"], g["nnz"], c["dim"]
out = {"config": cfg, "n": n, "nnz": nnz, "d": d}


def graph_of(gd):
    return _hip.Graph.from_device(n, n, nnz, gd["rowptr"].data_ptr(), gd["col"].data_ptr(), gd["val_left"].data_ptr(), None, 0,
                                  keepalive=(gd["rowptr"], gd["col"], gd["val_left"]))


def time_launch(graph, xp, yp, iters=10, warm=4):
    """ms per SpMM + fused L2 launch, ping-pong between the two buffers (raw device pointers)."""
    p, q = xp, yp
    for _ in range(warm):
        _hip.check(L.cleora_propagate_dev(graph.handle, 0, p, d, d, q, d, _hip.F_L2NORM, 0.0, None, None, None, S))
        p, q = q, p
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        _hip.check(L.cleora_propagate_dev(graph.handle, 0, p, d, d, q, d, _hip.F_L2NORM, 0.0, None, None, None, S))
        p, q = q, p
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def relabel(gd, pi):
    """Row i of the graph becomes row pi[i]; columns relabelled; every row keeps its edge order."""
    deg = torch.diff(gd["rowptr"])
    old_row = torch.repeat_interleave(torch.arange(n, device=dev), deg)
    new_row = pi[old_row]
    del old_row
    order = torch.sort(new_row, stable=True).indices
    del new_row
    col = pi[gd["col"].long()[order]].to(torch.int32)
    val = gd["val_left"][order].contiguous()
    del order
    inv = torch.empty_like(pi)
    inv[pi] = torch.arange(n, device=dev)
    rowptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    torch.cumsum(deg[inv], 0, out=rowptr[1:])
    return {"rowptr": rowptr, "col": col, "val_left": val}


x = torch.empty((n, d), dtype=torch.float32, device=dev)
y = torch.empty((n, d), dtype=torch.float32, device=dev)
_hip.check(L.cleora_init_dev(hashes.data_ptr(), n, d, 0, x.data_ptr(), d, S))
x0 = x.clone()
base = graph_of(g)

if what in ("rows", "both"):
    indeg = torch.bincount(g["col"].long(), minlength=n)
    res = {}
    res["as_generated"] = [round(time_launch(base, x.data_ptr(), y.data_ptr()), 3) for _ in range(3)]
    x.copy_(x0)
    _hip.check(L.cleora_propagate_dev(base.handle, 0, x.data_ptr(), d, d, y.data_ptr(), d, _hip.F_L2NORM, 0.0, None, None, None, S))
    torch.cuda.synchronize()
    y_ref = y.clone()
    by_deg = torch.sort(indeg, descending=True, stable=True).indices         # by_deg[k] = the row placed k-th
    variants = {}
    pi_full = torch.empty(n, dtype=torch.int64, device=dev)
    pi_full[by_deg] = torch.arange(n, device=dev)
    variants["all_rows_by_indegree"] = pi_full
    for k in (262_144, 1_048_576):
        hot = torch.zeros(n, dtype=torch.bool, device=dev)
        hot[by_deg[:k]] = True
        pi = torch.empty(n, dtype=torch.int64, device=dev)
        pi[by_deg[:k]] = torch.arange(k, device=dev)                         # hottest first
        cold = (~hot).nonzero().squeeze(1)
        pi[cold] = k + torch.arange(n - k, device=dev)                       # the rest keep their relative order
        variants[f"hot_{k}_rows_first"] = pi
        del hot, cold
    for name, pi in variants.items():
        gd = relabel(g, pi)
        gr = graph_of(gd)
        inv = torch.empty_like(pi)
        inv[pi] = torch.arange(n, device=dev)
        x.copy_(x0[inv])                                                     # X'[pi(i)] = X[i]
        _hip.check(L.cleora_propagate_dev(gr.handle, 0, x.data_ptr(), d, d, y.data_ptr(), d, _hip.F_L2NORM, 0.0, None, None, None, S))
        torch.cuda.synchronize()
        same = bool(torch.equal(y[pi].view(torch.int32), y_ref.view(torch.int32)))   # Y'[pi(i)] == Y[i]
        res[name] = {"ms": [round(time_launch(gr, x.data_ptr(), y.data_ptr()), 3) for _ in range(3)], "bit_equal_to_unpermuted": same}
        gr.close()
        del gd, inv
    res["as_generated_again"] = [round(time_launch(base, x.data_ptr(), y.data_ptr()), 3) for _ in range(2)]
    out["row_order"] = res
    print(json.dumps(out), flush=True)

if what in ("vmm", "both"):
    del x, y, x0
    torch.cuda.empty_cache()
    so = os.path.join(ROOT, "gpurun_out", "libvmm_probe.so")
    os.makedirs(os.path.dirname(so), exist_ok=True)
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", os.path.join(ROOT, "scripts", "r06", "vmm_probe.hip"), "-o", so])
    V = ctypes.CDLL(so)
    V.vmm_alloc.argtypes = [ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t)]
    V.vmm_free.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    V.plain_alloc.argtypes = [ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p)]
    V.plain_free.argtypes = [ctypes.c_void_p]
    gmin, grec = ctypes.c_size_t(0), ctypes.c_size_t(0)
    V.vmm_granularity(0, ctypes.byref(gmin), ctypes.byref(grec))
    res = {"granularity_min": gmin.value, "granularity_recommended": grec.value}
    nbytes = n * d * 4

    def run(kind, align=0, chunk=0):
        ptrs = []
        for _ in range(2):
            p, m = ctypes.c_void_p(0), ctypes.c_size_t(0)
            rc = V.plain_alloc(nbytes, ctypes.byref(p)) if kind == "plain" else V.vmm_alloc(0, nbytes, align, chunk, ctypes.byref(p), ctypes.byref(m))
            if rc != 0:
                return {"error": rc}
            ptrs.append((p, m))
        _hip.check(L.cleora_init_dev(hashes.data_ptr(), n, d, 0, ptrs[0][0].value, d, S))
        ms = [round(time_launch(base, ptrs[0][0].value, ptrs[1][0].value), 3) for _ in range(3)]
        addr = [hex(p.value) for p, _ in ptrs]
        for p, m in ptrs:
            V.plain_free(p) if kind == "plain" else V.vmm_free(p, m)
        return {"ms": ms, "addresses": addr}

    G = 1 << 30
    for rep in range(2):
        res.setdefault("plain_hipMalloc", []).append(run("plain"))
        res.setdefault("vmm_one_chunk_align_2M", []).append(run("vmm", 2 << 20, 0))
        res.setdefault("vmm_one_chunk_align_1G", []).append(run("vmm", G, 0))
        res.setdefault("vmm_1G_chunks_align_1G", []).append(run("vmm", G, G))
        res.setdefault("vmm_2M_chunks_align_2M", []).append(run("vmm", 2 << 20, 2 << 20) if nbytes < (32 << 30) else {"skipped": "too many chunks"})
    out["mapping"] = res
    print(json.dumps(out), flush=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"r06_placement_probe_{cfg}_{what}.json"), "w"), indent=1)
