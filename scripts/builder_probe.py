#!/usr/bin/env python3
"""Developer probe: host builder scaling on a C5-like hypergraph (arity 2+Poisson(6), Zipf members)."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cleora_amd import _host
rng = np.random.default_rng(5)
n_lines = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
ar = np.minimum(16, 2 + rng.poisson(6, n_lines))
members = np.minimum(rng.zipf(1.2, size=int(ar.sum())) - 1, 2_000_000 - 1)
t0 = time.perf_counter()
pos = 0; lines = []
for k in ar:
    lines.append(" ".join(map(str, members[pos:pos + k]))); pos += k
data, off = _host.pack_strings(lines)
print(f"python: {n_lines} lines generated+packed in {time.perf_counter()-t0:.1f} s", flush=True)
L = _host.lib(); ref = None
for t in (1, 8, 32, 64):
    L.cleora_host_set_threads(t)
    h = ctypes.c_void_p(); t2 = time.perf_counter()
    L.cleora_host_build_from_lines(data, off.ctypes.data_as(ctypes.c_void_p), len(lines), b"complex::reflexive::product", 16, ctypes.byref(h))
    t3 = time.perf_counter()
    hg = _host.HostGraph(h); n, nnz, _ = hg.sizes()
    ser = hg.serialize() if n_lines <= 2_000_000 else b""
    ref = ref or ser
    print(f"threads={t:3d}: {t3-t2:6.2f} s  n={n} nnz={nnz}  identical={ser == ref}", flush=True)
