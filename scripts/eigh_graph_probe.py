#!/usr/bin/env python3
"""Developer probe (round 2): is rocSOLVER's dsyevd (6 ms at d = 256: ~1000 dependent small launches) host-launch
bound, i.e. does replaying it from a hipGraph shorten it?  Captures cleora_whiten_transform_dev (cov scale ->
dsyevd -> transform) on a side stream after one eager warm-up call (so rocBLAS' workspace is already sized),
replays it, and compares time and output bits with the eager call."""
import ctypes, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cleora_amd import _hip
dev = torch.device("cuda:0")
L = _hip.lib()
torch.zeros(1, device=dev)
rt = None
for m in open("/proc/self/maps"):
    if "libamdhip64" in m:
        rt = ctypes.CDLL(m.split()[-1]); break
for d in (64, 256, 1024):
    n = 100000
    rng = np.random.default_rng(d)
    a = rng.standard_normal((d, 4 * d))
    gram = torch.from_numpy(a @ a.T * n / (4 * d)).to(dev)
    ws = torch.empty(L.cleora_eigh_workspace(d), dtype=torch.uint8, device=dev)
    t_eager, t_graph = torch.empty((d, d), dtype=torch.float32, device=dev), torch.empty((d, d), dtype=torch.float32, device=dev)
    st = torch.cuda.Stream()
    sp = ctypes.c_void_p(st.cuda_stream)
    def call(out):
        _hip.check(L.cleora_whiten_transform_dev(gram.data_ptr(), n, d, d, out.data_ptr(), None, ws.data_ptr(), sp))
    call(t_eager); st.synchronize()
    def timed(fn, reps=10):
        st.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): fn()
        st.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
    eager_ms = timed(lambda: call(t_eager))
    graph, gexec = ctypes.c_void_p(), ctypes.c_void_p()
    rc = rt.hipStreamBeginCapture(sp, ctypes.c_int(2))          # hipStreamCaptureModeRelaxed
    err = None
    try:
        call(t_graph)
    except Exception as ex:
        err = ex
    rc2 = rt.hipStreamEndCapture(sp, ctypes.byref(graph))
    if rc or rc2 or err or not graph.value:
        print(f"d={d}: eager {eager_ms:.2f} ms; capture failed (begin {rc}, end {rc2}, {err})", flush=True)
        rt.hipGetLastError()
        continue
    nn = ctypes.c_size_t(0)
    rt.hipGraphGetNodes(graph, None, ctypes.byref(nn))
    rc3 = rt.hipGraphInstantiate(ctypes.byref(gexec), graph, None, None, ctypes.c_size_t(0))
    if rc3:
        print(f"d={d}: eager {eager_ms:.2f} ms; {nn.value} nodes captured; instantiate failed ({rc3})", flush=True)
        continue
    rt.hipGraphLaunch(gexec, sp); st.synchronize()
    graph_ms = timed(lambda: rt.hipGraphLaunch(gexec, sp))
    same = bool(torch.equal(t_eager, t_graph))
    print(f"d={d}: eager {eager_ms:.2f} ms   graph replay {graph_ms:.2f} ms   nodes {nn.value}   identical output {same}", flush=True)
