"""ctypes binding of libcleora_hip.so (include/cleora_hip.h).

This is the only place Python touches the HIP library.  There is no CPU
fallback: if the shared object is missing or a call fails, a RuntimeError /
ValueError is raised with cleora_last_error().
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcleora_hip.so")

OK, E_INVALID, E_OOM, E_HIP, E_NODEVICE, E_RCCL = 0, -1, -2, -3, -4, -5
LEFT, SYMMETRIC = 0, 1
F_L2NORM, F_FASTNORM, F_RESIDUAL, F_SQDIFF, F_ROWSQ, F_SCALE, F_WHITEN = 1, 2, 4, 8, 16, 32, 64
F_L1NORM, F_BLEND_ANY, F_SQDIFF64, F_HUB_SEGMENTS, F_ROWSQ_CONT = 128, 256, 512, 1024, 2048
ABI_VERSION = 5
COMM_ID_BYTES = 128
ALLGATHER_RING, ALLGATHER_P2P, ALLGATHER_PEER = 0, 1, 2
BALANCE_AUTO, BALANCE_ROWS, BALANCE_NNZ = 0, 1, 2

c_u64, c_u32, c_i64, c_int, c_f32 = (ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int64,
                                     ctypes.c_int, ctypes.c_float)
vp = ctypes.c_void_p


class GraphInfo(ctypes.Structure):
    _fields_ = [("n_rows", c_u64), ("n_cols", c_u64), ("nnz", c_u64), ("n_hub_rows", c_u64),
                ("n_hub_segments", c_u64), ("device_bytes", c_u64), ("hot_rows", c_u64), ("hub_threshold", c_u32),
                ("hub_segment", c_u32), ("device", ctypes.c_int32), ("has_symmetric", ctypes.c_int32),
                ("n_inorder_rows", c_u64), ("hub_inorder_min", c_u64)]


class ShardedInfo(ctypes.Structure):
    _fields_ = [("n", c_u64), ("n_pad", c_u64), ("local_rows", c_u64), ("local_nnz", c_u64), ("device_bytes", c_u64),
                ("steps", c_u32), ("rank", ctypes.c_int32), ("world", ctypes.c_int32), ("balance", ctypes.c_int32),
                ("has_symmetric", ctypes.c_int32)]


class ColShardedInfo(ctypes.Structure):
    _fields_ = [("n", c_u64), ("nnz", c_u64), ("d_total", c_u32), ("d_local", c_u32), ("col_begin", c_u32), ("steps", c_u32),
                ("rank", ctypes.c_int32), ("world", ctypes.c_int32), ("has_symmetric", ctypes.c_int32), ("reserved", ctypes.c_int32)]


class MultiInfo(ctypes.Structure):
    _fields_ = [("n", c_u64), ("nnz", c_u64), ("n_pad", c_u64), ("world", c_u32), ("steps", c_u32),
                ("has_symmetric", ctypes.c_int32), ("reserved", ctypes.c_int32), ("device", ctypes.c_int32 * 64),
                ("local_rows", c_u64 * 64), ("local_nnz", c_u64 * 64), ("device_bytes", c_u64 * 64)]


# name -> (restype, argtypes); mirrors include/cleora_hip.h one to one
SIGNATURES = {
    "cleora_multi_create": (c_int, [vp, c_u32, c_u64, c_u64, vp, vp, vp, vp, c_u32, c_int, ctypes.POINTER(vp)]),
    "cleora_multi_destroy": (c_int, [vp]),
    "cleora_multi_get_info": (c_int, [vp, ctypes.POINTER(MultiInfo)]),
    "cleora_multi_embed": (c_int, [vp, vp, vp, c_int, c_u32, c_u64, c_i64, c_f32, c_f32, c_u32, vp, ctypes.POINTER(c_u64)]),
    "cleora_multi_propagate": (c_int, [vp, c_int, vp, c_u32, vp]),
    "cleora_sharded_set_stream": (c_int, [vp, vp]),
    "cleora_comm_local_id": (c_int, [vp]),
    "cleora_comm_create_local": (c_int, [vp, c_int, c_int, c_int, ctypes.POINTER(vp)]),
    "cleora_comm_enable_peer": (c_int, [vp]),
    "cleora_comm_register": (c_int, [vp, vp, c_u64]),
    "cleora_comm_unregister": (c_int, [vp, vp]),
    "cleora_comm_check": (c_int, [vp]),
    "cleora_comm_selftest": (c_int, [vp, c_u32]),
    "cleora_comm_peer_mode": (c_int, [vp, ctypes.POINTER(c_int)]),
    "cleora_sharded_plan": (c_int, [c_u64, vp, c_u32, c_u32, c_int, vp, ctypes.POINTER(c_u64), ctypes.POINTER(c_int)]),
    "cleora_sharded_create": (c_int, [vp, c_int, c_u64, c_u64, vp, vp, vp, vp, c_int, c_u32, c_int, ctypes.POINTER(vp)]),
    "cleora_sharded_destroy": (c_int, [vp]),
    "cleora_sharded_get_info": (c_int, [vp, ctypes.POINTER(ShardedInfo)]),
    "cleora_sharded_bounds": (c_int, [vp, vp]),
    "cleora_sharded_block": (c_int, [vp, c_u32, ctypes.POINTER(vp), ctypes.POINTER(c_u64), ctypes.POINTER(c_u64)]),
    "cleora_sharded_propagate_dev": (c_int, [vp, c_int, vp, vp, c_u32, c_u32, c_f32, vp, c_int, vp]),
    "cleora_sharded_set_timing": (c_int, [vp, c_int]),
    "cleora_sharded_get_timing": (c_int, [vp, ctypes.POINTER(ctypes.c_double * 2), ctypes.POINTER(c_u64)]),
    "cleora_embed_sharded_bytes": (c_u64, [c_u64, c_u64, c_u64, c_u32, c_u32, c_u32]),
    "cleora_sharded_debug_fail_first_gather": (c_int, [vp, c_int]),
    "cleora_colsharded_create": (c_int, [vp, c_int, c_u64, c_u64, vp, vp, vp, vp, c_int, c_u32, c_u32, ctypes.POINTER(vp)]),
    "cleora_colsharded_destroy": (c_int, [vp]),
    "cleora_colsharded_get_info": (c_int, [vp, vp]),
    "cleora_colsharded_block": (c_int, [vp, c_u32, ctypes.POINTER(vp), ctypes.POINTER(c_u64), ctypes.POINTER(c_u64)]),
    "cleora_colsharded_propagate_dev": (c_int, [vp, c_int, vp, vp, c_u32, c_f32, vp, vp]),
    "cleora_embed_colsharded": (c_int, [vp, vp, c_int, c_u64, c_f32, c_f32, c_u32, ctypes.POINTER(c_u64)]),
    "cleora_embed_sharded": (c_int, [vp, vp, c_int, c_u32, c_u64, c_f32, c_f32, c_u32, ctypes.POINTER(c_u64)]),
    "cleora_abi_version": (c_int, []),
    "cleora_last_error": (ctypes.c_char_p, []),
    "cleora_device_count": (c_int, [ctypes.POINTER(c_int)]),
    "cleora_set_device": (c_int, [c_int]),
    "cleora_malloc": (c_int, [c_u64, ctypes.POINTER(vp)]),
    "cleora_free": (c_int, [vp]),
    "cleora_memcpy_h2d": (c_int, [vp, vp, c_u64, vp]),
    "cleora_memcpy_d2h": (c_int, [vp, vp, c_u64, vp]),
    "cleora_memcpy_d2d": (c_int, [vp, vp, c_u64, vp]),
    "cleora_memset": (c_int, [vp, c_int, c_u64, vp]),
    "cleora_stream_sync": (c_int, [vp]),
    "cleora_stream_create": (c_int, [ctypes.POINTER(vp)]),
    "cleora_stream_destroy": (c_int, [vp]),
    "cleora_stream_wait_stream": (c_int, [vp, vp]),
    "cleora_comm_unique_id": (c_int, [vp]),
    "cleora_comm_create": (c_int, [vp, c_int, c_int, c_int, ctypes.POINTER(vp)]),
    "cleora_comm_destroy": (c_int, [vp]),
    "cleora_comm_info": (c_int, [vp, ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "cleora_comm_set_allgather": (c_int, [vp, c_int]),
    "cleora_comm_get_allgather": (c_int, [vp, ctypes.POINTER(c_int)]),
    "cleora_allgatherv_f32_dev": (c_int, [vp, vp, vp, vp]),
    "cleora_allgather_f32_dev": (c_int, [vp, vp, c_u64, vp]),
    "cleora_allreduce_f32_dev": (c_int, [vp, vp, c_u64, vp]),
    "cleora_allreduce_f64_dev": (c_int, [vp, vp, c_u64, vp]),
    "cleora_broadcast_dev": (c_int, [vp, vp, c_u64, c_int, vp]),
    "cleora_alltoall_f32_dev": (c_int, [vp, vp, vp, c_u64, vp]),
    "cleora_graph_create": (c_int, [c_int, c_u64, c_u64, c_u64, vp, vp, vp, vp, c_u32, c_u32,
                                    ctypes.POINTER(vp)]),
    "cleora_graph_create_dev": (c_int, [c_int, c_u64, c_u64, c_u64, vp, vp, vp, vp, c_u32, c_u32,
                                        ctypes.POINTER(vp)]),
    "cleora_graph_destroy": (c_int, [vp]),
    "cleora_graph_get_info": (c_int, [vp, ctypes.POINTER(GraphInfo)]),
    "cleora_graph_set_hot_cache": (c_int, [vp, c_i64]),
    "cleora_graph_set_hub_lanes": (c_int, [vp, c_int]),
    "cleora_graph_set_hub_chain_min": (c_int, [vp, c_u64]),
    "cleora_graph_set_hub_inorder_min": (c_int, [vp, c_u64]),
    "cleora_graph_set_timing": (c_int, [vp, c_int]),
    "cleora_graph_get_timing": (c_int, [vp, ctypes.POINTER(ctypes.c_double * 3), ctypes.POINTER(c_u64)]),
    "cleora_alloc_iterates": (c_int, [vp, c_u32, c_u32, ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_double * 2)]),
    "cleora_alloc_iterates_for": (c_int, [vp, c_u32, c_u32, c_u64, ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_double * 2)]),
    "cleora_propagate_dev": (c_int, [vp, c_int, vp, c_u64, c_u32, vp, c_u64, c_u32, c_f32, vp, vp, vp, vp]),
    "cleora_propagate_vals_dev": (c_int, [vp, vp, vp, c_u64, c_u32, vp, c_u64, c_u32, c_f32, vp, vp, vp, vp]),
    "cleora_propagate_attention_dev": (c_int, [vp, c_int, vp, c_u64, c_u32, c_f32, vp, c_u64, c_u32, c_f32, vp, vp, vp]),
    "cleora_edge_attention_dev": (c_int, [vp, c_int, vp, c_u64, c_u32, c_f32, vp, vp]),
    "cleora_rowops_dev": (c_int, [vp, c_u64, c_u64, c_u32, vp, c_u64, c_u32, c_f32, vp, vp, vp, vp]),
    "cleora_init_dev": (c_int, [vp, c_u64, c_u32, c_i64, vp, c_u64, vp]),
    "cleora_reduce_workspace": (c_u64, [c_u64]),
    "cleora_reduce_sum_f64_dev": (c_int, [vp, c_u64, vp, vp, vp]),
    "cleora_colsum_workspace": (c_u64, [c_u64, c_u32]),
    "cleora_colsum_dev": (c_int, [vp, c_u64, c_u64, c_u32, vp, vp, vp]),
    "cleora_gram_workspace": (c_u64, [c_u64, c_u32]),
    "cleora_centered_gram_dev": (c_int, [vp, c_u64, c_u64, c_u32, vp, vp, vp, vp]),
    "cleora_project_dev": (c_int, [vp, c_u64, c_u64, c_u32, vp, vp, c_u32, vp, c_u64, vp]),
    "cleora_project_general_dev": (c_int, [vp, c_u64, c_u64, c_u32, vp, vp, c_u32, vp, c_u64, vp, vp, c_u64, c_f32, c_f32, c_int,
                                           ctypes.POINTER(c_int), vp]),
    "cleora_csr_rowsum_dev": (c_int, [vp, c_int, vp, vp]),
    "cleora_csr_rowsums_dev": (c_int, [vp, c_int, vp, vp, vp]),
    "cleora_project_bounded_dev": (c_int, [vp, c_u64, c_u64, c_u32, vp, vp, c_u32, vp, c_u64, vp, vp, c_int, ctypes.POINTER(c_int),
                                           ctypes.POINTER(c_int), vp]),
    "cleora_mean_dev": (c_int, [vp, c_u64, c_u32, vp, vp, vp]),
    "cleora_eigh_workspace": (c_u64, [c_u32]),
    "cleora_whiten_transform_dev": (c_int, [vp, c_u64, c_u32, c_u32, vp, vp, vp, vp]),
    "cleora_whiten_stats_dev": (c_int, [vp, c_u64, c_u64, c_u32, vp, c_int, vp, vp, vp]),
    "cleora_whiten_transform_any_dev": (c_int, [vp, c_u64, c_u32, vp, vp, vp, ctypes.POINTER(c_int)]),
    "cleora_whiten_workspace": (c_u64, [c_u64, c_u32]),
    "cleora_whiten_dev": (c_int, [vp, c_u64, c_u64, c_u32, c_u32, vp, c_u64, vp, vp, vp]),
    "cleora_whiten": (c_int, [vp, c_u64, c_u32, c_u32, vp]),
    "cleora_whiten_set_timing": (c_int, [c_int]),
    "cleora_whiten_get_timing": (c_int, [ctypes.POINTER(ctypes.c_double * 4), ctypes.POINTER(c_u64)]),
    "cleora_cosine_scores_dev": (c_int, [vp, c_u64, c_u64, c_u32, vp, vp, vp]),
    "cleora_topk_workspace": (c_u64, [c_u64, c_u32]),
    "cleora_topk_workspace_for": (c_u64, [c_u64, c_u32, c_u32]),
    "cleora_topk_last_route": (c_int, []),
    "cleora_topk_set_route": (c_int, [c_int]),
    "cleora_cholesky_whiten_host": (c_int, [vp, c_u64, c_u32, vp, vp]),
    "cleora_topk_cosine_dev": (c_int, [vp, vp, c_u64, c_u64, c_u32, vp, c_u32, c_u32, c_int, c_int, vp, vp, vp, vp]),
    "cleora_propagate": (c_int, [vp, c_int, vp, c_u32, vp]),
    "cleora_l2_normalize": (c_int, [vp, c_u64, c_u32, vp]),
    "cleora_init": (c_int, [vp, c_u64, c_u32, c_i64, vp]),
    "cleora_embed": (c_int, [vp, vp, vp, c_int, c_u32, c_u64, c_i64, c_f32, c_f32, c_u32, vp,
                             ctypes.POINTER(c_u64)]),
    "cleora_last_embed_loop_ms": (ctypes.c_double, []),
    "cleora_embed_dev": (c_int, [vp, vp, c_int, c_u32, c_u64, c_f32, c_f32, c_u32, ctypes.POINTER(c_u64)]),
}

_lib = None


def _preload_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so (same SONAME as /opt/rocm's).  Two HIP
    runtimes in one process cannot both own the GPU ("no ROCm-capable device is detected" in
    whichever initialises second), so when torch is installed its copy is loaded first and
    libcleora_hip.so's NEEDED libamdhip64.so.7 then resolves to that already-loaded object —
    whatever the import order.  Without torch the system runtime under /opt/rocm is used."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    cand = os.path.join(libdir, "libamdhip64.so")
    if os.path.exists(cand):
        try:
            ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass
    # the whitening eigensolver is dlopen'ed by the library on first use (csrc/eigh.hip): point it at the
    # rocSOLVER that belongs to the same ROCm build as that runtime (nothing is loaded here)
    solver = os.path.join(libdir, "librocsolver.so")
    if os.path.exists(solver):
        os.environ.setdefault("CLEORA_ROCSOLVER", solver)
    # likewise RCCL for the multi-GPU entry points (csrc/comm.hip)
    rccl = os.path.join(libdir, "librccl.so")
    if os.path.exists(rccl):
        os.environ.setdefault("CLEORA_RCCL", rccl)


def lib():
    """Loads libcleora_hip.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(cleora_amd/csrc/build.sh).  cleora_amd has no CPU fallback.")
        _preload_hip_runtime()
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        if L.cleora_abi_version() != ABI_VERSION:
            raise RuntimeError("libcleora_hip.so ABI version mismatch: rebuild it (cleora_amd/csrc/build.sh)")
        _lib = L
    return _lib


def last_error():
    return lib().cleora_last_error().decode("utf-8", "replace")


def check(rc):
    """Maps a status code to the exceptions the PyO3 class raises (src/lib.rs):
    bad arguments -> ValueError, everything else -> RuntimeError."""
    if rc == OK:
        return
    msg = last_error()
    if rc == E_INVALID:
        raise ValueError(msg)
    if rc == E_OOM:
        raise MemoryError(msg)
    raise RuntimeError(msg)


def device_count():
    n = c_int(0)
    check(lib().cleora_device_count(ctypes.byref(n)))
    return n.value


def ptr(a):
    """Host pointer of a C-contiguous numpy array (or None)."""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(vp)


class Graph:
    """Owns a cleora_graph handle (device CSR shard)."""

    def __init__(self, handle, keepalive=None):
        self.handle = handle
        self._keepalive = keepalive

    @classmethod
    def from_host(cls, rowptr, col, val_left, val_sym=None, n_cols=None, device=0,
                  hub_threshold=0, hub_segment=0):
        rowptr = np.ascontiguousarray(rowptr, dtype=np.uint64)
        col = np.ascontiguousarray(col, dtype=np.uint32)
        val_left = np.ascontiguousarray(val_left, dtype=np.float32)
        if val_sym is not None:
            val_sym = np.ascontiguousarray(val_sym, dtype=np.float32)
        n_rows = rowptr.shape[0] - 1
        nnz = col.shape[0]
        if n_cols is None:
            n_cols = n_rows
        if val_left.shape[0] != nnz or (val_sym is not None and val_sym.shape[0] != nnz):
            raise ValueError("col / val length mismatch")
        h = vp()
        check(lib().cleora_graph_create(device, n_rows, n_cols, nnz, ptr(rowptr), ptr(col),
                                        ptr(val_left), ptr(val_sym), hub_threshold, hub_segment,
                                        ctypes.byref(h)))
        return cls(h)

    @classmethod
    def from_device(cls, n_rows, n_cols, nnz, rowptr_ptr, col_ptr, val_left_ptr, val_sym_ptr=None,
                    device=0, hub_threshold=0, hub_segment=0, keepalive=None):
        """Adopts device arrays (e.g. torch tensors' data_ptr()); keepalive holds their owners."""
        h = vp()
        check(lib().cleora_graph_create_dev(device, n_rows, n_cols, nnz, rowptr_ptr, col_ptr,
                                            val_left_ptr, val_sym_ptr, hub_threshold, hub_segment,
                                            ctypes.byref(h)))
        return cls(h, keepalive)

    def info(self):
        gi = GraphInfo()
        check(lib().cleora_graph_get_info(self.handle, ctypes.byref(gi)))
        return gi

    def set_hot_cache(self, hot_bytes):
        """-1 automatic, 0 off, > 0 forced byte budget (include/cleora_hip.h)."""
        check(lib().cleora_graph_set_hot_cache(self.handle, int(hot_bytes)))

    def set_hub_inorder_min(self, min_edges):
        """Long rows with more than min_edges edges run on the in-order hub launch, the others first in the main launch (same bits)."""
        check(lib().cleora_graph_set_hub_inorder_min(self.handle, int(min_edges)))

    def set_hub_chain_min(self, min_edges):
        """cleora_graph_set_hub_chain_min: rows of the hub launch with at least this many edges take the chain kernel (0 = automatic)."""
        check(lib().cleora_graph_set_hub_chain_min(self.handle, int(min_edges)))

    def set_hub_lanes(self, lanes):
        """Lanes per edge of the in-order hub launch: 0 automatic, 4 / 2 forced (same bits either way)."""
        check(lib().cleora_graph_set_hub_lanes(self.handle, int(lanes)))

    def set_timing(self, enable):
        check(lib().cleora_graph_set_timing(self.handle, 1 if enable else 0))

    def get_timing(self):
        """(ms[hub_partial, spmm_rows, hub_finish] summed, calls) since the last query."""
        ms = (ctypes.c_double * 3)()
        calls = c_u64(0)
        check(lib().cleora_graph_get_timing(self.handle, ctypes.byref(ms), ctypes.byref(calls)))
        return [ms[0], ms[1], ms[2]], calls.value

    def close(self):
        if self.handle:
            lib().cleora_graph_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MultiGraph:
    """Owns a cleora_multi handle: the row partition over several devices of THIS process (csrc/multi.hip).  devices may repeat
    (several shards on one GPU — the one-GPU test form)."""

    def __init__(self, handle, devices):
        self.handle = handle
        self.devices = list(devices)

    @classmethod
    def from_host(cls, devices, rowptr, col, val_left, val_sym=None, steps=0, balance=BALANCE_AUTO):
        devices = [int(v) for v in devices]
        if not devices:
            raise ValueError("need at least one device")
        rowptr = np.ascontiguousarray(rowptr, dtype=np.uint64)
        col = np.ascontiguousarray(col, dtype=np.uint32)
        val_left = np.ascontiguousarray(val_left, dtype=np.float32)
        if val_sym is not None:
            val_sym = np.ascontiguousarray(val_sym, dtype=np.float32)
        ids = (ctypes.c_int * len(devices))(*devices)
        h = vp()
        check(lib().cleora_multi_create(ids, len(devices), rowptr.shape[0] - 1, col.shape[0], ptr(rowptr), ptr(col), ptr(val_left),
                                        ptr(val_sym), int(steps), int(balance), ctypes.byref(h)))
        return cls(h, devices)

    def info(self):
        mi = MultiInfo()
        check(lib().cleora_multi_get_info(self.handle, ctypes.byref(mi)))
        return mi

    def embed(self, hashes, x0, kind, d, iterations, seed=0, residual_weight=0.0, threshold=0.0, flags=0):
        """(out n x d, iterations run): cleora_multi_embed.  hashes (u64[n]) or x0 (n x d f32) is the start."""
        n = int(self.info().n)
        out = np.empty((n, d), np.float32)
        ran = c_u64(0)
        if x0 is not None:
            x0 = np.ascontiguousarray(x0, dtype=np.float32)
        if hashes is not None:
            hashes = np.ascontiguousarray(hashes, dtype=np.uint64)
        check(lib().cleora_multi_embed(self.handle, ptr(hashes), ptr(x0), kind, d, int(iterations), int(seed), float(residual_weight),
                                       float(threshold), int(flags), ptr(out), ctypes.byref(ran)))
        return out, int(ran.value)

    def propagate(self, kind, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.empty_like(x)
        check(lib().cleora_multi_propagate(self.handle, kind, ptr(x), x.shape[1], ptr(out)))
        return out

    def close(self):
        if self.handle:
            lib().cleora_multi_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DevArray:
    """A device buffer shaped like a numpy array (hipMalloc through the C ABI).  Lets hosts
    without torch keep matrices resident in HBM across calls."""

    def __init__(self, shape, dtype, _ptr=None):
        self.shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        if _ptr is not None:
            self.ptr = _ptr
            return
        p = vp()
        check(lib().cleora_malloc(self.nbytes, ctypes.byref(p)))
        self.ptr = p

    @classmethod
    def iterates(cls, graph, rows, d, count, iterations=0):
        """`count` (rows, d) f32 buffers placed for the SpMM of `graph` (cleora_alloc_iterates_for): every SpMM of the
        loop should read or write the FIRST one.  iterations: the SpMM launches about to run on them (0 = unknown: always
        search).  Returns (list of DevArray, (first-candidate ms, chosen ms))."""
        bufs = (vp * count)()
        ms = (ctypes.c_double * 2)()
        check(lib().cleora_alloc_iterates_for(graph.handle, int(d), int(count), int(iterations), bufs, ctypes.byref(ms)))
        return [cls((rows, d), np.float32, _ptr=vp(bufs[i])) for i in range(count)], (ms[0], ms[1])

    @classmethod
    def from_host(cls, a):
        a = np.ascontiguousarray(a)
        d = cls(a.shape, a.dtype)
        if d.nbytes:
            check(lib().cleora_memcpy_h2d(d.ptr, ptr(a), d.nbytes, None))
        return d

    def to_host(self):
        out = np.empty(self.shape, dtype=self.dtype)
        if self.nbytes:
            check(lib().cleora_memcpy_d2h(ptr(out), self.ptr, self.nbytes, None))
        return out

    def offset(self, nbytes):
        return vp(self.ptr.value + int(nbytes))

    @property
    def __cuda_array_interface__(self):
        """Lets torch.as_tensor(dev_array, device=...) view the buffer without a copy (the tensor keeps this object,
        and with it the allocation, alive)."""
        return {"shape": self.shape, "typestr": self.dtype.str, "data": (int(self.ptr.value or 0), False), "version": 2}

    def free(self):
        if self.ptr:
            lib().cleora_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
