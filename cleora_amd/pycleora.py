"""Drop-in for the reference's compiled module `pycleora.pycleora` (PyO3 class SparseMatrix,
src/lib.rs:84-476 + src/sparse_matrix.rs:56-66): same class name, methods, signatures, defaults,
return types and exceptions.  Graph construction runs in libcleora_host.so (C++), every
propagate / normalise / init / embed call runs on the MI355X through libcleora_hip.so.

    import cleora_amd; cleora_amd.install()      # sys.modules["pycleora.pycleora"] = this module
    import pycleora                               # the reference's Python package, unmodified

Differences from the reference, all deliberate:
  * from_iterator / from_files always produce the single-consumer result; `num_workers` is
    accepted and ignored (the reference's multi-worker f32 sums are order-nondeterministic).
  * non-contiguous / non-C-ordered float32 inputs are copied instead of panicking
    (src/embedding.rs:78 `as_slice().unwrap()`).
Every row — hub rows included — is summed in the reference's order (csrc/spmm.hip): the propagate and the
embed_fast loops are bit-identical to the reference's arithmetic as restated by oracle/.

Several GPUs: `cleora_amd.install(devices=[0, 1, ..., 7])` (or CLEORA_DEVICES=0,1,...,7) makes the SAME calls — embed_fast*,
left / symmetric_markov_propagate, and pycleora.embed() through cleora_amd.accelerate() — run the graph row-partitioned over
those devices inside this one process (csrc/multi.hip: one host thread per device, peer-direct all-gather of the iterate).
"""
import ctypes
import os
import threading

import numpy as np

from . import _hip, _host

_PROPAGATIONS = {"left": _hip.LEFT, "symmetric": _hip.SYMMETRIC}


_DEVICES = None     # set by cleora_amd.install(devices=[...]); None: CLEORA_DEVICES, else the single CLEORA_DEVICE


def set_devices(devices):
    """The devices later SparseMatrix calls run on: a list of HIP device indices (None: back to the environment)."""
    global _DEVICES
    _DEVICES = None if devices is None else [int(v) for v in devices]


_HUB_SEGMENTS = None    # set by cleora_amd.install(hub_segments=...); None: CLEORA_HUB_SEGMENTS


def set_hub_segments(on):
    """The long-row sum of the loops later SparseMatrix / embed() calls run: False (default) = every row in the reference's
    order, bit-equal (src/embedding.rs:76-83); True = CLEORA_F_HUB_SEGMENTS, the segmented sum (within 2e-6 of the sum of |terms|
    per row, not the reference's bits) — for graphs with a pathological hub (10^7 edges in one row: an in-order chain of as many
    dependent additions).  None: back to the environment."""
    global _HUB_SEGMENTS
    _HUB_SEGMENTS = None if on is None else bool(on)


def loop_flags():
    on = _HUB_SEGMENTS if _HUB_SEGMENTS is not None else os.environ.get("CLEORA_HUB_SEGMENTS", "").strip() not in ("", "0")
    return _hip.F_HUB_SEGMENTS if on else 0


def _devices():
    if _DEVICES is not None:
        return list(_DEVICES)
    env = os.environ.get("CLEORA_DEVICES", "").strip()
    if env:
        return [int(v) for v in env.split(",") if v.strip() != ""]
    return [_device_index()]


def _device_index():
    return int(os.environ.get("CLEORA_DEVICE", "0"))


def _as_f32_matrix(x, name="x"):
    if not isinstance(x, np.ndarray):
        raise TypeError(f"argument '{name}': expected a 2-D numpy.ndarray of float32")
    if x.dtype != np.float32 or x.ndim != 2:
        raise TypeError(f"argument '{name}': expected a 2-D numpy.ndarray of float32, "
                        f"got dtype={x.dtype}, ndim={x.ndim}")
    return np.ascontiguousarray(x)


class SparseMatrix:
    __slots__ = ("_host", "_arr", "_ids", "_dev_graph", "_multi_graph", "_lock", "_lookup", "__weakref__")

    # ---- construction -------------------------------------------------------------------------
    def __new__(cls, *args):
        if len(args) != 0:
            raise ValueError("SparseMatrix cannot be constructed directly. Use "
                             "SparseMatrix.from_files() or SparseMatrix.from_iterator().")
        self = object.__new__(cls)
        self._adopt(_host.HostGraph.empty())
        return self

    def _adopt(self, host):
        self._host = host
        self._arr = host.arrays()
        self._ids = None
        self._dev_graph = None
        self._multi_graph = None
        self._lock = threading.Lock()
        self._lookup = {}       # id -> index tables, built on first use

    @classmethod
    def _wrap(cls, host):
        self = object.__new__(cls)
        self._adopt(host)
        return self

    @staticmethod
    def from_iterator(hyperedges, columns, hyperedge_trim_n=16, num_workers=None):
        lines = []
        it = iter(hyperedges)
        while True:
            try:
                line = next(it)
            except StopIteration:
                break
            except Exception as e:  # src/lib.rs:113-115
                raise ValueError(f"Error reading iterator element: {e}")
            if not isinstance(line, str):
                raise ValueError("Iterator elements must be strings")
            try:
                line.encode("utf-8")
            except UnicodeEncodeError:
                raise ValueError("Iterator elements must be valid UTF-8")
            lines.append(line)
        return SparseMatrix._wrap(_host.HostGraph.from_lines(lines, columns, int(hyperedge_trim_n)))

    @staticmethod
    def from_files(filepaths, columns, hyperedge_trim_n=16, num_workers=None):
        filepaths = list(filepaths)
        if not filepaths:
            raise ValueError("At least one file path is required")
        for fp in filepaths:
            if not (fp.endswith(".tsv") or fp.endswith(".csv") or fp.endswith(".txt")):
                raise ValueError(f"Unsupported file format: {fp}. Supported: .tsv, .csv, .txt")
        return SparseMatrix._wrap(_host.HostGraph.from_files(filepaths, columns, int(hyperedge_trim_n)))

    # ---- device plumbing ----------------------------------------------------------------------
    def _graph(self):
        if self._dev_graph is None:
            a = self._arr
            self._dev_graph = _hip.Graph.from_host(a["rowptr"], a["col"], a["val_left"], a["val_sym"],
                                                   device=_devices()[0])
        return self._dev_graph

    def _multi(self):
        """The row partition over the configured devices (csrc/multi.hip), or None when one device is configured."""
        devs = _devices()
        if len(devs) < 2:
            return None
        if self._multi_graph is None or self._multi_graph.devices != devs:
            if self._multi_graph is not None:
                self._multi_graph.close()
            a = self._arr
            self._multi_graph = _hip.MultiGraph.from_host(devs, a["rowptr"], a["col"], a["val_left"], a["val_sym"])
        return self._multi_graph

    # ---- propagation (src/lib.rs:29-47, 86-102) --------------------------------------------------
    def _markov_propagate(self, x, kind):
        x = _as_f32_matrix(x)
        n = self.num_entities
        if x.shape[0] != n:
            raise ValueError(f"Embedding matrix has {x.shape[0]} rows but graph has {n} entities")
        d = x.shape[1]
        if n == 0 or d == 0:
            return np.zeros((n, d), np.float32)
        with self._lock:
            m = self._multi()
            if m is not None:                       # every device its own rows, its own PCIe link
                return m.propagate(kind, x)
        out = np.empty((n, d), np.float32)
        with self._lock:
            # the host-pointer entry point: device staging buffers live with the graph handle, both copies run
            # through the pinned pipeline of csrc/stager.hip (what a Rust host's FFI call would do)
            _hip.check(_hip.lib().cleora_propagate(self._graph().handle, kind, _hip.ptr(x), d, _hip.ptr(out)))
        return out

    def left_markov_propagate(self, x, num_workers=None):
        return self._markov_propagate(x, _hip.LEFT)

    def symmetric_markov_propagate(self, x, num_workers=None):
        return self._markov_propagate(x, _hip.SYMMETRIC)

    # ---- fused loops (src/lib.rs:320-412) ------------------------------------------------------------
    def _embed(self, feature_dim, iterations, propagation, seed, residual_weight, threshold):
        if propagation not in _PROPAGATIONS:
            raise ValueError(f"Unknown propagation '{propagation}'. Use 'left' or 'symmetric'.")
        n, d = self.num_entities, int(feature_dim)
        out = np.zeros((n, d), np.float32)
        if n == 0 or d == 0:
            return out, int(iterations)
        ran = ctypes.c_uint64(0)
        with self._lock:
            m = self._multi()
            if m is not None:
                return m.embed(self._arr["hashes"], None, _PROPAGATIONS[propagation], d, iterations, seed, residual_weight, threshold, loop_flags())
            g = self._graph()
            _hip.check(_hip.lib().cleora_embed(
                g.handle, _hip.ptr(self._arr["hashes"]), None, _PROPAGATIONS[propagation], d,
                int(iterations), int(seed), float(residual_weight), float(threshold), loop_flags(),
                _hip.ptr(out), ctypes.byref(ran)))
        return out, int(ran.value)

    def embed_fast(self, feature_dim, num_iterations, propagation="left", seed=0,
                   residual_weight=0.0, num_workers=None):
        return self._embed(feature_dim, num_iterations, propagation, seed, residual_weight, 0.0)[0]

    def embed_fast_convergence(self, feature_dim, max_iterations, propagation="left", seed=0,
                               residual_weight=0.0, convergence_threshold=0.0, num_workers=None):
        return self._embed(feature_dim, max_iterations, propagation, seed, residual_weight,
                           convergence_threshold)

    def l2_normalize(self, x, num_workers=None):
        x = _as_f32_matrix(x)
        out = np.empty_like(x)
        if x.size:
            _hip.check(_hip.lib().cleora_l2_normalize(_hip.ptr(x), x.shape[0], x.shape[1], _hip.ptr(out)))
        return out

    def initialize_deterministically(self, feature_dim, seed=0):
        n, d = self.num_entities, int(feature_dim)
        out = np.zeros((n, d), np.float32)
        if n and d:
            _hip.check(_hip.lib().cleora_init(_hip.ptr(self._arr["hashes"]), n, d, int(seed), _hip.ptr(out)))
        return out

    # ---- structure queries: host data, no kernel (src/lib.rs:175-318) -----------------------------------
    def to_sparse_csr(self, markov_type=None):
        mt = "left" if markov_type is None else markov_type
        if mt not in ("left", "symmetric"):
            raise ValueError(f"Unknown markov_type '{mt}'. Use 'left' or 'symmetric'.")
        a = self._arr
        n = self.num_entities
        rows = np.repeat(np.arange(n, dtype=np.uint32), np.diff(a["rowptr"].astype(np.int64)))
        vals = a["val_sym"] if mt == "symmetric" else a["val_left"]
        return rows, a["col"].copy(), vals.copy(), n, n

    def get_entity_column_mask(self, column_name):
        a_id, a_name, b_id, b_name = self._host.descriptor()
        by_name = {a_name: a_id, b_name: b_id}  # HashMap::from: the later pair wins on equal names
        if column_name not in by_name:
            raise ValueError(f"Column name '{column_name}' not found. Available: '{a_name}', '{b_name}'")
        return self._arr["column_ids"] == by_name[column_name]

    @property
    def entity_ids(self):
        if self._ids is None:
            self._ids = self._host.entity_ids()
        return list(self._ids)

    @entity_ids.setter
    def entity_ids(self, ids):
        ids = [str(s) for s in ids]
        self._host.set_entity_ids(ids)
        self._ids = ids
        self._lookup = {}
        self._arr["hashes"] = self._host.arrays()["hashes"]

    @property
    def entity_degrees(self):
        return self._arr["row_sum"].copy()

    @property
    def num_entities(self):
        return len(self._ids) if self._ids is not None else self._host.sizes()[0]

    @property
    def num_edges(self):
        return int(self._arr["col"].shape[0])

    def _ids_list(self):
        if self._ids is None:
            self._ids = self._host.entity_ids()
        return self._ids

    def get_entity_index(self, entity_id):
        """Position of the FIRST entity with this id (`iter().position`, src/lib.rs:216-224); the lookup table is
        built once per id list instead of scanning n strings per call."""
        first = self._lookup.get("first")
        if first is None:
            first = {}
            for i, e in enumerate(self._ids_list()):
                first.setdefault(e, i)
            self._lookup["first"] = first
        try:
            return first[entity_id]
        except (KeyError, TypeError):
            raise ValueError(f"Entity '{entity_id}' not found")

    def get_entity_indices(self, entity_ids):
        """HashMap collected from (id, index) pairs: with duplicate ids the LAST index wins (src/lib.rs:226-240)."""
        index = self._lookup.get("last")
        if index is None:
            index = {e: i for i, e in enumerate(self._ids_list())}
            self._lookup["last"] = index
        out = []
        for e in entity_ids:
            if e not in index:
                raise ValueError(f"Entity '{e}' not found")
            out.append(index[e])
        return out

    def get_neighbors(self, entity_id):
        i = self.get_entity_index(entity_id)
        a = self._arr
        b, e = int(a["rowptr"][i]), int(a["rowptr"][i + 1])
        ids = self.entity_ids
        return [(ids[int(c)], float(v)) for c, v in zip(a["col"][b:e], a["val_left"][b:e])]

    # ---- dunder (src/lib.rs:426-475) --------------------------------------------------------------------
    def __repr__(self):
        _, a_name, _, b_name = self._host.descriptor()
        return (f"SparseMatrix(entities={self.num_entities}, edges={self.num_edges}, "
                f"columns=('{a_name}', '{b_name}'))")

    def __len__(self):
        return self.num_entities

    def __getstate__(self):
        return self._host.serialize()

    def __setstate__(self, state):
        if not isinstance(state, (bytes, bytearray)):
            raise TypeError("state must be bytes")
        self._adopt(_host.HostGraph.deserialize(bytes(state)))

    def __reduce__(self):
        return (_reconstruct, (type(self),), self.__getstate__())


def _reconstruct(cls):
    return cls.__new__(cls)
