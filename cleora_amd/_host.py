"""ctypes binding of libcleora_host.so (include/cleora_host.h): entity hashing, graph
construction and the bincode pickle format — host C++, no GPU involved."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# CLEORA_HOST_LIB: another build of the same library (the sanitizer build of tests/test_host_sanitizers.py)
LIB_PATH = os.environ.get("CLEORA_HOST_LIB") or os.path.join(_HERE, "libcleora_host.so")

vp, c_u64, c_u32, c_int = ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int
u8p = ctypes.POINTER(ctypes.c_uint8)

SIGNATURES = {
    "cleora_host_last_error": (ctypes.c_char_p, []),
    "cleora_xxh64": (c_u64, [ctypes.c_char_p, c_u64, c_u64]),
    "cleora_host_build_from_lines": (c_int, [ctypes.c_char_p, vp, c_u64, ctypes.c_char_p, c_u32,
                                             ctypes.POINTER(vp)]),
    "cleora_host_build_from_files": (c_int, [ctypes.POINTER(ctypes.c_char_p), c_u64, ctypes.c_char_p,
                                             c_u32, ctypes.POINTER(vp)]),
    "cleora_host_free": (None, [vp]),
    "cleora_host_set_threads": (None, [c_u32]),
    "cleora_host_empty": (c_int, [ctypes.POINTER(vp)]),
    "cleora_host_sizes": (c_int, [vp, ctypes.POINTER(c_u64), ctypes.POINTER(c_u64), ctypes.POINTER(c_u64)]),
    "cleora_host_copy": (c_int, [vp, vp, vp, vp, vp, vp, vp, vp]),
    "cleora_host_copy_ids": (c_int, [vp, vp, vp]),
    "cleora_host_descriptor": (c_int, [vp, u8p, ctypes.POINTER(ctypes.c_char_p), u8p,
                                       ctypes.POINTER(ctypes.c_char_p)]),
    "cleora_host_set_ids": (c_int, [vp, ctypes.c_char_p, vp, c_u64]),
    "cleora_host_serialize": (c_int, [vp, ctypes.POINTER(u8p), ctypes.POINTER(c_u64)]),
    "cleora_host_deserialize": (c_int, [ctypes.c_char_p, c_u64, ctypes.POINTER(vp)]),
    "cleora_host_free_bytes": (None, [u8p]),
}

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run cleora_amd/csrc/build_host.sh "
                               "(or __graft_entry__.build())")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def last_error():
    return lib().cleora_host_last_error().decode("utf-8", "replace")


def xxh64(data: bytes, seed: int = 0) -> int:
    return lib().cleora_xxh64(data, len(data), seed)


def pack_strings(strings):
    """list[str] -> (utf-8 bytes, offsets u64[n+1])."""
    joined = "".join(strings)
    if joined.isascii():   # one encode instead of one per string: character counts are byte counts
        offsets = np.zeros(len(strings) + 1, dtype=np.uint64)
        if strings:
            offsets[1:] = np.cumsum(np.fromiter(map(len, strings), dtype=np.uint64, count=len(strings)))
        return joined.encode("ascii"), offsets
    enc = [s.encode("utf-8") for s in strings]
    offsets = np.zeros(len(enc) + 1, dtype=np.uint64)
    if enc:
        offsets[1:] = np.cumsum([len(e) for e in enc], dtype=np.uint64)
    return b"".join(enc), offsets


class HostGraph:
    """Owns a cleora_hostgraph handle and exposes its arrays as numpy copies."""

    def __init__(self, handle):
        self.handle = handle

    @classmethod
    def from_lines(cls, lines, columns, trim_n):
        data, offsets = pack_strings(lines)
        h = vp()
        rc = lib().cleora_host_build_from_lines(data, offsets.ctypes.data_as(vp), len(lines),
                                                columns.encode("utf-8"), trim_n, ctypes.byref(h))
        if rc != 0:
            raise ValueError(last_error())
        return cls(h)

    @classmethod
    def from_files(cls, paths, columns, trim_n):
        arr = (ctypes.c_char_p * len(paths))(*[p.encode("utf-8") for p in paths])
        h = vp()
        rc = lib().cleora_host_build_from_files(arr, len(paths), columns.encode("utf-8"), trim_n,
                                                ctypes.byref(h))
        if rc != 0:
            raise ValueError(last_error())
        return cls(h)

    @classmethod
    def empty(cls):
        h = vp()
        lib().cleora_host_empty(ctypes.byref(h))
        return cls(h)

    @classmethod
    def deserialize(cls, data: bytes):
        h = vp()
        if lib().cleora_host_deserialize(data, len(data), ctypes.byref(h)) != 0:
            raise RuntimeError(last_error())
        return cls(h)

    def serialize(self) -> bytes:
        p, n = u8p(), c_u64(0)
        if lib().cleora_host_serialize(self.handle, ctypes.byref(p), ctypes.byref(n)) != 0:
            raise RuntimeError(last_error())
        try:
            return ctypes.string_at(p, n.value)
        finally:
            lib().cleora_host_free_bytes(p)

    def sizes(self):
        n, nnz, b = c_u64(0), c_u64(0), c_u64(0)
        lib().cleora_host_sizes(self.handle, ctypes.byref(n), ctypes.byref(nnz), ctypes.byref(b))
        return n.value, nnz.value, b.value

    def arrays(self):
        n, nnz, _ = self.sizes()
        out = {
            "rowptr": np.zeros(n + 1, np.uint64), "col": np.zeros(nnz, np.uint32),
            "val_left": np.zeros(nnz, np.float32), "val_sym": np.zeros(nnz, np.float32),
            "row_sum": np.zeros(n, np.float32), "hashes": np.zeros(n, np.uint64),
            "column_ids": np.zeros(n, np.uint8),
        }
        p = lambda k: out[k].ctypes.data_as(vp)
        lib().cleora_host_copy(self.handle, p("rowptr"), p("col"), p("val_left"), p("val_sym"),
                               p("row_sum"), p("hashes"), p("column_ids"))
        return out

    def entity_ids(self):
        n, _, nbytes = self.sizes()
        buf = ctypes.create_string_buffer(max(nbytes, 1))
        offsets = np.zeros(n + 1, np.uint64)
        lib().cleora_host_copy_ids(self.handle, buf, offsets.ctypes.data_as(vp))
        raw = buf.raw
        off = offsets.astype(np.int64)
        return [raw[off[i]:off[i + 1]].decode("utf-8") for i in range(n)]

    def set_entity_ids(self, ids):
        data, offsets = pack_strings(ids)
        if lib().cleora_host_set_ids(self.handle, data, offsets.ctypes.data_as(vp), len(ids)) != 0:
            raise ValueError(last_error())

    def descriptor(self):
        a, b = ctypes.c_uint8(0), ctypes.c_uint8(0)
        an, bn = ctypes.c_char_p(), ctypes.c_char_p()
        lib().cleora_host_descriptor(self.handle, ctypes.byref(a), ctypes.byref(an), ctypes.byref(b),
                                     ctypes.byref(bn))
        return a.value, (an.value or b"").decode("utf-8"), b.value, (bn.value or b"").decode("utf-8")

    def close(self):
        if self.handle:
            lib().cleora_host_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
