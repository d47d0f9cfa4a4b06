"""CPU-only: the C-ABI library loads and exports every symbol include/cleora_hip.h declares;
no compute call is made (there is no GPU here and no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

from cleora_amd import _hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "cleora_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cleora_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported_and_bound():
    L = _hip.lib()
    names = declared_functions()
    assert len(names) >= 30
    for name in names:
        assert hasattr(L, name), f"libcleora_hip.so does not export {name}"
    # the ctypes table covers the header one to one
    assert sorted(_hip.SIGNATURES) == names


def test_abi_version_and_error_channel():
    L = _hip.lib()
    assert L.cleora_abi_version() == _hip.ABI_VERSION == 5
    # argument validation happens before any device work
    n = ctypes.c_int(-1)
    assert L.cleora_device_count(ctypes.byref(n)) == _hip.OK and n.value >= 0
    assert L.cleora_graph_get_info(None, None) == _hip.E_INVALID
    assert "NULL" in _hip.last_error()


def test_comm_entry_points_validate_before_touching_rccl():
    """The multi-GPU section of the ABI (csrc/comm.hip): argument checks come first, and without a device the
    communicator cannot be created — there is no host-side emulation of the collectives."""
    L = _hip.lib()
    assert L.cleora_comm_create(None, 0, 1, 0, None) == _hip.E_INVALID
    h = ctypes.c_void_p()
    assert L.cleora_comm_create(None, 0, 1, 0, ctypes.byref(h)) == _hip.E_INVALID and "id" in _hip.last_error()
    ident = (ctypes.c_char * _hip.COMM_ID_BYTES)()
    assert L.cleora_comm_create(ctypes.cast(ident, ctypes.c_void_p), 2, 2, 0, ctypes.byref(h)) == _hip.E_INVALID
    assert L.cleora_allreduce_f32_dev(None, None, 4, None) == _hip.E_INVALID
    assert L.cleora_allgatherv_f32_dev(None, None, None, None) == _hip.E_INVALID
    assert L.cleora_comm_destroy(None) == _hip.OK
    if _hip.device_count() == 0:
        rc = L.cleora_comm_create(ctypes.cast(ident, ctypes.c_void_p), 0, 1, 0, ctypes.byref(h))
        assert rc in (_hip.E_RCCL, _hip.E_HIP, _hip.E_NODEVICE) and not h.value


@pytest.mark.skipif(_hip.device_count() > 0, reason="only meaningful without a GPU")
def test_no_cpu_fallback_without_gpu():
    """The product path fails loudly when there is no device."""
    rowptr = np.array([0, 1, 2], np.uint64)
    col = np.array([1, 0], np.uint32)
    val = np.array([1.0, 1.0], np.float32)
    with pytest.raises(RuntimeError, match="no HIP device|no CPU fallback"):
        _hip.Graph.from_host(rowptr, col, val)
    x = np.ones((2, 4), np.float32)
    with pytest.raises(RuntimeError):
        _hip.check(_hip.lib().cleora_l2_normalize(_hip.ptr(x), 2, 4, _hip.ptr(x.copy())))


def test_product_never_imports_oracle():
    """Nothing under cleora_amd/ may import, load or execute the oracle."""
    pkg = os.path.join(ROOT, "cleora_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", ".sh")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, f
                assert "libcleora_oracle" not in text, f


def test_workspace_queries_are_plain_arithmetic():
    """Workspace sizes are computed on the host (no device needed): the top-k workspace grows with the batch class
    (<= 8 queries: vector-unit form, <= 64 and beyond: matrix-core form with 64 / 256 queries per pass) and the generic
    query serves any batch."""
    L = _hip.lib()
    n, k = 1_000_000, 10
    w1, w8, w9, w64, w65, wany = (L.cleora_topk_workspace_for(n, k, q) for q in (1, 8, 9, 64, 65, 100_000))
    assert w1 == w8 < w9 == w64 < w65 == wany == L.cleora_topk_workspace(n, k)
    assert w8 >= 8 * n * 4 and w64 >= 64 * n * 4 and w65 >= 256 * n * 4
    assert L.cleora_whiten_workspace(n, 256) > 0 and L.cleora_gram_workspace(n, 256) > 0
