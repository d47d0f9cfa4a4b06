"""CPU: the host logic around the one-process multi-device handle (csrc/multi.hip) — device-list plumbing of the drop-in and the
argument checks of the C ABI, which come before any device work.  (The partition itself runs in tests/test_gpu_multi.py.)"""
import ctypes

import numpy as np
import pytest

import cleora_amd
from cleora_amd import _hip
from cleora_amd import pycleora as mod


@pytest.fixture(autouse=True)
def _reset():
    yield
    mod.set_devices(None)


def test_device_list_from_install_and_environment(monkeypatch):
    monkeypatch.delenv("CLEORA_DEVICES", raising=False)
    monkeypatch.delenv("CLEORA_DEVICE", raising=False)
    assert mod._devices() == [0]
    monkeypatch.setenv("CLEORA_DEVICE", "3")
    assert mod._devices() == [3]
    monkeypatch.setenv("CLEORA_DEVICES", "0, 1,2 ,7")
    assert mod._devices() == [0, 1, 2, 7]
    cleora_amd.install(devices=range(4))                 # install(devices=...) wins over the environment
    assert mod._devices() == [0, 1, 2, 3]
    cleora_amd.install()                                 # a plain install() keeps what was configured
    assert mod._devices() == [0, 1, 2, 3]
    mod.set_devices(None)
    assert mod._devices() == [0, 1, 2, 7]
    # one device configured: no partition handle is ever built
    mod.set_devices([0])
    g = mod.SparseMatrix.from_iterator(iter(["a b", "b c"]), "complex::reflexive::x")
    assert g._multi() is None


def test_multi_entry_points_validate_before_touching_a_device():
    L = _hip.lib()
    rowptr = np.array([0, 1, 2], np.uint64)
    col = np.array([1, 0], np.uint32)
    val = np.array([1.0, 1.0], np.float32)
    h = ctypes.c_void_p()
    ids = (ctypes.c_int * 2)(0, 0)
    assert L.cleora_multi_create(None, 2, 2, 2, _hip.ptr(rowptr), _hip.ptr(col), _hip.ptr(val), None, 0, 0, ctypes.byref(h)) == _hip.E_INVALID
    assert L.cleora_multi_create(ids, 0, 2, 2, _hip.ptr(rowptr), _hip.ptr(col), _hip.ptr(val), None, 0, 0, ctypes.byref(h)) == _hip.E_INVALID
    assert L.cleora_multi_create(ids, 2, 2, 2, None, _hip.ptr(col), _hip.ptr(val), None, 0, 0, ctypes.byref(h)) == _hip.E_INVALID
    assert L.cleora_multi_create(ids, 2, 2, 2, _hip.ptr(rowptr), _hip.ptr(col), _hip.ptr(val), None, 0, 0, None) == _hip.E_INVALID
    # without a GPU the handle cannot be created: there is no host-side emulation of the partition
    if _hip.device_count() == 0:
        rc = L.cleora_multi_create(ids, 2, 2, 2, _hip.ptr(rowptr), _hip.ptr(col), _hip.ptr(val), None, 0, 0, ctypes.byref(h))
        assert rc == _hip.E_NODEVICE and "no HIP device" in _hip.last_error()
    assert L.cleora_multi_embed(None, None, None, 0, 8, 1, 0, 0.0, 0.0, 0, None, None) == _hip.E_INVALID
    assert L.cleora_multi_propagate(None, 0, None, 8, None) == _hip.E_INVALID
    assert L.cleora_multi_get_info(None, None) == _hip.E_INVALID
    assert L.cleora_multi_destroy(None) == _hip.OK
