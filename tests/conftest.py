import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a fresh checkout has no built artefacts (they are git-ignored): build them once
    need = [os.path.join(ROOT, "cleora_amd", "libcleora_hip.so"), os.path.join(ROOT, "cleora_amd", "libcleora_host.so")]
    if not all(os.path.exists(p) for p in need):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def _warm_page_cache(paths):
    """Read files once, sequentially, in the background: a freshly provisioned GPU box pages the 931 MB system
    librocsolver.so in through random faults when the torch-free C host first calls dsyevd (100-200 s measured for
    tests/test_zz_c_host.py::test_c_host_whitened_loop); a sequential read ahead of time is an order of magnitude faster.
    Test infrastructure only: nothing in the product depends on it."""
    for path in paths:
        try:
            with open(path, "rb", buffering=0) as f:
                while f.read(8 << 20):
                    pass
        except OSError:
            pass


def pytest_sessionstart(session):
    if "gpu" not in (session.config.getoption("-m") or "") or "not gpu" in (session.config.getoption("-m") or ""):
        return
    import glob
    import threading
    paths = [p for p in ("/opt/rocm/lib/librocsolver.so.0", "/opt/rocm/lib/librocblas.so.5") if os.path.exists(p)]
    paths += sorted(glob.glob("/opt/rocm/lib/rocblas/library/*gfx950*"))
    if paths:
        threading.Thread(target=_warm_page_cache, args=(paths,), daemon=True).start()
    # what `import torch` maps (the multi-process tests and bench.py import it in fresh processes: on a box with slow storage the
    # first of them took 233 s of a 560 s suite, 12 s on a box with fast storage) — a second reader, most needed first
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        tlib = os.path.join(os.path.dirname(spec.origin), "lib") if spec and spec.origin else None
    except Exception:                               # noqa: BLE001
        tlib = None
    if tlib and os.path.isdir(tlib):
        names = ("libtorch_cpu.so", "libtorch_hip.so", "libamdhip64.so", "libhsa-runtime64.so", "libc10.so", "libtorch_python.so", "librocblas.so",
                 "libhipblaslt.so", "libamd_comgr.so", "libMIOpen.so", "librocsolver.so", "librocsparse.so", "librocrand.so", "libmagma.so", "librccl.so")
        tpaths = [os.path.join(tlib, nm) for nm in names if os.path.exists(os.path.join(tlib, nm))]
        threading.Thread(target=_warm_page_cache, args=(tpaths,), daemon=True).start()
