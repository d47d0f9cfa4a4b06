"""cleora_amd — MI355X-native implementation of pycleora's Markov-propagation hot path.

  cleora_amd.pycleora.SparseMatrix   drop-in for pycleora.pycleora.SparseMatrix (src/lib.rs)
  cleora_amd.embed                   device-resident embed()/whitening loop (pycleora/__init__.py:51-164)
  cleora_amd.variants                device-resident embed variants + predict_links (pycleora/__init__.py:206-410, 636-681, 784-852)
  cleora_amd.sharded                 column- / row-partitioned multi-GPU propagation (one process per GPU)
  cleora_amd._hip                    ctypes binding of libcleora_hip.so (include/cleora_hip.h)
"""
__version__ = "0.1.0"


def install(devices=None, hub_segments=None):
    """Make this package's SparseMatrix the reference's compiled module: after this call
    `import pycleora` (the reference's unmodified Python package) binds
    `from .pycleora import SparseMatrix` (pycleora/__init__.py:4) to cleora_amd.pycleora.

    devices: HIP device indices, e.g. range(8) — the graph is then row-partitioned over them INSIDE this process for
    embed_fast*, left / symmetric_markov_propagate and (after accelerate()) pycleora.embed(); None keeps what
    CLEORA_DEVICES / CLEORA_DEVICE say (default: device 0).
    hub_segments: True = the loops sum rows of more than 256 edges in segments (CLEORA_F_HUB_SEGMENTS: not the reference's bits,
    within 2e-6 of the sum of |terms|) — the way out for a graph whose longest row is an in-order chain of 10^7 additions; None keeps
    what CLEORA_HUB_SEGMENTS says (default: the reference's order)."""
    import importlib.util
    import sys

    from . import pycleora as _mod
    if devices is not None:
        _mod.set_devices(devices)
    if hub_segments is not None:
        _mod.set_hub_segments(hub_segments)
    sys.modules["pycleora.pycleora"] = _mod
    # pickles name the class by module path; use the reference's (src/sparse_matrix.rs:56) when its
    # Python package is importable so pickles interchange with real pycleora, else keep ours
    if "pycleora" in sys.modules or importlib.util.find_spec("pycleora") is not None:
        _mod.SparseMatrix.__module__ = "pycleora.pycleora"
    return _mod


def accelerate(package=None):
    """Optional second step after install(): rebind the reference package's own hot-path entry
    points to the device-resident versions, so that its CLI / benchmark / CleoraEmbedder (which all
    call `pycleora.embed`, pycleora/cli.py:148, __init__.py:885) keep the iterate in HBM instead of
    crossing PCIe twice per iteration and whitening in numpy:
        pycleora.embed              -> cleora_amd.embed.embed           (same signature)
        pycleora.whiten_embeddings  -> cleora_amd.embed.whiten_embeddings
        pycleora.embed_multiscale / embed_weighted / embed_directed / embed_with_attention /
        embed_edge_features / predict_links -> cleora_amd.variants.* (same signatures)
    The one normalisation the device path does not run — 'spectral', a full SVD per iteration
    (pycleora/__init__.py:951-956) — is forwarded to the original."""
    import importlib

    from . import embed as _dev
    pkg = package if package is not None else importlib.import_module("pycleora")
    original_embed = pkg.embed

    def embed(graph, *args, **kwargs):
        norm = kwargs.get("normalization", args[3] if len(args) > 3 else "l2")
        if norm not in ("l2", "l1", "none") or not isinstance(graph, _dev.SparseMatrix):
            return original_embed(graph, *args, **kwargs)
        return _dev.embed(graph, *args, **kwargs)

    embed.__wrapped__ = original_embed
    pkg.embed = embed
    pkg.whiten_embeddings = _dev.whiten_embeddings

    # the embed variants and link prediction: same signatures, device-resident loop
    import inspect

    from . import variants as _var

    def rebind(name):
        original = getattr(pkg, name, None)
        device_fn = getattr(_var, name)
        if original is None:
            return
        sig = inspect.signature(device_fn)

        def wrapper(*args, **kwargs):
            try:
                bound = sig.bind(*args, **kwargs).arguments
            except TypeError:
                return original(*args, **kwargs)
            graph = bound.get("graph")
            if bound.get("normalization", "l2") not in ("l2", "l1", "none") or \
                    (graph is not None and not isinstance(graph, _dev.SparseMatrix)):
                return original(*args, **kwargs)
            return device_fn(*args, **kwargs)

        wrapper.__wrapped__ = original
        wrapper.__name__ = name
        setattr(pkg, name, wrapper)

    for name in ("embed_multiscale", "embed_weighted", "embed_directed", "embed_with_attention",
                 "embed_edge_features", "predict_links"):
        rebind(name)
    return pkg
