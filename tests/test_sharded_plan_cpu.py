"""CPU: the host-side arithmetic of the row-partitioned loops under the C ABI (csrc/sharded.hip) — no GPU needed.

  * cleora_sharded_plan, the row boundaries of the world * steps blocks, against the Python model of the same partition
    (cleora_amd.model.row_bounds, the one the world-2 / world-4 gloo tests run against the oracle): identical boundaries,
    padded size and mode for permuted ids (equal rows), degree-ordered ids (balanced on the rowptr prefix sum), tiny graphs,
    empty graphs, more blocks than rows;
  * the memory plan of cleora_embed_sharded (missing #5): BASELINE config 4 with the DEFAULT (whitened) loop on
    8 GPUs must fit the 288 GB of an MI355X — two full replicas + the rank's own rows, not four replicas.
"""
import numpy as np
import pytest
import torch

from cleora_amd import _hip, sharded
from tests import sharded_model as model


def _rowptr(deg):
    rp = np.zeros(len(deg) + 1, dtype=np.uint64)
    rp[1:] = np.cumsum(deg).astype(np.uint64)
    return rp


@pytest.mark.parametrize("world,steps", [(1, 1), (2, 1), (2, 3), (4, 4), (8, 4), (3, 5)])
@pytest.mark.parametrize("kind", ["permuted", "ordered", "tiny", "empty_rows", "one_hub"])
@pytest.mark.parametrize("balance", ["auto", "rows", "nnz"])
def test_plan_equals_the_python_model(world, steps, kind, balance):
    rng = np.random.default_rng(world * 100 + steps)
    if kind == "permuted":
        deg = rng.zipf(1.8, 20_000).clip(max=5000)
    elif kind == "ordered":
        deg = np.sort(rng.zipf(1.6, 30_001).clip(max=20_000))[::-1].copy()         # hubs first: equal rows would be unbalanced
    elif kind == "tiny":
        deg = rng.integers(0, 4, 7)
    elif kind == "empty_rows":
        deg = np.where(rng.random(5000) < 0.5, 0, rng.integers(1, 9, 5000))
    else:
        deg = np.ones(9999, dtype=np.int64)
        deg[1234] = 400_000
    rp = _rowptr(deg)
    n = len(deg)
    want_bounds, want_pad, want_mode = model.row_bounds(n, torch.from_numpy(rp.view(np.int64)), world, steps, balance)
    got_bounds, got_pad, got_mode = sharded.plan_rows(n, rp, world, steps, balance)
    assert got_bounds == [int(b) for b in want_bounds]
    assert (got_pad, got_mode) == (want_pad, want_mode)
    # what the loops rely on: monotone, multiples of 4 rows, every real row covered
    assert all(b % 4 == 0 for b in got_bounds) and got_bounds == sorted(got_bounds) and got_bounds[-1] == got_pad >= n


def test_plan_of_an_empty_graph_and_argument_checks():
    assert sharded.plan_rows(0, np.zeros(1, np.uint64), 2, 2) == ([0, 4, 8, 12, 16], 16, "rows")
    L = _hip.lib()
    out = np.zeros(3, np.uint64)
    assert L.cleora_sharded_plan(4, None, 2, 1, 0, _hip.ptr(out), None, None) == _hip.E_INVALID
    assert L.cleora_sharded_plan(4, _hip.ptr(np.arange(5, dtype=np.uint64)), 0, 1, 0, _hip.ptr(out), None, None) == _hip.E_INVALID
    assert L.cleora_sharded_plan(4, _hip.ptr(np.arange(5, dtype=np.uint64)), 2, 1, 7, _hip.ptr(out), None, None) == _hip.E_INVALID


def test_config4_default_loop_fits_eight_gpus():
    """ogbn-papers100M (n ~ 111 M, nnz ~ 3.3 G after symmetrisation, d = 256), 8 ranks x 4 steps: what one rank of
    cleora_embed_sharded holds with CLEORA_F_WHITEN — the caller's replica, one more replica, Z for its own rows, the whitening
    workspace, its CSR blocks (u64 rowptr, u32 col, two f32 value streams) — against 288 GB.  Round 3's Python loop held FOUR
    replicas (455 GB)."""
    n, nnz, d, world, steps = 111_000_000, 3_300_000_000, 256, 8, 4
    block = -(-n // (world * steps))
    block = max(4, -(-block // 4) * 4)
    n_pad, local_rows = block * world * steps, block * steps
    L = _hip.lib()
    replica = n_pad * d * 4
    extra_whitened = int(L.cleora_embed_sharded_bytes(n_pad, local_rows, n, world, d, _hip.F_WHITEN))
    extra_plain = int(L.cleora_embed_sharded_bytes(n_pad, local_rows, n, world, d, 0))
    assert extra_plain == replica
    csr = (local_rows + steps) * 8 + (nnz // world) * (4 + 4 + 4 + 4)       # + the private col copy of the gather cache policy
    total = replica + extra_whitened + csr
    assert replica == pytest.approx(113.7e9, rel=0.01)
    assert extra_whitened - replica < 16e9                                    # Z (14.2 GB) + workspace, not another two replicas
    assert total < 0.9 * 288e9, total / 1e9
    assert replica * 4 > 288e9                                                # what round 3's plan would have needed
