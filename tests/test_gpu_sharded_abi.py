"""GPU: the row-partitioned loops and the peer-direct transport THROUGH THE C ABI (csrc/sharded.hip, csrc/peer.hip) — what a
Rust host would call (missing #2 / next #4, #7).  No torch anywhere in this file: numpy + ctypes only.

  * world 1 (no communicator): cleora_embed_sharded equals the one-GPU cleora_embed bit for bit (plain loop, any block count /
    balance) and to the whitened loop's tolerances (reorganised and reference order);
  * world 2 and 3 with the ranks SHARING the one GPU of the test box over a LOCAL communicator (cleora_comm_create_local: hipIpc
    mappings; RCCL refuses two ranks on one device): the collectives themselves (peer-direct all-gather-v into registered
    buffers, all-reduce summed in rank order, broadcast), then the loops — replicas identical on every rank, the plain loop
    bit-equal to the one-GPU loop, the whitened loops against it and against the oracle's loop.
The reference has no counterpart (single process: src/embedding.rs:59-63); the loops' arithmetic is the reference's
(src/embedding.rs:106-188, pycleora/__init__.py:109-117) and is pinned through the one-GPU entry points and the oracle."""
import ctypes
import multiprocessing as mp
import os

import numpy as np
import pytest

import oracle
from cleora_amd import _hip, comm as comm_mod, sharded
from oracle import whiten as ow
from tests.graphs import random_csr

pytestmark = pytest.mark.gpu


def _graph(n, seed):
    return random_csr(n, 9, seed=seed, empty_frac=0.03, hubs=[(17, 2200)])


def _cosines(e, rows):
    u = e[rows].astype(np.float64)
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    return u @ u.T


def _one_gpu(rowptr, col, val, x0, d, iters, rw, thr, flags, val_sym=None, kind=_hip.LEFT):
    g = _hip.Graph.from_host(rowptr, col, val, val_sym)
    out = np.empty_like(x0)
    ran = ctypes.c_uint64(0)
    _hip.check(_hip.lib().cleora_embed(g.handle, None, _hip.ptr(x0), kind, d, iters, 0, rw, thr, flags, _hip.ptr(out), ctypes.byref(ran)))
    g.close()
    return out, ran.value


def _run_sharded(sg, x0, d, iters, rw, thr, flags, kind=_hip.LEFT):
    xp = np.zeros((sg.n_pad, d), np.float32)
    xp[: sg.n] = x0
    dx = _hip.DevArray.from_host(xp)
    ran = sg.embed(dx, kind, d, iters, rw, thr, flags)
    out = dx.to_host()
    dx.free()
    assert not out[sg.n:].any()                            # padding rows stay zero
    return out[: sg.n], ran


@pytest.mark.parametrize("steps,balance", [(1, "rows"), (3, "nnz"), (4, "auto")])
def test_world_of_one_equals_the_one_gpu_loops(steps, balance):
    n, d = 6001, 256
    rowptr, col, vl, vs = _graph(n, 51)
    x0 = np.random.default_rng(52).standard_normal((n, d)).astype(np.float32)
    sg = sharded.DeviceShardedGraph(n, rowptr, col, vl, vs, None, steps, balance)
    assert sg.world == 1 and sg.n_pad % 4 == 0 and sg.local_rows == sg.n_pad and sg.local_nnz == col.shape[0]
    # plain loop with residual and a convergence test: bit-equal, same stop iteration
    got, ran = _run_sharded(sg, x0, d, 12, 0.25, 2e-3, 0)
    want, wran = _one_gpu(rowptr, col, vl, x0, d, 12, 0.25, 2e-3, 0)
    assert ran == wran
    np.testing.assert_array_equal(got, want)
    rows = np.random.default_rng(1).choice(n, 400, replace=False)
    # the default loop: reorganised form (no threshold) and the reference's order (a threshold that is never met)
    for thr in (0.0, 1e-30):
        got, ran = _run_sharded(sg, x0, d, 4, 0.2, thr, _hip.F_WHITEN)
        want, _ = _one_gpu(rowptr, col, vl, x0, d, 4, 0.2, thr, _hip.F_WHITEN)
        assert ran == 4 and np.isfinite(got).all()
        assert np.abs(_cosines(got, rows) - _cosines(want, rows)).max() < 1e-4
        assert np.abs(np.cov(got.astype(np.float64).T) - np.eye(d)).max() < 5e-3
    # the symmetric Markov values (src/embedding.rs:7-10) through the partition: bit-equal as well
    got, _ = _run_sharded(sg, x0, d, 3, 0.0, 0.0, 0, kind=_hip.SYMMETRIC)
    want, _ = _one_gpu(rowptr, col, vl, x0, d, 3, 0.0, 0.0, 0, val_sym=vs, kind=_hip.SYMMETRIC)
    np.testing.assert_array_equal(got, want)
    # normalization="l1" with whitening (pycleora/__init__.py:947-950): not rotation invariant, so the reference's order on both sides
    got, _ = _run_sharded(sg, x0, d, 3, 0.0, 0.0, _hip.F_WHITEN | _hip.F_L1NORM)
    want, _ = _one_gpu(rowptr, col, vl, x0, d, 3, 0.0, 0.0, _hip.F_WHITEN | _hip.F_L1NORM)
    assert np.abs(_cosines(got, rows) - _cosines(want, rows)).max() < 1e-4
    sg.close()


def _shard_cuts(world):
    """Row boundaries of an all-gather-v test: rank r owns 148 (r + 1) rows."""
    return np.cumsum([0] + [4 * 37 * (r + 1) for r in range(world)]).astype(np.int64)


def _worker(rank, world, ident, q, cases):
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    try:
        L = _hip.lib()
        cm = comm_mod.RcclComm(ident, rank, world, 0, stream_fn=lambda: None, local=True)
        out = {}
        # ---- the collectives themselves ---------------------------------------------------------------------------------
        d = 24
        cuts = _shard_cuts(world)                                        # shards of different lengths
        rows = int(cuts[-1])
        buf = np.full((rows, d), -1.0, np.float32)
        buf[cuts[rank]:cuts[rank + 1]] = (np.arange(cuts[rank], cuts[rank + 1], dtype=np.float32)[:, None] * 100 + np.arange(d, dtype=np.float32)) + 0.5 * rank
        db = _hip.DevArray.from_host(buf)
        cm.register(db)
        off = np.asarray([c * d for c in cuts], dtype=np.uint64)
        for _ in range(3):                                               # repeated: sequence numbers, no stale flags
            _hip.check(L.cleora_allgatherv_f32_dev(cm.handle, db.ptr, _hip.ptr(off), None))
        _hip.check(L.cleora_stream_sync(None))
        out["gathered"] = db.to_host()
        v32 = _hip.DevArray.from_host(np.arange(1000, dtype=np.float32) * (rank + 1))
        v64 = _hip.DevArray.from_host((np.arange(70_000, dtype=np.float64) + 0.25) * (rank + 1))
        for _ in range(3):
            _hip.check(L.cleora_allreduce_f32_dev(cm.handle, v32.ptr, 1000, None))
        _hip.check(L.cleora_allreduce_f64_dev(cm.handle, v64.ptr, 70_000, None))
        bc = _hip.DevArray.from_host(np.full(5000, float(rank + 7), np.float32))
        _hip.check(L.cleora_broadcast_dev(cm.handle, bc.ptr, 5000 * 4, world - 1, None))
        _hip.check(L.cleora_stream_sync(None))
        cm.check()
        out["allreduce32"], out["allreduce64"], out["broadcast"] = v32.to_host(), v64.to_host(), bc.to_host()
        cm.unregister(db)
        # ---- the loops ---------------------------------------------------------------------------------------------------
        for name, (n, d, seed, steps, balance, iters, rw, thr, flags) in cases.items():
            rowptr, col, vl, vs = _graph(n, seed)
            x0 = np.random.default_rng(seed + 1).standard_normal((n, d)).astype(np.float32)
            sg = sharded.DeviceShardedGraph(n, rowptr, col, vl, vs, cm, steps, balance)
            res, ran = _run_sharded(sg, x0, d, iters, rw, thr, flags)
            out[name] = (res, ran, sg.balance, sg.local_rows)
            sg.close()
        cm.check()
        cm.close()
        q.put((rank, out, None))
    except BaseException as e:                       # noqa: BLE001 - reported to the parent
        import traceback
        q.put((rank, None, f"{type(e).__name__}: {e}\n{traceback.format_exc()}"))


CASES = {
    "plain": (6001, 256, 51, 2, "rows", 12, 0.25, 2e-3, 0),
    "plain_nnz": (5003, 64, 61, 3, "nnz", 5, 0.0, 0.0, 0),
    "whitened": (6001, 256, 51, 2, "auto", 4, 0.2, 0.0, _hip.F_WHITEN),
    "whitened_reference_order": (4003, 64, 71, 2, "nnz", 3, 0.0, 1e-30, _hip.F_WHITEN),
}


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_sharing_the_gpu_over_the_local_communicator(world):
    ident = comm_mod.RcclComm.unique_id(local=True)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, ident, q, CASES)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        got = {}
        for _ in range(world):
            rank, out, err = q.get(timeout=900)
            assert err is None, f"rank {rank}: {err}"
            got[rank] = out
    finally:
        for p in procs:
            p.join(timeout=120)
            if p.is_alive():
                p.kill()
    # ---- collectives
    for r in range(world):
        g = got[r]["gathered"]
        rows = g.shape[0]
        d = g.shape[1]
        owner = np.zeros(rows, np.int64)
        cuts = _shard_cuts(world)
        assert rows == cuts[-1]
        for k in range(world):
            owner[cuts[k]:cuts[k + 1]] = k
        want = np.arange(rows, dtype=np.float32)[:, None] * 100 + np.arange(d, dtype=np.float32) + 0.5 * owner[:, None].astype(np.float32)
        np.testing.assert_array_equal(g, want)
        s = sum(k + 1 for k in range(world))
        np.testing.assert_array_equal(got[r]["allreduce32"], got[0]["allreduce32"])          # bit-identical on every rank
        np.testing.assert_array_equal(got[r]["allreduce32"], np.arange(1000, dtype=np.float32) * (s * world * world))   # three in a row
        np.testing.assert_array_equal(got[r]["allreduce64"], (np.arange(70_000, dtype=np.float64) + 0.25) * s)
        np.testing.assert_array_equal(got[r]["broadcast"], np.full(5000, float(world - 1 + 7), np.float32))
    # ---- loops
    for name, (n, d, seed, steps, balance, iters, rw, thr, flags) in CASES.items():
        rowptr, col, vl, vs = _graph(n, seed)
        x0 = np.random.default_rng(seed + 1).standard_normal((n, d)).astype(np.float32)
        want, wran = _one_gpu(rowptr, col, vl, x0, d, iters, rw, thr, flags)
        for r in range(world):
            res, ran, _, _ = got[r][name]
            np.testing.assert_array_equal(res, got[0][name][0])                                # replicas identical
            assert ran == wran
        res = got[0][name][0]
        if not flags & _hip.F_WHITEN:
            np.testing.assert_array_equal(res, want)        # the one-GPU kernel on row blocks: bit-identical rows, same stop iteration
        else:
            rows = np.random.default_rng(1).choice(n, 400, replace=False)
            assert np.isfinite(res).all()
            assert np.abs(_cosines(res, rows) - _cosines(want, rows)).max() < 1e-4
            ref, _ = ow.embed_slow(lambda v: oracle.spmm(rowptr, col, vl, v), x0, iters, residual_weight=rw, whiten=True)
            assert np.abs(_cosines(res, rows) - _cosines(ref, rows)).max() < 1e-3
    assert got[0]["plain_nnz"][2] == "nnz" and got[0]["plain"][2] == "rows"
    assert sum(got[r]["plain_nnz"][3] for r in range(world)) >= 5003


def _peer_mode(cm):
    m = ctypes.c_int(-2)
    _hip.check(_hip.lib().cleora_comm_peer_mode(cm.handle, ctypes.byref(m)))
    return m.value


def _handshake_worker(rank, world, ident, q):
    """The peer transport's safety net on shared-GPU ranks: the self-test inside communicator creation, the
    injected PUSH failure -> every rank on the PULL form -> the loops still bit-equal; the injected failure of both forms -> the same
    error on every rank; cleora_embed_sharded's first-use check with one and two injected mismatches."""
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    try:
        L = _hip.lib()
        out = {}
        cm = comm_mod.RcclComm(ident[0], rank, world, 0, stream_fn=lambda: None, local=True)
        out["mode_after_create"] = _peer_mode(cm)                      # creation ran the self-test: PUSH works between ranks of one GPU
        n, d, seed = 5003, 64, 61
        rowptr, col, vl, vs = _graph(n, seed)
        x0 = np.random.default_rng(seed + 1).standard_normal((n, d)).astype(np.float32)
        # (1) first-use check of the loop with ONE injected mismatch: PUSH is abandoned for PULL, the loop runs, results unchanged
        sg = sharded.DeviceShardedGraph(n, rowptr, col, vl, vs, cm, 3, "nnz")
        _hip.check(L.cleora_sharded_debug_fail_first_gather(sg.handle, 1))
        res, ran = _run_sharded(sg, x0, d, 5, 0.0, 0.0, 0)
        out["loop_after_one_injected_mismatch"] = (res, ran, _peer_mode(cm))
        # (2) the PULL form in the whitened loop as well
        res, ran = _run_sharded(sg, x0, d, 3, 0.0, 0.0, _hip.F_WHITEN)
        out["whitened_on_pull"] = (res, ran)
        # (3) TWO injected mismatches: both forms "fail" -> an error, on every rank, instead of results
        _hip.check(L.cleora_sharded_debug_fail_first_gather(sg.handle, 2))
        xp = _hip.DevArray.from_host(np.zeros((sg.n_pad, d), np.float32))
        rc = L.cleora_embed_sharded(sg.handle, xp.ptr, _hip.LEFT, d, 2, 0.0, 0.0, 0, None)
        out["two_injected_mismatches"] = (rc, _hip.last_error())
        xp.free()
        sg.close()
        cm.close()
        # (4) a fresh communicator: the stand-alone self-test with the PUSH checker failing -> PULL everywhere; collectives still right
        cm = comm_mod.RcclComm(ident[1], rank, world, 0, stream_fn=lambda: None, local=True)
        _hip.check(L.cleora_comm_selftest(cm.handle, 1))
        out["mode_after_failed_push"] = _peer_mode(cm)
        cuts = _shard_cuts(world)
        rows, dd = int(cuts[-1]), 24
        buf = np.full((rows, dd), -1.0, np.float32)
        buf[cuts[rank]:cuts[rank + 1]] = np.arange(cuts[rank], cuts[rank + 1], dtype=np.float32)[:, None] * 10 + rank
        db = _hip.DevArray.from_host(buf)
        cm.register(db)
        off = np.asarray([c * dd for c in cuts], dtype=np.uint64)
        for _ in range(3):
            _hip.check(L.cleora_allgatherv_f32_dev(cm.handle, db.ptr, _hip.ptr(off), None))
        _hip.check(L.cleora_stream_sync(None))
        out["gathered_on_pull"] = db.to_host()
        cm.unregister(db)
        # (5) both checkers failing: the transport is refused, with the same verdict everywhere
        rc = L.cleora_comm_selftest(cm.handle, 3)
        out["selftest_both_fail"] = (rc, _hip.last_error())
        cm.close()
        q.put((rank, out, None))
    except BaseException as e:                       # noqa: BLE001 - reported to the parent
        import traceback
        q.put((rank, None, f"{type(e).__name__}: {e}\n{traceback.format_exc()}"))


@pytest.mark.parametrize("world", [2, 3])
def test_peer_transport_selftest_and_the_pull_fallback(world):
    ident = (comm_mod.RcclComm.unique_id(local=True), comm_mod.RcclComm.unique_id(local=True))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_handshake_worker, args=(r, world, ident, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        got = {}
        for _ in range(world):
            rank, out, err = q.get(timeout=600)
            assert err is None, f"rank {rank}: {err}"
            got[rank] = out
    finally:
        for p in procs:
            p.join(timeout=120)
            if p.is_alive():
                p.kill()
    n, d, seed = 5003, 64, 61
    rowptr, col, vl, vs = _graph(n, seed)
    x0 = np.random.default_rng(seed + 1).standard_normal((n, d)).astype(np.float32)
    want, wran = _one_gpu(rowptr, col, vl, x0, d, 5, 0.0, 0.0, 0)
    wantw, _ = _one_gpu(rowptr, col, vl, x0, d, 3, 0.0, 0.0, _hip.F_WHITEN)
    rows = np.random.default_rng(1).choice(n, 400, replace=False)
    cuts = _shard_cuts(world)
    owner = np.zeros(int(cuts[-1]), np.float32)
    for k in range(world):
        owner[cuts[k]:cuts[k + 1]] = k
    for r in range(world):
        o = got[r]
        assert o["mode_after_create"] == 0                                   # PUSH passed the handshake
        res, ran, mode = o["loop_after_one_injected_mismatch"]
        assert mode == 1 and ran == wran                                     # every rank fell back to PULL ...
        np.testing.assert_array_equal(res, want)                             # ... and the plain loop is still the one-GPU loop, bit for bit
        resw, ranw = o["whitened_on_pull"]
        assert ranw == 3 and np.abs(_cosines(resw, rows) - _cosines(wantw, rows)).max() < 1e-4
        rc, msg = o["two_injected_mismatches"]
        assert rc == _hip.E_RCCL and "stale rows" in msg
        assert o["mode_after_failed_push"] == 1
        np.testing.assert_array_equal(o["gathered_on_pull"], np.arange(int(cuts[-1]), dtype=np.float32)[:, None] * 10 + owner[:, None] + np.zeros((1, 24), np.float32))
        rc, msg = o["selftest_both_fail"]
        assert rc == _hip.E_RCCL and "not coherent" in msg


def _col_worker(rank, world, ident, q, d, cases):
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    try:
        cm = comm_mod.RcclComm(ident, rank, world, 0, stream_fn=lambda: None, local=True)
        n, seed = 6001, 51
        rowptr, col, vl, vs = _graph(n, seed)
        x0 = np.random.default_rng(seed + 1).standard_normal((n, d)).astype(np.float32)
        out = {}
        for name, (steps, kind, iters, rw, thr) in cases.items():
            cg = sharded.DeviceColShardedGraph(n, rowptr, col, vl, vs, d, cm, steps)
            assert (cg.world, cg.rank, cg.dl, cg.c0) == (world, rank, d // world, rank * (d // world))
            dx = _hip.DevArray.from_host(np.ascontiguousarray(x0[:, cg.c0:cg.c0 + cg.dl]))
            ran = cg.embed(dx, kind, iters, rw, thr)
            out[name] = (dx.to_host(), ran)
            dx.free()
            cg.close()
        cm.check()
        cm.close()
        q.put((rank, out, None))
    except BaseException as e:                       # noqa: BLE001 - reported to the parent
        import traceback
        q.put((rank, None, f"{type(e).__name__}: {e}\n{traceback.format_exc()}"))


COL_CASES = {
    "plain": (1, _hip.LEFT, 6, 0.0, 0.0),
    "blocks_residual_convergence": (3, _hip.LEFT, 12, 0.25, 2e-3),
    "symmetric": (2, _hip.SYMMETRIC, 4, 0.0, 0.0),
}


@pytest.mark.parametrize("world,d", [(2, 256), (3, 192), (4, 64)])
def test_column_partition_through_the_c_abi_is_bit_equal_to_one_gpu(world, d):
    """csrc/colsharded.hip on ranks sharing the GPU (local communicator): every rank owns d / P columns and the whole CSR; the rows' sums of
    squares travel from rank to rank in the reference's order (CLEORA_F_ROWSQ_CONT), so the plain loop — residual blend, convergence
    test, symmetric values, several row blocks — is the one-GPU loop BIT FOR BIT, column slice by column slice."""
    ident = comm_mod.RcclComm.unique_id(local=True)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_col_worker, args=(r, world, ident, q, d, COL_CASES)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        got = {}
        for _ in range(world):
            rank, out, err = q.get(timeout=600)
            assert err is None, f"rank {rank}: {err}"
            got[rank] = out
    finally:
        for p in procs:
            p.join(timeout=120)
            if p.is_alive():
                p.kill()
    n, seed = 6001, 51
    rowptr, col, vl, vs = _graph(n, seed)
    x0 = np.random.default_rng(seed + 1).standard_normal((n, d)).astype(np.float32)
    for name, (steps, kind, iters, rw, thr) in COL_CASES.items():
        want, wran = _one_gpu(rowptr, col, vl, x0, d, iters, rw, thr, 0, val_sym=vs, kind=kind)
        res = np.concatenate([got[r][name][0] for r in range(world)], axis=1)
        assert all(got[r][name][1] == wran for r in range(world)), (name, [got[r][name][1] for r in range(world)], wran)
        np.testing.assert_array_equal(res, want, err_msg=name)


def test_column_partition_world_of_one_and_argument_checks():
    n, d = 3001, 64
    rowptr, col, vl, vs = _graph(n, 7)
    x0 = np.random.default_rng(8).standard_normal((n, d)).astype(np.float32)
    cg = sharded.DeviceColShardedGraph(n, rowptr, col, vl, vs, d, None, 2)
    dx = _hip.DevArray.from_host(x0)
    ran = cg.embed(dx, _hip.LEFT, 5, 0.3, 0.0)
    want, _ = _one_gpu(rowptr, col, vl, x0, d, 5, 0.3, 0.0, 0)
    assert ran == 5
    np.testing.assert_array_equal(dx.to_host(), want)
    L = _hip.lib()
    dy = _hip.DevArray((n, d), np.float32)
    assert L.cleora_colsharded_propagate_dev(cg.handle, _hip.LEFT, dx.ptr, dy.ptr, _hip.F_L1NORM, 0.0, None, None) == _hip.E_INVALID
    assert L.cleora_colsharded_propagate_dev(cg.handle, _hip.LEFT, dx.ptr, dx.ptr, _hip.F_L2NORM, 0.0, None, None) == _hip.E_INVALID
    assert L.cleora_embed_colsharded(cg.handle, dx.ptr, _hip.LEFT, 2, 0.0, 0.0, _hip.F_WHITEN, None) == _hip.E_INVALID
    cg.close()
