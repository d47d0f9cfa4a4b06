"""GPU: the multi-GPU section of the C ABI (csrc/comm.hip, RCCL bound with dlopen) and the partitioned loops of
cleora_amd/sharded.py on the HIP backend.

  * world 1 over RCCL: the communicator, every collective entry point and the compute/communication stream
    ordering (cleora_stream_wait_stream) run for real on the one GPU of the test box;
  * world 2 with both ranks SHARING the GPU and torch.distributed/gloo as the communicator (RCCL refuses two
    ranks on one device): the N > 1 control flow of both partitions on the HIP kernels, against the oracle;
  * world 2 over RCCL on two GPUs when the box has them (skipped otherwise): equals the single-GPU result.
The collective logic itself is also covered on CPU (tests/test_sharded_cpu.py, gloo)."""
import ctypes
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from cleora_amd import _hip, comm as comm_mod, sharded
from tests import sharded_model as model
from tests.graphs import random_csr

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_rccl_world1_collectives_and_stream_ordering():
    dev = torch.device("cuda:0")
    c = comm_mod.RcclComm(comm_mod.RcclComm.unique_id(), 0, 1, 0)
    L = _hip.lib()
    r, w, d = ctypes.c_int(-1), ctypes.c_int(-1), ctypes.c_int(-1)
    _hip.check(L.cleora_comm_info(c.handle, ctypes.byref(r), ctypes.byref(w), ctypes.byref(d)))
    assert (r.value, w.value, d.value) == (0, 1, 0)
    x = torch.arange(12, dtype=torch.float32, device=dev).reshape(4, 3).contiguous()
    want = x.clone()
    with pytest.raises(ValueError, match="enable the peer transport"):
        c.set_allgather(_hip.ALLGATHER_PEER)        # the third algorithm needs cleora_comm_enable_peer first
    c.enable_peer()                                 # (a world of one: the mailbox only, nothing to map)
    c.register(x)
    for algo in (_hip.ALLGATHER_RING, _hip.ALLGATHER_P2P, _hip.ALLGATHER_PEER):
        c.set_allgather(algo)
        c.allgather_rows(x, [1, 4])                 # one shard: rows 1..3
        c.join()
    c.check()
    c.unregister(x)
    c.set_allgather(_hip.ALLGATHER_RING)
    c.allreduce_async(x)
    c.join()
    c.allreduce(x)
    c.broadcast(x, 0)
    x64 = x.double()
    c.allreduce(x64)
    y = torch.empty_like(x)
    c.alltoall(x, y)
    torch.cuda.synchronize()
    assert torch.equal(x, want) and torch.equal(y, want) and torch.equal(x64, want.double())
    with pytest.raises(ValueError):
        _hip.check(L.cleora_comm_set_allgather(c.handle, 7))
    # stream ordering: work on a second stream waits for the default stream and vice versa
    s = _hip.vp()
    _hip.check(L.cleora_stream_create(ctypes.byref(s)))
    a = torch.zeros(1 << 24, dtype=torch.float32, device=dev)
    a.add_(1.0)                                      # default-stream work ...
    cur = torch.cuda.current_stream().cuda_stream
    _hip.check(L.cleora_stream_wait_stream(s, cur))  # ... that the side stream must see finished
    b = torch.empty_like(a)
    _hip.check(L.cleora_memcpy_d2d(b.data_ptr(), a.data_ptr(), a.numel() * 4, s))
    _hip.check(L.cleora_stream_wait_stream(cur, s))
    assert float(b.sum()) == float(1 << 24)
    _hip.check(L.cleora_stream_destroy(s))
    c.close()


@pytest.mark.parametrize("steps,balance", [(1, "rows"), (3, "nnz")])
def test_row_partition_on_hip_backend_with_rccl_world1(steps, balance):
    """ShardedGraph driven through the C-ABI communicator (world 1): block decomposition, the forked
    communication stream and the join, bit-exact against the oracle."""
    dev = torch.device("cuda:0")
    n, d = 6001, 256
    rowptr, col, vl, vs = random_csr(n, 9, seed=51, empty_frac=0.03, hubs=[(17, 2200)])
    t = lambda a, dt: torch.from_numpy(a.view(dt) if a.dtype.kind == "u" else a).to(dev)
    c = comm_mod.RcclComm(comm_mod.RcclComm.unique_id(), 0, 1, 0)
    sg = model.ShardedGraph(n, t(rowptr, np.int64), t(col, np.int32), t(vl, None), t(vs, None), 0, 1, steps,
                              sharded.HipBackend(dev), comm=c, balance=balance)
    x0 = np.zeros((sg.n_pad, d), np.float32)
    x0[:n] = np.random.default_rng(52).standard_normal((n, d)).astype(np.float32)
    x, ran = model.embed_sharded(sg, 0, torch.from_numpy(x0).to(dev), 4, 0.25, 0.0)
    want, _ = oracle.embed(rowptr, col, vl, x0[:n], 4, residual_weight=0.25)
    np.testing.assert_allclose(x[:n].cpu().numpy(), want, rtol=0, atol=2e-6)
    c.close()


def _shared_gpu_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        n, d = 4003, 64
        rowptr, col, vl, vs = random_csr(n, 8, seed=61, empty_frac=0.04, hubs=[(9, 1500)])
        t = lambda a, dt: torch.from_numpy(a.view(dt) if a.dtype.kind == "u" else a).to(dev)
        be, cm = sharded.HipBackend(dev), comm_mod.TorchComm()
        x0 = np.random.default_rng(62).standard_normal((n, d)).astype(np.float32)
        sg = model.ShardedGraph(n, t(rowptr, np.int64), t(col, np.int32), t(vl, None), t(vs, None), rank, world, 2,
                                  be, comm=cm, balance="nnz")
        xp = np.zeros((sg.n_pad, d), np.float32)
        xp[:n] = x0
        xr, _ = model.embed_sharded(sg, 0, torch.from_numpy(xp).to(dev), 3)
        cg = sharded.ColumnShardedGraph(n, t(rowptr, np.int64), t(col, np.int32), t(vl, None), t(vs, None), d, rank,
                                        world, be, comm=cm, steps=2)
        xl, _ = sharded.embed_column_sharded(cg, 1, torch.from_numpy(np.ascontiguousarray(x0[:, cg.c0:cg.c0 + cg.dl])).to(dev), 3)
        q.put((rank, xr[:n].cpu().numpy(), cg.gather_columns(xl).cpu().numpy()))
    finally:
        dist.destroy_process_group()


def test_two_ranks_sharing_the_gpu_over_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_shared_gpu_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=900) for _ in range(world)]
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    n, d = 4003, 64
    rowptr, col, vl, vs = random_csr(n, 8, seed=61, empty_frac=0.04, hubs=[(9, 1500)])
    x0 = np.random.default_rng(62).standard_normal((n, d)).astype(np.float32)
    want_row, _ = oracle.embed(rowptr, col, vl, x0, 3)
    want_col, _ = oracle.embed(rowptr, col, vs, x0, 3)
    for _, xr, xc in got:
        np.testing.assert_array_equal(xr, want_row)                      # every row in the reference's order, hub row 9 included
        np.testing.assert_allclose(xc, want_col, rtol=0, atol=2e-6)      # column partition: the row norm is a sum of per-slice partials
    np.testing.assert_array_equal(got[0][1], got[1][1])                  # replicas identical
    np.testing.assert_array_equal(got[0][2], got[1][2])


def _rccl_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)      # launcher only
    try:
        dev = torch.device("cuda", rank)
        torch.cuda.set_device(dev)
        cm = comm_mod.RcclComm.from_torch_distributed(rank)
        n, d = 20011, 256
        rowptr, col, vl, vs = random_csr(n, 10, seed=71, empty_frac=0.02, hubs=[(5, 3000)])
        t = lambda a, dt: torch.from_numpy(a.view(dt) if a.dtype.kind == "u" else a).to(dev)
        be = sharded.HipBackend(dev)
        x0 = np.random.default_rng(72).standard_normal((n, d)).astype(np.float32)
        out = {}
        for algo in (_hip.ALLGATHER_RING, _hip.ALLGATHER_P2P):
            cm.set_allgather(algo)
            for balance in ("rows", "nnz"):
                sg = model.ShardedGraph(n, t(rowptr, np.int64), t(col, np.int32), t(vl, None), None, rank, world, 3,
                                          be, comm=cm, balance=balance)
                xp = np.zeros((sg.n_pad, d), np.float32)
                xp[:n] = x0
                xr, _ = model.embed_sharded(sg, 0, torch.from_numpy(xp).to(dev), 4, 0.2, 0.0)
                out[(algo, balance)] = xr[:n].cpu().numpy()
        cg = sharded.ColumnShardedGraph(n, t(rowptr, np.int64), t(col, np.int32), t(vl, None), None, d, rank, world,
                                        be, comm=cm, steps=3)
        xl, _ = sharded.embed_column_sharded(cg, 0, torch.from_numpy(np.ascontiguousarray(x0[:, cg.c0:cg.c0 + cg.dl])).to(dev), 4, 0.2)
        out["column"] = cg.gather_columns(xl).cpu().numpy()
        xw = np.zeros((cg.n_pad, cg.dl), np.float32)
        xw[:n] = x0[:, cg.c0:cg.c0 + cg.dl]
        xw2, _ = sharded.embed_column_sharded(cg, 0, torch.from_numpy(xw).to(dev), 2, whiten=True)
        out["column_whiten"] = cg.gather_columns(xw2)[:n].cpu().numpy()
        q.put((rank, out))
        cm.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: RCCL refuses two ranks on one device")
def test_two_gpus_over_rccl_equal_the_single_gpu_result():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rccl_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=900) for _ in range(world))
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    n, d = 20011, 256
    rowptr, col, vl, vs = random_csr(n, 10, seed=71, empty_frac=0.02, hubs=[(5, 3000)])
    x0 = np.random.default_rng(72).standard_normal((n, d)).astype(np.float32)
    g1 = _hip.Graph.from_host(rowptr, col, vl)
    out = np.empty((n, d), np.float32)
    _hip.check(_hip.lib().cleora_embed(g1.handle, None, _hip.ptr(x0), _hip.LEFT, d, 4, 0, 0.2, 0.0, 0, _hip.ptr(out), None))
    for key in got[0]:
        np.testing.assert_array_equal(got[0][key], got[1][key])          # replicas identical
    hub = np.zeros(n, bool)
    hub[5] = True
    for key, x in got[0].items():
        if key == "column":
            np.testing.assert_allclose(x, out, rtol=0, atol=3e-7 * 4)    # row norm = sum of per-slice partials
        elif key == "column_whiten":
            from oracle import whiten as ow
            want, _ = ow.embed_slow(lambda v: oracle.spmm(rowptr, col, vl, v), x0, 2, whiten=True)
            s = np.sign((x * want).sum(axis=0))
            assert np.abs(x * s - want).max() <= 2e-3 * np.abs(want).max()
        else:
            # the row partition runs the single-GPU kernel on row blocks: rows are bit-identical to the 1-GPU loop
            # (the hub row too: the same in-order kernel)
            np.testing.assert_array_equal(x, out)
