"""GPU: the parity checks bench.py attaches to its line, at a size that runs in seconds — in particular the sampled form used when
the iterate does not fit a host-side oracle iteration (config 4's size), which must cover the longest (hub) rows too."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_sampled_row_check_covers_hub_rows_and_is_bit_equal():
    import torch
    import bench
    from cleora_amd import _hip, synth
    dev = torch.device("cuda:0")
    g = synth.power_law_graph(300_000, 2_850_000, 2, dev)
    n, nnz, d = g["n"], g["nnz"], 128
    graph = _hip.Graph.from_device(n, n, nnz, g["rowptr"].data_ptr(), g["col"].data_ptr(), g["val_left"].data_ptr(), None, 0, keepalive=g)
    info = graph.info()
    assert info.n_hub_rows > 0
    x = torch.randn((n, d), dtype=torch.float32, device=dev)
    y = torch.empty_like(x)
    L = _hip.lib()
    _hip.check(L.cleora_propagate_dev(graph.handle, _hip.LEFT, x.data_ptr(), d, d, y.data_ptr(), d, _hip.F_L2NORM, 0.0, None, None, None,
                                      torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    res = bench.sampled_row_check(g, x, y, n, d, info.hub_threshold, rows=512, hub_rows=8)
    assert res["sampled_rows_compared"] >= 512 and res["sampled_rows_bit_equal"] == res["sampled_rows_compared"]
    assert res["sampled_hub_rows_compared"] >= 1 and res["sampled_hub_rows_bit_equal"] == res["sampled_hub_rows_compared"]
    assert res["longest_row_edges"] == int(torch.diff(g["rowptr"]).max())
    # a corrupted row is seen
    y[int(torch.diff(g["rowptr"]).argmax())] += 1e-3
    bad = bench.sampled_row_check(g, x, y, n, d, info.hub_threshold, rows=512, hub_rows=8)
    assert bad["sampled_rows_bit_equal"] == bad["sampled_rows_compared"] - 1
    graph.close()
