"""Which (row, column) of a hub row differs from the oracle — diagnostic for hub_inorder_kernel."""
import sys
import numpy as np
sys.path.insert(0, ".")
import oracle
from cleora_amd import _hip
from tests.graphs import random_csr
from tests.test_gpu_spmm import run_dev
import tests.test_gpu_spmm as T
T.L = _hip.lib()
for d in (256, 64, 8):
    n = 4000
    hubs = [(17, 5000), (1234, 1025), (3999, 20000), (2000, 1024),
            (5, 1026), (6, 1027), (7, 1028), (8, 1152), (9, 1153), (10, 1151), (11, 1343), (12, 1344), (13, 1345), (14, 1088)]
    rowptr, col, vl, _ = random_csr(n, 8, seed=11, hubs=hubs)
    x = np.random.default_rng(12).standard_normal((n, d)).astype(np.float32)
    g = _hip.Graph.from_host(rowptr, col, vl)
    want = oracle.spmm(rowptr, col, vl, x)
    for rep in range(3):
        got = run_dev(g, _hip.LEFT, x)
        bad = np.argwhere(got != want)
        rows = np.unique(bad[:, 0])
        print("d", d, "rep", rep, "bad elements", len(bad), "rows", rows.tolist())
        for r in rows[:6]:
            cols = bad[bad[:, 0] == r][:, 1]
            print("   row", r, "len", int(rowptr[r + 1] - rowptr[r]), "n bad cols", len(cols), "cols", cols[:40].tolist())
