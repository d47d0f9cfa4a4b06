"""GPU: `python bench.py --gpus 2` the way the driver starts a bench (no torchrun, WORLD_SIZE unset) on the one-GPU test box —
bench.py launches its own two ranks, which share the GPU over the library's peer-direct transport (RCCL refuses two ranks on one
device): the C-ABI communicator, the self-test under every all-gather algorithm, the row partition through
cleora_sharded_propagate_dev, the column partition, cleora_embed_sharded with CLEORA_F_WHITEN — and ONE JSON line with the fields
VERDICT round 3 (next #1c) asked for.  The numbers mean nothing (two ranks on one GPU); the path is what is tested."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_sharing_the_gpu_without_torchrun():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--nodes", "300000", "--pairs", "2850000",
                        "--steps", "3", "--warmup", "1", "--whiten-iters", "3"], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["unit"] == "edge*dim/s" and j["value"] > 0
    cfg = j["config"]
    assert cfg["partition"] == "row" and cfg["ranks"] == 2 and "csrc/sharded.hip" in cfg["parallelism"]
    assert "peer-direct" in cfg["collectives"] and cfg["allgather"] == "peer_direct"
    for key in ("spmm_kernels_max_rank", "allgather_on_comm_stream_max_rank", "exposed_beyond_spmm", "wall"):
        assert cfg["per_iteration_ms"][key] >= 0
    assert cfg["ceiling"]["links"] == 1 and cfg["ceiling"]["received_GB_per_rank_per_iteration"] > 0
    st = j["selftest"]
    assert st["row_bit_equal"] and st["row_max_abs_diff"] == 0.0 and st["algorithms_checked"] == ["peer_direct"] and st["column_max_abs_diff"] == 0.0 and st["column_bit_equal"]
    assert set(j["partitions"]) == {"row", "column"}
    for part in j["partitions"].values():
        assert part["checks"]["finite"] and part["checks"]["max_abs_row_norm_minus_1"] < 1e-5
    ws = j["whitened_sharded"]
    assert "error" not in ws and ws["iterations"] == 3 and ws["ms_per_iter"] > 0 and ws["max_abs_cov_minus_identity_all_rows"] < 5e-3


def test_bench_eight_ranks_sharing_the_gpu_the_drivers_scale_invocation():
    """`python bench.py --gpus 8` as the driver's SCALE run will start it (WORLD_SIZE unset), rehearsed with all eight ranks on the one GPU
    of the test box at a small size: the 8 x 4 block plan, the 8-way registration of the replicas, the self-test under the algorithm,
    eight graph generators taking turns, the launcher's teardown — and a line that is complete under the (d) rule for N > 1: `cpu_baseline`
    (rank 0's host cores on a row block of the same graph) and the `expected` note about N = 2.  Round 5's first rehearsal failed its
    self-test on three of eight ranks (a zero fill racing the peers' pushes) — a race two and three ranks never showed."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--share-gpu", "--nodes", "400000", "--pairs", "3800000",
                        "--steps", "2", "--warmup", "1", "--whiten-iters", "2", "--partition", "row"], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 8 and j["value"] > 0 and j["config"]["ranks"] == 8 and j["config"]["partition"] == "row"
    st = j["selftest"]
    assert st["row_bit_equal"] and st["row_max_abs_diff"] == 0.0        # (the column partition rides in the two-rank test above: --partition row here)
    assert j["cpu_baseline"]["value"] > 0 and j["cpu_baseline"]["kind"] == "port" and j["cpu_baseline"]["cores"] >= 1
    assert "N = 2" in j["expected"] or "P = 2" in j["expected"]
    assert j["config"]["ceiling"]["links"] == 7
    for part in j["partitions"].values():
        assert part["checks"]["finite"] and part["checks"]["max_abs_row_norm_minus_1"] < 1e-5
    assert "error" not in j["whitened_sharded"]
