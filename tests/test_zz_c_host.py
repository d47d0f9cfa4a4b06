"""(Named to run last.)  The C ABI from plain C: both headers compile as strict C99 and C++11, and examples/embed_file.c (no Python,
no torch, system HIP runtime) builds against the in-tree libraries.  Without a GPU the program must fail loudly
(exit code 3, "no CPU fallback"); on a GPU its output equals the drop-in's embed_fast bit for bit."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXAMPLE = os.path.join(ROOT, "examples", "embed_file")
LINES = ["a b", "b c", "c a d", "d e", "e a", "f a b c"]


def build_example():
    subprocess.check_call(["bash", os.path.join(ROOT, "examples", "build.sh")], stdout=subprocess.DEVNULL)


def run_example(args, env):
    """Runs the C host; a device-library failure that is about the ENVIRONMENT of a process without Python's ROCm
    libraries (exit code 3: no device visible to the system HIP runtime, rocSOLVER not loadable) skips the test
    with the program's message instead of failing it — wrong OUTPUT still fails."""
    r = subprocess.run([EXAMPLE] + args, env=env, capture_output=True, text=True)
    if r.returncode == 3:
        pytest.skip("C host could not use the device from a torch-free process: " + r.stderr.strip().splitlines()[-1])
    assert r.returncode == 0, r.stderr


def read_tsv(path):
    ids, rows = [], []
    for line in open(path):
        eid, vals = line.rstrip("\n").split("\t")
        ids.append(eid)
        rows.append(np.array(vals.split(" "), dtype=np.float32))
    return ids, np.stack(rows)


def test_headers_are_plain_c(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "cleora_hip.h"\n#include "cleora_host.h"\n'
                   "int main(void) { cleora_graph_info i; (void)i; return CLEORA_ABI_VERSION == 5 ? 0 : 1; }\n")
    inc = os.path.join(ROOT, "include")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", inc, "-c", str(src),
                           "-o", str(tmp_path / "t.o")])
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", inc, "-x", "c++", "-c",
                           str(src), "-o", str(tmp_path / "t2.o")])


def test_c_host_fails_loudly_without_a_gpu(tmp_path):
    from cleora_amd import _hip
    build_example()
    edges = tmp_path / "edges.tsv"
    edges.write_text("\n".join(LINES) + "\n")
    r = subprocess.run([EXAMPLE, "complex::bad::n", "8", "3", str(tmp_path / "o.tsv"), str(edges)], capture_output=True, text=True)
    assert r.returncode == 2 and "Unrecognized column field modifier" in r.stderr     # src/configuration.rs:51-56
    if _hip.device_count() > 0:
        pytest.skip("a GPU is present: the no-device behaviour cannot be observed")
    r = subprocess.run([EXAMPLE, "complex::reflexive::n", "8", "3", str(tmp_path / "o.tsv"), str(edges)], capture_output=True, text=True)
    assert r.returncode == 3 and "no CPU fallback" in r.stderr and "6 entities" in r.stderr
    assert not (tmp_path / "o.tsv").exists()


@pytest.mark.gpu
def test_c_host_matches_the_drop_in(tmp_path):
    from cleora_amd.pycleora import SparseMatrix
    build_example()
    edges = tmp_path / "edges.tsv"
    rng = np.random.default_rng(8)
    lines = LINES + [" ".join(f"n{int(v)}" for v in rng.integers(0, 60, rng.integers(2, 6))) for _ in range(300)]
    edges.write_text("\n".join(lines) + "\n")
    g = SparseMatrix.from_files([str(edges)], "complex::reflexive::n")
    env = {k: v for k, v in os.environ.items() if k != "CLEORA_ROCSOLVER"}      # the C host finds rocSOLVER by itself
    out = tmp_path / "o.tsv"
    run_example(["complex::reflexive::n", "32", "5", str(out), str(edges)], env)
    ids, got = read_tsv(out)
    assert ids == g.entity_ids
    np.testing.assert_array_equal(got, g.embed_fast(32, 5))
    run_example(["--symmetric", "complex::reflexive::n", "16", "3", str(out), str(edges)], env)
    np.testing.assert_array_equal(read_tsv(out)[1], g.embed_fast(16, 3, propagation="symmetric"))
    # one process, several devices (cleora_multi_*, csrc/multi.hip) from plain C: three shards on the one GPU, same bits
    run_example(["--devices", "0,0,0", "complex::reflexive::n", "32", "5", str(out), str(edges)], env)
    ids, got = read_tsv(out)
    assert ids == g.entity_ids
    np.testing.assert_array_equal(got, g.embed_fast(32, 5))


@pytest.mark.gpu
def test_c_host_whitened_loop(tmp_path):
    """The default (whitened) path of pycleora.embed() as ONE C call from a process that never loads Python's
    ROCm libraries: rocSOLVER comes from the system ROCm through the library's own dlopen.  Eigenvector signs are
    the solver's, so columns are sign-aligned; 1e-4 relative.
    Slow on a freshly provisioned box (200 s measured): the first dsyevd of the process pages in the 931 MB system
    librocsolver.so.  The intermediate iterations take the in-house Cholesky kernel here (no rocBLAS in this process);
    with rocSOLVER's potrf/trtri the same test took 556 s (docs/history.md §3.7)."""
    from cleora_amd import embed as dev_embed
    from cleora_amd.pycleora import SparseMatrix
    build_example()
    edges = tmp_path / "edges.tsv"
    rng = np.random.default_rng(9)
    lines = LINES + [" ".join(f"n{int(v)}" for v in rng.integers(0, 60, rng.integers(2, 6))) for _ in range(300)]
    edges.write_text("\n".join(lines) + "\n")
    g = SparseMatrix.from_files([str(edges)], "complex::reflexive::n")
    env = {k: v for k, v in os.environ.items() if k != "CLEORA_ROCSOLVER"}
    out = tmp_path / "o.tsv"
    run_example(["--whiten", "complex::reflexive::n", "8", "4", str(out), str(edges)], env)
    got, want = read_tsv(out)[1], dev_embed.embed(g, 8, 4)
    got = got * np.sign((got * want).sum(axis=0))
    assert np.abs(got - want).max() <= 1e-4 * np.abs(want).max()


SHARDED = os.path.join(ROOT, "examples", "sharded_embed")


def _graph_file(tmp_path, seed):
    edges = tmp_path / "edges.tsv"
    rng = np.random.default_rng(seed)
    lines = LINES + [" ".join(f"n{int(v)}" for v in rng.integers(0, 200, rng.integers(2, 6))) for _ in range(900)]
    edges.write_text("\n".join(lines) + "\n")
    return edges


@pytest.mark.gpu
def test_c_host_row_partitioned_loop_world1(tmp_path):
    """examples/sharded_embed.c: north_star's row-partitioned loop (row blocks, replicas of X, in-place all-gather over
    the C-ABI communicator, communication stream beside the compute stream) from plain C — one rank here, so RCCL,
    the streams and the block arithmetic run for real on the one GPU; equals embed_fast bit for bit."""
    from cleora_amd.pycleora import SparseMatrix
    build_example()
    edges = _graph_file(tmp_path, 10)
    g = SparseMatrix.from_files([str(edges)], "complex::reflexive::n")
    env = {k: v for k, v in os.environ.items() if k not in ("CLEORA_ROCSOLVER", "CLEORA_RCCL")}
    out = tmp_path / "o.tsv"
    r = subprocess.run([SHARDED, "0", "1", str(tmp_path / "id"), "complex::reflexive::n", "32", "5", str(out), str(edges)],
                       env=env, capture_output=True, text=True)
    if r.returncode == 3:
        pytest.skip("C host could not use the device / RCCL from a torch-free process: " + r.stderr.strip().splitlines()[-1])
    assert r.returncode == 0, r.stderr
    ids, got = read_tsv(out)
    assert ids == g.entity_ids
    np.testing.assert_array_equal(got, g.embed_fast(32, 5))


@pytest.mark.gpu
@pytest.mark.parametrize("whiten", [False, True])
def test_c_host_two_ranks_sharing_the_gpu_over_the_local_transport(tmp_path, whiten):
    """examples/sharded_embed --local: two torch-free C processes on the ONE GPU of the test box, the peer-direct (hipIpc)
    communicator, cleora_sharded_create + cleora_embed_sharded, the id through a file.  Plain loop: equals embed_fast bit for bit;
    --whiten: the default embed() loop, pairwise cosines against the one-GPU device loop."""
    from cleora_amd.pycleora import SparseMatrix
    build_example()
    edges = _graph_file(tmp_path, 12)
    g = SparseMatrix.from_files([str(edges)], "complex::reflexive::n")
    env = {k: v for k, v in os.environ.items() if k not in ("CLEORA_ROCSOLVER", "CLEORA_RCCL")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    out = tmp_path / "o.tsv"
    dim, iters = (16, 4) if whiten else (32, 5)
    args = ["--local"] + (["--whiten"] if whiten else [])
    procs = [subprocess.Popen([SHARDED] + args + [str(r), "2", str(tmp_path / "id"), "complex::reflexive::n", str(dim), str(iters), str(out), str(edges)],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    errs = []
    for p in procs:
        try:
            _, err = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            p.kill()
            _, err = p.communicate()
        errs.append(err)
    assert [p.returncode for p in procs] == [0, 0], errs
    assert "peer-direct" in errs[0]
    ids, got = read_tsv(out)
    assert ids == g.entity_ids
    if not whiten:
        np.testing.assert_array_equal(got, g.embed_fast(dim, iters))
    else:
        from cleora_amd import embed as dev_embed
        want = dev_embed.embed(g, dim, iters)
        cos = lambda e: (lambda u: u @ u.T)(e.astype(np.float64) / np.linalg.norm(e.astype(np.float64), axis=1, keepdims=True))
        assert np.abs(cos(got) - cos(want)).max() < 1e-3


@pytest.mark.gpu
def test_c_host_row_partitioned_loop_two_gpus(tmp_path):
    """Two processes, two GPUs, RCCL over the C ABI, the unique id through a file: equals the single-GPU result."""
    from cleora_amd import _hip
    from cleora_amd.pycleora import SparseMatrix
    if _hip.device_count() < 2:
        pytest.skip("needs two GPUs: RCCL refuses two ranks on one device")
    build_example()
    edges = _graph_file(tmp_path, 11)
    g = SparseMatrix.from_files([str(edges)], "complex::reflexive::n")
    env = {k: v for k, v in os.environ.items() if k not in ("CLEORA_ROCSOLVER", "CLEORA_RCCL")}
    out = tmp_path / "o.tsv"
    procs = [subprocess.Popen([SHARDED, str(r), "2", str(tmp_path / "id"), "complex::reflexive::n", "64", "6", str(out), str(edges)],
                              env=env, stderr=subprocess.PIPE, text=True) for r in range(2)]
    for p in procs:
        _, err = p.communicate(timeout=900)
        assert p.returncode == 0, err
    ids, got = read_tsv(out)
    assert ids == g.entity_ids
    np.testing.assert_array_equal(got, g.embed_fast(64, 6))
