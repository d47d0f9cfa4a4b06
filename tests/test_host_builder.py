"""CPU-only: the C++ host builder (libcleora_host.so) against the pure-Python builder oracle and
the reference's golden vectors; the drop-in class's host-side behaviour (no GPU needed)."""
import os
import pickle

import numpy as np
import pytest
import xxhash

import oracle
from cleora_amd import _host
from cleora_amd.pycleora import SparseMatrix
from oracle import refgraph, stdrng


def same_graph(h, g):
    a = h.arrays()
    assert h.entity_ids() == list(g.entity_ids)
    np.testing.assert_array_equal(a["hashes"], g.entity_hashes)
    np.testing.assert_array_equal(a["column_ids"], g.column_ids)
    np.testing.assert_array_equal(a["row_sum"], g.row_sum)
    np.testing.assert_array_equal(a["rowptr"], g.rowptr)
    np.testing.assert_array_equal(a["col"], g.col)
    np.testing.assert_array_equal(a["val_left"], g.val_left)      # bit-exact f32
    np.testing.assert_array_equal(a["val_sym"], g.val_sym)


def test_xxh64_matches_python_xxhash():
    rng = np.random.default_rng(1)
    for n in list(range(0, 70)) + [255, 1024, 5000]:
        data = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert _host.xxh64(data) == xxhash.xxh64_intdigest(data, seed=0) == oracle.xxh64(data)
    assert _host.xxh64(b"alice") == 8332761332120969289


@pytest.mark.parametrize("kind", ["reflexive", "complex"])
def test_snapshot_fixtures_through_cpp_builder(kind, golden_dir):
    """The reference's insta snapshots, with the graph built by the C++ builder and the SpMM by
    the oracle: entity order, weights and normalisation all have to be right."""
    lines, columns, emb = stdrng.snapshot_fixture(kind)
    h = _host.HostGraph.from_lines(lines, columns, 16)
    same_graph(h, refgraph.build_graph(lines, columns, 16))
    snaps = np.load(os.path.join(golden_dir, "snapshot_markov.npz"))
    a = h.arrays()
    tag = "01" if kind == "reflexive" else "02"
    for name, val in ((f"left_{tag}", a["val_left"]), (f"sym_{tag}", a["val_sym"])):
        y = oracle.spmm(a["rowptr"], a["col"], val, emb)
        got = np.trunc(y * np.float32(1000.0)).astype(np.int64)
        assert np.abs(got - snaps[name]).max() <= 1


def random_lines(rng, n_lines, shape):
    vocab = [f"e{i}" for i in range(60)] + ["ü", "日本", "x y".replace(" ", "_")]
    lines = []
    for _ in range(n_lines):
        if shape == "reflexive":
            k = int(rng.integers(1, 7))
            lines.append(" ".join(rng.choice(vocab, k)))
        elif shape == "complex_complex":
            ka, kb = int(rng.integers(1, 5)), int(rng.integers(1, 5))
            lines.append(" ".join(rng.choice(vocab, ka)) + "\t" + " ".join(rng.choice(vocab, kb)))
        elif shape == "plain_csv":
            lines.append(f" {rng.choice(vocab)} , {rng.choice(vocab)} ")
        elif shape == "plain_complex":
            kb = int(rng.integers(1, 6))
            lines.append(f"{rng.choice(vocab)}\t" + " ".join(rng.choice(vocab, kb)))
    return lines


@pytest.mark.parametrize("shape,columns", [
    ("reflexive", "complex::reflexive::item"), ("reflexive", "REFLEXIVE::Complex::item"),
    ("complex_complex", "complex::a complex::b"), ("plain_csv", "user product"),
    ("plain_complex", "user complex::basket")])
def test_random_hypergraphs_match_oracle(shape, columns):
    rng = np.random.default_rng(hash(shape) % 1000)
    lines = random_lines(rng, 400, shape)
    lines += ["", "   ", "a\tb\tc", "5 5", "dup dup dup", " padded　"]  # degenerate inputs
    same_graph(_host.HostGraph.from_lines(lines, columns, 16), refgraph.build_graph(lines, columns, 16))


def test_hyperedge_trim_without_ties():
    # occurrences are made strictly different before the long line arrives, so the high/low
    # partition is unambiguous (ties are implementation-defined in the reference)
    lines = []
    for i in range(12):
        lines += [f"n{i} pad{i}_{j}" for j in range(i + 1)]
    lines.append(" ".join(f"n{i}" for i in range(12)))
    for trim in (4, 8, 16):
        same_graph(_host.HostGraph.from_lines(lines, "complex::reflexive::x", trim),
                   refgraph.build_graph(lines, "complex::reflexive::x", trim))


def test_bad_column_specs():
    for spec, msg in (("reflexive::a", "REFLEXIVE but NOT COMPLEX"), ("foo::a b", "Unrecognized column field modifier"),
                      ("a b c", "More than one relation"), ("a", "More than one relation")):
        with pytest.raises(ValueError, match=msg):
            SparseMatrix.from_iterator(iter(["x y"]), spec)


def test_drop_in_host_surface(tmp_path, golden_dir):
    k = np.load(os.path.join(golden_dir, "karate_ref.npz"))
    lines = [str(s) for s in k["edges"]]
    g = SparseMatrix.from_iterator(iter(lines), str(k["columns"]))
    assert repr(g) == "SparseMatrix(entities=34, edges=190, columns=('member', 'member'))"
    assert len(g) == g.num_entities == 34 and g.num_edges == 190
    assert g.entity_ids == [str(s) for s in k["entity_ids"]]
    np.testing.assert_array_equal(g.entity_degrees, k["row_sum"])
    rows, cols, vals, n1, n2 = g.to_sparse_csr()
    assert (n1, n2) == (34, 34) and rows.dtype == np.uint32 and vals.dtype == np.float32
    np.testing.assert_array_equal(vals, k["val_left"])
    np.testing.assert_array_equal(g.to_sparse_csr("symmetric")[2], k["val_sym"])
    with pytest.raises(ValueError, match="Unknown markov_type"):
        g.to_sparse_csr("right")
    assert g.get_entity_index("0") == 0 and g.get_entity_indices(["1", "0"]) == [1, 0]
    with pytest.raises(ValueError, match="Entity 'zz' not found"):
        g.get_entity_index("zz")
    nb = dict(g.get_neighbors("0"))
    assert nb["0"] == 0.5 and len(nb) == 17
    with pytest.raises(ValueError, match="not found"):
        g.get_entity_indices(["0", "zz"])
    # duplicate ids after the setter: position() finds the first, the collected HashMap keeps the last
    # (src/lib.rs:216-240); the cached lookup tables are rebuilt when the ids change
    dup = SparseMatrix.from_iterator(iter(["a b", "b c"]), "complex::reflexive::n")
    assert dup.get_entity_index("c") == 2
    dup.entity_ids = ["x", "y", "x"]
    assert dup.get_entity_index("x") == 0 and dup.get_entity_indices(["x", "y"]) == [2, 1]
    with pytest.raises(ValueError, match="Entity 'c' not found"):
        dup.get_entity_index("c")
    # reference quirk kept: for a reflexive column both descriptor names are equal and
    # HashMap::from keeps the LAST pair (col_b_id = 1), while every column_id is 0 (src/lib.rs:180-196)
    assert not g.get_entity_column_mask("member").any()
    g_two = SparseMatrix.from_iterator(iter(["u1\tp1 p2", "u2\tp1"]), "user complex::product")
    np.testing.assert_array_equal(g_two.get_entity_column_mask("user"), [True, False, False, True])
    np.testing.assert_array_equal(g_two.get_entity_column_mask("product"), [False, True, True, False])
    with pytest.raises(ValueError, match="Column name 'nope' not found"):
        g.get_entity_column_mask("nope")
    # pickle = bincode bytes of struct SparseMatrix, round trip
    state = g.__getstate__()
    assert isinstance(state, bytes) and state[:1] == b"\x00" and b"member" in state[:32]
    g2 = pickle.loads(pickle.dumps(g))
    assert g2.entity_ids == g.entity_ids and g2.__getstate__() == state
    with pytest.raises(RuntimeError, match="Deserialization failed"):
        SparseMatrix().__setstate__(state[:-3])
    # constructors
    assert len(SparseMatrix()) == 0
    with pytest.raises(ValueError, match="cannot be constructed directly"):
        SparseMatrix(1)
    with pytest.raises(ValueError, match="Iterator elements must be strings"):
        SparseMatrix.from_iterator(iter(["a b", 3]), "complex::reflexive::m")
    with pytest.raises(ValueError, match="At least one file path"):
        SparseMatrix.from_files([], "a b")
    with pytest.raises(ValueError, match="Unsupported file format"):
        SparseMatrix.from_files(["x.json"], "a b")
    p = tmp_path / "edges.tsv"
    p.write_text("\n".join(lines) + "\n\n")
    g3 = SparseMatrix.from_files([str(p)], str(k["columns"]))
    assert g3.__getstate__() == state
    # shape check comes before any device work (src/lib.rs:36-43)
    with pytest.raises(ValueError, match="Embedding matrix has 3 rows but graph has 34 entities"):
        g.left_markov_propagate(np.zeros((3, 4), np.float32))
    # entity_ids setter re-keys the deterministic init (src/lib.rs:75 hashes the current ids)
    g3.entity_ids = [f"id{i}" for i in range(34)]
    assert g3._arr["hashes"][0] == xxhash.xxh64_intdigest(b"id0", seed=0)


def test_install_shim():
    import sys
    import types
    import cleora_amd
    saved = sys.modules.get("pycleora.pycleora")
    saved_parent = sys.modules.get("pycleora")
    if saved_parent is None:  # stand-in for the reference's Python package being installed
        sys.modules["pycleora"] = types.ModuleType("pycleora")
    try:
        mod = cleora_amd.install()
        assert sys.modules["pycleora.pycleora"] is mod
        assert mod.SparseMatrix.__module__ == "pycleora.pycleora"
        g = mod.SparseMatrix.from_iterator(iter(["a b", "b c"]), "complex::reflexive::n")
        assert pickle.loads(pickle.dumps(g)).entity_ids == ["a", "b", "c"]
    finally:
        mod.SparseMatrix.__module__ = "cleora_amd.pycleora"
        if saved_parent is None:
            sys.modules.pop("pycleora", None)
        if saved is None:
            sys.modules.pop("pycleora.pycleora", None)
        else:
            sys.modules["pycleora.pycleora"] = saved


def test_accelerate_rebinds_reference_entry_points():
    """cleora_amd.accelerate() on a stand-in package object: l2 / l1 / none go to the device loop, 'spectral'
    and foreign graph types are forwarded to the original function."""
    import types
    import cleora_amd
    from cleora_amd import embed as dev
    calls = []
    pkg = types.SimpleNamespace(embed=lambda g, *a, **k: calls.append(("orig", k.get("normalization"))) or "orig",
                                whiten_embeddings=lambda x: x)
    cleora_amd.accelerate(pkg)
    assert pkg.whiten_embeddings is dev.whiten_embeddings and pkg.embed.__wrapped__ is not None
    assert pkg.embed(object(), 8, 2) == "orig"                                  # not our SparseMatrix
    g = SparseMatrix.from_iterator(iter(["a b", "b c"]), "complex::reflexive::n")
    assert pkg.embed(g, 8, 2, normalization="spectral") == "orig"
    assert calls == [("orig", None), ("orig", "spectral")]
    # the variants are rebound when the package has them; foreign graphs / normalisations are forwarded
    pkg2 = types.SimpleNamespace(embed=pkg.embed.__wrapped__, whiten_embeddings=None,
                                 embed_multiscale=lambda *a, **k: "orig-ms", predict_links=lambda *a, **k: "orig-pl")
    cleora_amd.accelerate(pkg2)
    assert pkg2.embed_multiscale(object(), 8) == "orig-ms"
    assert pkg2.embed_multiscale(g, 8, normalization="spectral") == "orig-ms"
    assert pkg2.predict_links(object(), np.zeros((2, 2), np.float32)) == "orig-pl"
    assert not hasattr(pkg2, "embed_weighted")
    with pytest.raises(ValueError, match="scales must be"):
        pkg2.embed_multiscale(g, 8, scales=[0])                                 # our implementation validates first
    with pytest.raises(RuntimeError):          # l2 goes to the device path: no GPU here, and no fallback
        if __import__("cleora_amd._hip", fromlist=["x"]).device_count() == 0:
            pkg.embed(g, 8, 2, whiten=False)
        else:
            raise RuntimeError("gpu present")


def test_parallel_builder_is_thread_count_invariant_and_matches_the_oracle():
    """Inputs of >= 20 000 lines take the 4-phase parallel pipeline (cleora_host.cpp).  Its result must be the
    single-consumer result for ANY thread count (identical pickle bytes), and equal the Python restatement of the
    reference builder (oracle/refgraph.py) — including lines longer than hyperedge_trim_n, skipped lines,
    duplicate tokens inside a line, and unicode ids."""
    from cleora_amd import _host
    from oracle import refgraph
    rng = np.random.default_rng(321)
    n_lines = 24_000
    arity = np.minimum(22, 2 + rng.poisson(4, n_lines))
    arity[rng.integers(0, n_lines, 300)] = rng.integers(17, 23, 300)        # > hyperedge_trim_n = 16: trimmed
    members = np.minimum(rng.zipf(1.3, int(arity.sum())) - 1, 4000)
    lines, pos = [], 0
    for k in arity:
        lines.append(" ".join(f"e{m}" if m % 7 else f"é{m}" for m in members[pos:pos + k]))
        pos += k
    lines[100] = "a\tb"            # wrong column count for a one-column spec: skipped
    lines[101] = ""                # empty
    lines[102] = "  x7   x7  x8 "  # duplicate token, surrounding and repeated blanks
    L = _host.lib()
    blobs = []
    try:
        for t in (1, 2, 5, 8):
            L.cleora_host_set_threads(t)
            g = SparseMatrix.from_iterator(iter(lines), "complex::reflexive::e")
            blobs.append(g.__getstate__())
    finally:
        L.cleora_host_set_threads(0)
    assert all(b == blobs[0] for b in blobs[1:])
    want = refgraph.build_graph(lines, "complex::reflexive::e", 16)
    assert g.entity_ids == list(want.entity_ids)
    a = g._arr
    np.testing.assert_array_equal(a["rowptr"], want.rowptr)
    np.testing.assert_array_equal(a["col"], want.col)
    np.testing.assert_array_equal(a["val_left"], want.val_left)
    np.testing.assert_array_equal(a["val_sym"], want.val_sym)
    np.testing.assert_array_equal(a["row_sum"], want.row_sum)
    np.testing.assert_array_equal(a["hashes"], want.entity_hashes)
