"""ORACLE (test infrastructure): pure-Python restatement of the reference's
graph builder, for small inputs only.

Follows, in order (file:line under /root/reference):
  * column-spec DSL          src/configuration.rs:19-70
  * relation descriptor      src/sparse_matrix.rs:5-46 (exactly one relation)
  * line parsing             src/pipeline.rs:223-240, wrong-width skip :60-79
  * hashing / first-seen ids src/entity.rs:67-114, src/sparse_matrix_builder.rs:58-70
  * hyperedge expansion      src/sparse_matrix_builder.rs:170-233
  * reduce / sort / Markov   src/sparse_matrix_builder.rs:275-343

It restates the SINGLE-consumer behaviour (one SparseMatrixBuffer): with
several workers the reference sums the per-worker f32 partials in a
nondeterministic order and ranks `hyperedge_trim_n` candidates by worker-local
occurrence, so only the one-buffer order is well defined.  Ties inside
`select_nth_unstable_by_key` (sparse_matrix_builder.rs:201-203) are resolved by
Rust's pdqselect and are not restated: the oracle keeps first-position order
among equal occurrences (documented divergence; no reference test has a line
longer than trim_n).

Pinned by the four insta snapshots (tests/test_oracle_golden.py).
"""
import numpy as np
import xxhash

f32 = np.float32


def hash_entity(token: str) -> int:
    """src/entity.rs:109-114 — XXH64(seed 0) of the UTF-8 bytes."""
    return xxhash.xxh64_intdigest(token.encode("utf-8"), seed=0)


def parse_fields(columns: str):
    """src/configuration.rs:19-70 → list of (name, complex, reflexive)."""
    out = []
    for col in columns.split(" "):
        parts = col.split("::")
        complex_, reflexive = False, False
        if len(parts) > 1:
            name = parts[-1]
            for part in parts[:-1]:
                if part.lower() == "complex":
                    complex_ = True
                elif part.lower() == "reflexive":
                    reflexive = True
                else:
                    raise ValueError(f"Unrecognized column field modifier: {part}")
        else:
            name = col
        out.append((name, complex_, reflexive))
    for name, complex_, reflexive in out:
        if reflexive and not complex_:
            raise ValueError(
                f"A field cannot be REFLEXIVE but NOT COMPLEX. It does not make sense: {name}")
    return out


def relation_descriptor(cols):
    """src/sparse_matrix.rs:5-46 → (col_a_id, col_a_name, col_b_id, col_b_name)."""
    descs = []
    nf = len(cols)
    reflexive_count = 0
    for i in range(nf):
        for j in range(i, nf):
            if i < j:
                descs.append((i, cols[i][0], j, cols[j][0]))
            elif i == j and cols[i][2]:
                descs.append((i, cols[i][0], nf + reflexive_count, cols[j][0]))
                reflexive_count += 1
    if len(descs) != 1:
        raise ValueError("More than one relation! Adjust your columns so there is only one relation.")
    return descs[0]


def parse_line(line: str):
    """src/pipeline.rs:223-240."""
    t = line.strip()  # Rust str::trim: Unicode whitespace both ends
    if "\t" in t:
        return [c.split(" ") for c in t.split("\t")]
    if "," in t:
        return [c.strip().split(" ") for c in t.split(",")]
    return [t.split(" ")]


def edges_iter(hashes, slices, col_a, col_b):
    """Hyperedge::edges_iter (src/entity.rs:31-41): cartesian product of the two column slices,
    first column major — the order in which handle_combinations visits pairs."""
    a = hashes[slices[col_a][0]:slices[col_a][1]]
    b = hashes[slices[col_b][0]:slices[col_b][1]]
    return [(x, y) for x in a for y in b]


class RefGraph:
    """Result of the build: the same fields as struct SparseMatrix
    (src/sparse_matrix.rs:56-78) in SoA/CSR form."""

    def __init__(self):
        self.descriptor = None
        self.entity_ids = []
        self.entity_hashes = None  # u64[n]
        self.column_ids = None     # u8[n]
        self.row_sum = None        # f32[n]
        self.rowptr = None         # u64[n+1]
        self.col = None            # u32[nnz]
        self.val_left = None       # f32[nnz]
        self.val_sym = None        # f32[nnz]


def build_graph(lines, columns: str, hyperedge_trim_n: int = 16) -> RefGraph:
    cols = parse_fields(columns)
    desc = relation_descriptor(cols)
    col_a, _, col_b, _ = desc
    ncols = len(cols)

    key2index = {}
    index2key, index2id, index2col = [], [], []
    row_occ = {}   # hash -> occurrence (u32)
    row_sum = {}   # hash -> f32
    edges = {}     # (ha, hb) -> f32

    def index(h, token, column_id):
        if h in key2index:
            return
        key2index[h] = len(key2index)
        index2key.append(h)
        index2id.append(token)
        index2col.append(column_id)

    def update_row(h, count):
        row_occ[h] = row_occ.get(h, 0) + count
        row_sum[h] = f32(row_sum.get(h, f32(0.0)) + f32(1.0) / f32(count))

    def high_low(nodes):
        if len(nodes) > hyperedge_trim_n:
            order = sorted(range(len(nodes)), key=lambda i: -row_occ.get(nodes[i], 0))
            ranked = [nodes[i] for i in order]
            return ranked[:hyperedge_trim_n], ranked[hyperedge_trim_n:]
        return list(nodes), []

    def update_edge(a, b, v):
        edges[(a, b)] = f32(edges.get((a, b), f32(0.0)) + v)

    for line in lines:
        row = parse_line(line)
        if len(row) != ncols:
            continue  # pipeline.rs:60-79: warn + skip
        # process_row_and_get_edges (entity.rs:67-106)
        hashes = []
        slices = {}
        reflexive_count = 0
        offset = 0
        for i, ents in enumerate(row):
            name, complex_, reflexive = cols[i]
            if complex_:
                for tok in ents:
                    h = hash_entity(tok)
                    hashes.append(h)
                    index(h, tok, i)
                length = len(ents)
                slices[i] = (offset, offset + length)
                if reflexive:
                    slices[ncols + reflexive_count] = (offset, offset + length)
                    reflexive_count += 1
                offset += length
            else:
                tok = ents[0]
                h = hash_entity(tok)
                hashes.append(h)
                index(h, tok, i)
                slices[i] = (offset, offset + 1)
                offset += 1
        # handle_hyperedge (sparse_matrix_builder.rs:170-193)
        sa, sb = slices.get(col_a, (0, 0)), slices.get(col_b, (0, 0))
        nodes_a = hashes[sa[0]:sa[1]]
        nodes_b = hashes[sb[0]:sb[1]]
        total = len(nodes_a) * len(nodes_b)
        for h in nodes_a:
            update_row(h, len(nodes_b))
        for h in nodes_b:
            update_row(h, len(nodes_a))
        value = f32(1.0) / f32(total)
        a_high, a_low = high_low(nodes_a)
        b_high, b_low = high_low(nodes_b)
        for A, B in ((a_high, b_high), (a_high, b_low), (a_low, b_high)):
            for a in A:
                for b in B:
                    update_edge(a, b, value)
                    update_edge(b, a, value)

    n = len(index2key)
    g = RefGraph()
    g.descriptor = desc
    g.entity_ids = index2id
    g.entity_hashes = np.array(index2key, dtype=np.uint64)
    g.column_ids = np.array(index2col, dtype=np.uint8)
    g.row_sum = np.array([row_sum.get(h, f32(0.0)) for h in index2key], dtype=np.float32)

    trip = sorted(((key2index[a], key2index[b], v) for (a, b), v in edges.items()),
                  key=lambda t: (t[0], t[1]))
    nnz = len(trip)
    rows = np.array([t[0] for t in trip], dtype=np.int64)
    g.col = np.array([t[1] for t in trip], dtype=np.uint32)
    vals = np.array([t[2] for t in trip], dtype=np.float32)
    counts = np.bincount(rows, minlength=n) if nnz else np.zeros(n, dtype=np.int64)
    g.rowptr = np.zeros(n + 1, dtype=np.uint64)
    g.rowptr[1:] = np.cumsum(counts).astype(np.uint64)
    rs_row = g.row_sum[rows] if nnz else np.zeros(0, np.float32)
    rs_col = g.row_sum[g.col.astype(np.int64)] if nnz else np.zeros(0, np.float32)
    # sparse_matrix_builder.rs:315-332 — all f32
    g.val_left = (vals / rs_row).astype(np.float32)
    g.val_sym = (vals / np.sqrt(rs_row * rs_col, dtype=np.float32)).astype(np.float32)
    return g
