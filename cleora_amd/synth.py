"""Synthetic benchmark graphs, built straight into CSR on the GPU with torch.

torch is plumbing here (device RNG, sort, unique); the Markov values follow the
reference builder's closed forms (src/sparse_matrix_builder.rs:170-233, 315-332;
SURVEY.md Appendix B):

  two plain columns `u p`, distinct (u,p) lines:   E[u,p] = E[p,u] = 1, row_sum = degree
      left = 1/deg(row)                 sym = 1/sqrt(deg(row) deg(col))
  one `complex::reflexive` column, 2-token lines `a b` (a != b, distinct pairs):
      E[a,b] = E[b,a] = 1/2, E[a,a] = deg(a)/2, row_sum = deg
      left[a,b] = (1/2)/deg(a), left[a,a] = 1/2    sym[a,b] = (1/2)/sqrt(deg a deg b), sym[a,a] = 1/2

All value arithmetic is f32 like the reference's.  Entities that occur in no line do not
exist (ids are compacted), as in the reference.
"""
import torch


def _sqrt_f32_correctly_rounded(p):
    """IEEE-correct f32 square root of an f32 tensor, whatever the backend's math library does (torch's CPU sqrt
    is off by one ulp on some inputs; the reference divides by Rust's correctly rounded f32::sqrt,
    src/sparse_matrix_builder.rs:326-330).  A candidate c is the correctly rounded root iff
    (c - ulp/2)^2 <= p <= (c + ulp/2)^2; the midpoints have 25 significant bits, so their squares are exact in f64."""
    s = torch.sqrt(p)
    p64 = p.double()
    inf = torch.full_like(s, float("inf"))
    for _ in range(2):
        up, down = torch.nextafter(s, inf), torch.nextafter(s, -inf)
        hi = (s.double() + up.double()) * 0.5
        lo = (s.double() + down.double()) * 0.5
        s = torch.where(hi * hi < p64, up, torch.where(lo * lo > p64, down, s))
    return s


def _csr_from_undirected(a, b, n_nodes, reflexive):
    """a, b: int64 endpoint tensors of DISTINCT undirected pairs with a != b."""
    dev = a.device
    deg = torch.bincount(torch.cat([a, b]), minlength=n_nodes)
    present = deg > 0
    n = int(present.sum())
    remap = torch.cumsum(present.to(torch.int64), 0) - 1
    a, b, deg = remap[a], remap[b], deg[present]
    del remap, present
    rows = torch.cat([a, b])
    cols = torch.cat([b, a])
    del a, b
    if reflexive:
        diag = torch.arange(n, device=dev, dtype=torch.int64)
        rows = torch.cat([rows, diag])
        cols = torch.cat([cols, diag])
        del diag
    key = rows * n + cols
    del rows, cols
    key, _ = torch.sort(key)
    rows = torch.div(key, n, rounding_mode="floor")
    cols = key - rows * n
    del key
    counts = torch.bincount(rows, minlength=n)
    rowptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    torch.cumsum(counts, 0, out=rowptr[1:])
    degf = deg.to(torch.float32)
    base = torch.tensor(0.5 if reflexive else 1.0, dtype=torch.float32, device=dev)
    val_left = base / degf[rows]
    val_sym = base / _sqrt_f32_correctly_rounded(degf[rows] * degf[cols])
    if reflexive:
        is_diag = rows == cols
        val_left[is_diag] = 0.5   # (deg/2)/deg
        val_sym[is_diag] = 0.5    # (deg/2)/sqrt(deg*deg)
    col32 = cols.to(torch.int32)
    return {"n": n, "nnz": int(col32.numel()), "rowptr": rowptr, "col": col32,
            "val_left": val_left.contiguous(), "val_sym": val_sym.contiguous(), "deg": deg}


def _distinct_pairs(a, b, n_nodes):
    keep = a != b
    a, b = a[keep], b[keep]
    lo, hi = torch.minimum(a, b), torch.maximum(a, b)
    key = torch.unique(lo * n_nodes + hi)
    lo = torch.div(key, n_nodes, rounding_mode="floor")
    return lo, key - lo * n_nodes


def power_law_graph(n_nodes, n_pairs, seed, device):
    """BASELINE config 3 shape: pairs a = pi(floor(n r1^3)), b = pi(floor(n r2)),
    pi a random permutation (no index locality); reflexive column semantics, so
    nnz = 2 * distinct pairs + n.  Returns the CSR dict (device tensors)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    r1 = torch.rand(n_pairs, generator=g, device=device, dtype=torch.float64)
    r2 = torch.rand(n_pairs, generator=g, device=device, dtype=torch.float64)
    perm = torch.randperm(n_nodes, generator=g, device=device)
    a = perm[(r1 * r1 * r1 * n_nodes).to(torch.int64).clamp_(max=n_nodes - 1)]
    b = perm[(r2 * n_nodes).to(torch.int64).clamp_(max=n_nodes - 1)]
    del r1, r2, perm
    a, b = _distinct_pairs(a, b, n_nodes)
    return _csr_from_undirected(a, b, n_nodes, reflexive=True)


def bipartite_graph(n_users, n_items, n_pairs, seed, device):
    """BASELINE config 2 shape: u ~ U[0, n_users), p = floor(n_items r^2) (popularity skew),
    two plain columns `user product` => nnz = 2 * distinct pairs, no self loops."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    u = torch.randint(0, n_users, (n_pairs,), generator=g, device=device, dtype=torch.int64)
    r = torch.rand(n_pairs, generator=g, device=device, dtype=torch.float64)
    p = (r * r * n_items).to(torch.int64).clamp_(max=n_items - 1) + n_users
    del r
    key = torch.unique(u * (n_users + n_items) + p)
    u = torch.div(key, n_users + n_items, rounding_mode="floor")
    p = key - u * (n_users + n_items)
    return _csr_from_undirected(u, p, n_users + n_items, reflexive=False)


def entity_hashes(n, seed, device):
    """Stand-in for XXH64(entity id): n distinct-ish 64-bit values (splitmix64 of the index) so
    that initialize_deterministically has realistic inputs without 10M strings."""
    off = ((seed + 1) * -7046029254386353131 + (1 << 63)) % (1 << 64) - (1 << 63)     # wraps like the int64 arithmetic below
    x = torch.arange(n, device=device, dtype=torch.int64) + off
    x = (x ^ (x >> 30)) * -4658895280553007687
    x = (x ^ (x >> 27)) * -7723592293110705685
    return x ^ (x >> 31)
