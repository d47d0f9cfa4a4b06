"""ORACLE (test infrastructure): rand 0.8.5 `StdRng` restated in Python.

The reference's only golden vectors for the hot path (tests/snapshot.rs:52-117)
are generated from `StdRng::seed_from_u64(2137)`.  rand / rand_chacha /
rand_core are third-party crates (Cargo.lock: rand 0.8.5, rand_chacha 0.3.1,
rand_core 0.6.4) that are not vendored under /root/reference, so their
published algorithms are restated here:

* `SeedableRng::seed_from_u64` (rand_core 0.6.4): a PCG32 stream expands the
  u64 into the 32-byte seed, 4 bytes at a time, little-endian.
* `StdRng` = `ChaCha12Rng`: ChaCha with 12 rounds, 64-bit block counter in
  state words 12-13 starting at 0, stream id (words 14-15) = 0; `next_u32`
  consumes the key-stream word by word.
* `Uniform::<f32>::new(low, high)` sampling (rand 0.8.5 `UniformFloat`):
  `value1_2 = f32::from_bits((next_u32() >> 9) | 0x3F80_0000)`,
  `(value1_2 - 1.0) * scale + low` with scale = high - low (for (0, 10) the
  constructor's shrink loop leaves scale at exactly 10.0).

The restatement is pinned by the snapshots themselves: the graphs and
embeddings it generates reproduce all four .snap files (tests/test_oracle_golden.py).
"""
import numpy as np

_MASK32 = 0xFFFFFFFF
_MASK64 = 0xFFFFFFFFFFFFFFFF


def _rotl32(x, r):
    return ((x << r) | (x >> (32 - r))) & _MASK32


def _quarter(s, a, b, c, d):
    s[a] = (s[a] + s[b]) & _MASK32; s[d] = _rotl32(s[d] ^ s[a], 16)
    s[c] = (s[c] + s[d]) & _MASK32; s[b] = _rotl32(s[b] ^ s[c], 12)
    s[a] = (s[a] + s[b]) & _MASK32; s[d] = _rotl32(s[d] ^ s[a], 8)
    s[c] = (s[c] + s[d]) & _MASK32; s[b] = _rotl32(s[b] ^ s[c], 7)


def _seed_from_u64(state):
    """rand_core 0.6.4 SeedableRng::seed_from_u64: PCG32 expansion to 32 bytes."""
    mul, inc = 6364136223846793005, 11634580027462260723
    words = []
    for _ in range(8):
        state = (state * mul + inc) & _MASK64
        xorshifted = (((state >> 18) ^ state) >> 27) & _MASK32
        rot = state >> 59
        words.append(((xorshifted >> rot) | (xorshifted << ((32 - rot) & 31))) & _MASK32)
    return words


class StdRng:
    def __init__(self, seed_u64):
        self.key = _seed_from_u64(seed_u64)
        self.counter = 0
        self.buf = []

    def _block(self):
        init = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574] + list(self.key) + [
            self.counter & _MASK32, (self.counter >> 32) & _MASK32, 0, 0]
        s = list(init)
        for _ in range(6):  # 12 rounds = 6 double rounds
            _quarter(s, 0, 4, 8, 12); _quarter(s, 1, 5, 9, 13)
            _quarter(s, 2, 6, 10, 14); _quarter(s, 3, 7, 11, 15)
            _quarter(s, 0, 5, 10, 15); _quarter(s, 1, 6, 11, 12)
            _quarter(s, 2, 7, 8, 13); _quarter(s, 3, 4, 9, 14)
        self.counter += 1
        return [(s[i] + init[i]) & _MASK32 for i in range(16)]

    def next_u32(self):
        if not self.buf:
            self.buf = self._block()
        return self.buf.pop(0)

    def uniform_f32(self, low, high, count):
        scale = np.float32(high) - np.float32(low)
        out = np.empty(count, dtype=np.float32)
        for i in range(count):
            bits = np.uint32((self.next_u32() >> 9) | 0x3F800000)
            v12 = bits.view(np.float32)
            out[i] = (v12 - np.float32(1.0)) * scale + np.float32(low)
        return out


def snapshot_fixture(kind):
    """Inputs of tests/snapshot.rs: kind 'reflexive' (:89-117) or 'complex' (:52-87).

    Returns (lines, columns_spec, embeddings[100,32] f32)."""
    rng = StdRng(2137)
    lines = []
    if kind == "reflexive":
        for _ in range(1000):
            a = rng.next_u32() % 100
            b = rng.next_u32() % 100
            lines.append(f"{a} {b}")
        columns = "reflexive::complex::entity_id"
    elif kind == "complex":
        for _ in range(1000):
            a1 = rng.next_u32() % 100
            a2 = rng.next_u32() % 100
            b1 = rng.next_u32() % 100
            b2 = rng.next_u32() % 100
            lines.append(f"{a1} {a2}\t{b1} {b2}")
        columns = "complex::entity_a complex::entity_b"
    else:
        raise ValueError(kind)
    emb = rng.uniform_f32(0.0, 10.0, 100 * 32).reshape(100, 32)
    return lines, columns, emb
