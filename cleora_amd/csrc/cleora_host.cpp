// cleora_host.cpp — host-side graph construction for the drop-in (include/cleora_host.h).
// Written from the behaviour of the reference's Rust builder (citations: paths under the
// reference checkout); data structures and control flow are this project's own: tokens are hashed
// in parallel, entities are interned to dense indices in line order, each worker owns a contiguous
// row range and accumulates its edges in a private open-addressing table keyed by (row << 32 | col),
// and the CSR is the concatenation of the workers' sorted ranges (see `struct Builder`).
#include "../../include/cleora_host.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <string_view>
#include <atomic>
#include <chrono>
#include <functional>
#include <thread>
#include <vector>

namespace {

thread_local std::string g_err;

// ---- XXH64 (public specification; twox-hash is not under the reference tree) -------------------
constexpr uint64_t P1 = 0x9E3779B185EBCA87ULL, P2 = 0xC2B2AE3D27D4EB4FULL, P3 = 0x165667B19E3779F9ULL,
                   P4 = 0x85EBCA77C2B2AE63ULL, P5 = 0x27D4EB2F165667C5ULL;
inline uint64_t rotl(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
inline uint64_t rd64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }
inline uint32_t rd32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
inline uint64_t rnd(uint64_t acc, uint64_t in) { return rotl(acc + in * P2, 31) * P1; }
inline uint64_t mrg(uint64_t h, uint64_t v) { return (h ^ rnd(0, v)) * P1 + P4; }

uint64_t xxh64(const uint8_t *p, uint64_t len, uint64_t seed) {
    const uint8_t *end = p + len;
    uint64_t h;
    if (len >= 32) {
        uint64_t v1 = seed + P1 + P2, v2 = seed + P2, v3 = seed, v4 = seed - P1;
        const uint8_t *lim = end - 32;
        do {
            v1 = rnd(v1, rd64(p)); v2 = rnd(v2, rd64(p + 8)); v3 = rnd(v3, rd64(p + 16)); v4 = rnd(v4, rd64(p + 24));
            p += 32;
        } while (p <= lim);
        h = rotl(v1, 1) + rotl(v2, 7) + rotl(v3, 12) + rotl(v4, 18);
        h = mrg(mrg(mrg(mrg(h, v1), v2), v3), v4);
    } else {
        h = seed + P5;
    }
    h += len;
    for (; p + 8 <= end; p += 8) h = rotl(h ^ rnd(0, rd64(p)), 27) * P1 + P4;
    if (p + 4 <= end) { h = rotl(h ^ (rd32(p) * P1), 23) * P2 + P3; p += 4; }
    for (; p < end; ++p) h = rotl(h ^ (*p * P5), 11) * P1;
    h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3; h ^= h >> 32;
    return h;
}

// ---- column spec (src/configuration.rs:19-70) and relation (src/sparse_matrix.rs:5-46) -----------
struct Column { std::string name; bool complex = false, reflexive = false; };

bool ieq(std::string_view a, const char *b) {
    size_t n = strlen(b);
    if (a.size() != n) return false;
    for (size_t i = 0; i < n; ++i) if (tolower((unsigned char)a[i]) != b[i]) return false;
    return true;
}

std::vector<std::string_view> split(std::string_view s, std::string_view sep) {
    std::vector<std::string_view> out;
    size_t pos = 0;
    for (;;) {
        size_t q = s.find(sep, pos);
        if (q == std::string_view::npos) { out.push_back(s.substr(pos)); break; }
        out.push_back(s.substr(pos, q - pos));
        pos = q + sep.size();
    }
    return out;
}

bool parse_fields(const char *spec, std::vector<Column> &cols) {
    for (auto col : split(spec, " ")) {
        auto parts = split(col, "::");
        Column c;
        if (parts.size() > 1) {
            c.name = std::string(parts.back());
            for (size_t i = 0; i + 1 < parts.size(); ++i) {
                if (ieq(parts[i], "complex")) c.complex = true;
                else if (ieq(parts[i], "reflexive")) c.reflexive = true;
                else { g_err = "Unrecognized column field modifier: " + std::string(parts[i]); return false; }
            }
        } else {
            c.name = std::string(col);
        }
        cols.push_back(c);
    }
    for (auto &c : cols)
        if (c.reflexive && !c.complex) {
            g_err = "A field cannot be REFLEXIVE but NOT COMPLEX. It does not make sense: " + c.name;
            return false;
        }
    return true;
}

struct Descriptor { uint8_t a_id = 0, b_id = 0; std::string a_name, b_name; };

bool relation(const std::vector<Column> &cols, Descriptor &d) {
    std::vector<Descriptor> all;
    const size_t nf = cols.size();
    size_t refl = 0;
    for (size_t i = 0; i < nf; ++i)
        for (size_t j = i; j < nf; ++j) {
            if (i < j) all.push_back({(uint8_t)i, (uint8_t)j, cols[i].name, cols[j].name});
            else if (cols[i].reflexive) { all.push_back({(uint8_t)i, (uint8_t)(nf + refl), cols[i].name, cols[j].name}); ++refl; }
        }
    if (all.size() != 1) {
        g_err = "More than one relation! Adjust your columns so there is only one relation.";
        return false;
    }
    d = all[0];
    return true;
}

// ---- Rust str::trim (Unicode White_Space) on UTF-8 ------------------------------------------------
bool ws_at(std::string_view s, size_t i, size_t &len) {
    unsigned char c = s[i];
    if (c == ' ' || (c >= 9 && c <= 13)) { len = 1; return true; }
    if (c == 0xC2 && i + 1 < s.size() && ((unsigned char)s[i + 1] == 0x85 || (unsigned char)s[i + 1] == 0xA0)) { len = 2; return true; }
    if (i + 2 < s.size()) {
        unsigned char c1 = s[i + 1], c2 = s[i + 2];
        if (c == 0xE1 && c1 == 0x9A && c2 == 0x80) { len = 3; return true; }                  // U+1680
        if (c == 0xE2 && c1 == 0x80 && ((c2 >= 0x80 && c2 <= 0x8A) || c2 == 0xA8 || c2 == 0xA9 || c2 == 0xAF)) { len = 3; return true; }
        if (c == 0xE2 && c1 == 0x81 && c2 == 0x9F) { len = 3; return true; }                  // U+205F
        if (c == 0xE3 && c1 == 0x80 && c2 == 0x80) { len = 3; return true; }                  // U+3000
    }
    return false;
}

std::string_view trim(std::string_view s) {
    size_t b = 0, len;
    while (b < s.size() && ws_at(s, b, len)) b += len;
    size_t e = s.size();
    for (;;) {
        if (e <= b) break;
        // step back one code point
        size_t k = e - 1;
        while (k > b && ((unsigned char)s[k] & 0xC0) == 0x80) --k;
        if (ws_at(s, k, len) && k + len == e) e = k; else break;
    }
    return s.substr(b, e - b);
}

// ---- open-addressing tables ---------------------------------------------------------------------------
inline uint64_t mix(uint64_t x) { x ^= x >> 32; x *= 0xD6E8FEB86659FD93ULL; x ^= x >> 32; return x; }

struct Interner {  // entity hash -> dense index, first-seen order (SyncNodeIndexerBuilder, :58-70)
    std::vector<uint64_t> keys; std::vector<uint32_t> vals; size_t mask = 0, count = 0;
    Interner() { resize(1 << 12); }
    void resize(size_t cap) {
        std::vector<uint64_t> ok = std::move(keys); std::vector<uint32_t> ov = std::move(vals);
        keys.assign(cap, 0); vals.assign(cap, UINT32_MAX); mask = cap - 1;
        for (size_t i = 0; i < ok.size(); ++i) if (ov[i] != UINT32_MAX) put(ok[i], ov[i]);
    }
    void put(uint64_t k, uint32_t v) { size_t i = mix(k) & mask; while (vals[i] != UINT32_MAX) i = (i + 1) & mask; keys[i] = k; vals[i] = v; }
    uint32_t find_or_add(uint64_t k, bool &added) {
        size_t i = mix(k) & mask;
        while (vals[i] != UINT32_MAX) { if (keys[i] == k) { added = false; return vals[i]; } i = (i + 1) & mask; }
        added = true;
        uint32_t v = (uint32_t)count++;
        keys[i] = k; vals[i] = v;
        if (count * 2 > mask) resize((mask + 1) * 2);
        return v;
    }
};

struct EdgeTable {  // (row << 32 | col) -> f32 running sum (SparseMatrixBuffer::hashes_2_edge)
    static constexpr uint64_t EMPTY = ~0ULL;
    std::vector<uint64_t> keys; std::vector<float> vals; size_t mask = 0, count = 0;
    EdgeTable() { keys.assign(1 << 14, EMPTY); vals.assign(1 << 14, 0.f); mask = (1 << 14) - 1; }
    void grow() {
        std::vector<uint64_t> ok = std::move(keys); std::vector<float> ov = std::move(vals);
        size_t cap = (mask + 1) * 2;
        keys.assign(cap, EMPTY); vals.assign(cap, 0.f); mask = cap - 1;
        for (size_t i = 0; i < ok.size(); ++i) if (ok[i] != EMPTY) { size_t j = mix(ok[i]) & mask; while (keys[j] != EMPTY) j = (j + 1) & mask; keys[j] = ok[i]; vals[j] = ov[i]; }
    }
    void add(uint32_t r, uint32_t c, float v) {
        const uint64_t k = ((uint64_t)r << 32) | c;
        size_t i = mix(k) & mask;
        while (keys[i] != EMPTY) { if (keys[i] == k) { vals[i] += v; return; } i = (i + 1) & mask; }
        keys[i] = k; vals[i] = v;
        if (++count * 2 > mask) grow();
    }
};

}  // namespace

struct cleora_hostgraph {
    Descriptor desc;
    std::vector<std::string> ids;
    std::vector<uint64_t> hashes;
    std::vector<uint8_t> column_ids;
    std::vector<float> row_sum;
    std::vector<uint64_t> rowptr{0};
    std::vector<uint32_t> col;
    std::vector<float> val_left, val_sym;
};

namespace {

// ---- the builder ------------------------------------------------------------------------------------
// Four phases, deterministic for any thread count (the result is always the reference's
// single-consumer result: every f32 sum is accumulated in input-line order):
//   A  parallel over line chunks : parse_line + XXH64 of every token        (pipeline.rs:223-240, entity.rs:109-114)
//   B  sequential, line order    : first-seen interning, Row::occurrence / row_sum, hyperedge_trim_n
//                                  partitions                              (sparse_matrix_builder.rs:58-70,170-233)
//   C  parallel, two steps       : producers route every E[r, c] update of their chunk of hyperedges to the
//                                  bucket of the row's owner (rows are cut into ranges of equal work);
//                                  each owner drains its buckets in chunk order = line order (no locks)
//   D  parallel per row range    : sort by (row, col), Markov normalisation; ranges concatenate into CSR
struct ParsedLine {
    uint32_t tok_begin = 0;   // into Parsed::hash / Parsed::span
    uint32_t na = 0, nb = 0;  // tokens of the descriptor's two node lists (reflexive: the same list, nb == na)
    uint16_t ncols = 0;       // 0 = skipped line
    bool reflexive = false;
};

struct Parsed {
    std::vector<ParsedLine> lines;
    std::vector<uint64_t> hash;                        // per token
    std::vector<std::pair<const char *, uint32_t>> span;  // per token: text
    std::vector<uint8_t> column;                       // per token: column id
};

struct Builder {
    std::vector<Column> cols;
    Descriptor desc;
    uint32_t trim_n = 16;
    unsigned threads = 1;

    // phase A: one line -> tokens of column a then column b (or the single reflexive column)
    void parse(std::string_view raw, Parsed &out) const {
        ParsedLine pl;
        pl.tok_begin = (uint32_t)out.hash.size();
        std::string_view t = trim(raw);
        std::vector<std::string_view> columns;
        bool comma = false;
        if (t.find('\t') != std::string_view::npos) columns = split(t, "\t");
        else if (t.find(',') != std::string_view::npos) { columns = split(t, ","); comma = true; }
        else columns.push_back(t);
        if (columns.size() != cols.size()) { out.lines.push_back(pl); return; }  // skipped (pipeline.rs:60-79)
        for (size_t i = 0; i < columns.size(); ++i) {
            std::string_view c = comma ? trim(columns[i]) : columns[i];
            auto toks = split(c, " ");
            size_t take = cols[i].complex ? toks.size() : 1;   // non-complex: first token only (entity.rs:94)
            for (size_t k = 0; k < take; ++k) {
                out.hash.push_back(xxh64(reinterpret_cast<const uint8_t *>(toks[k].data()), toks[k].size(), 0));
                out.span.emplace_back(toks[k].data(), (uint32_t)toks[k].size());
                out.column.push_back((uint8_t)i);
            }
            if (i == 0) pl.na = (uint32_t)take; else pl.nb = (uint32_t)take;
        }
        pl.ncols = (uint16_t)cols.size();
        pl.reflexive = cols.size() == 1;
        if (pl.reflexive) pl.nb = pl.na;
        out.lines.push_back(pl);
    }

    cleora_hostgraph *build(const std::vector<std::string_view> &lines) {
        const size_t nl = lines.size();
        unsigned T = threads ? threads : 1;
        if (nl < 20000) T = 1;
        // ---- A ----
        std::vector<Parsed> chunks(T);
        {
            std::vector<std::thread> pool;
            for (unsigned t = 0; t < T; ++t)
                pool.emplace_back([&, t] {
                    const size_t lo = nl * t / T, hi = nl * (t + 1) / T;
                    chunks[t].lines.reserve(hi - lo);
                    for (size_t i = lo; i < hi; ++i) parse(lines[i], chunks[t]);
                });
            for (auto &th : pool) th.join();
        }
        // ---- B ----
        Interner interner;
        auto *g = new cleora_hostgraph();
        g->desc = desc;
        std::vector<uint32_t> occurrence;
        std::vector<float> &row_sum = g->row_sum;
        // hyperedges in line order: node indices of list a then list b (after the trim reorder),
        // with the number of "high" nodes of each list
        struct Hyper { uint64_t begin; uint32_t na, nb, ah, bh; float value; };
        std::vector<Hyper> hypers;
        std::vector<uint32_t> nodes;
        hypers.reserve(nl);
        auto high_first = [&](uint32_t *v, size_t n) -> size_t {
            // get_high_low_nodes (sparse_matrix_builder.rs:195-208); ties keep line order (documented)
            if (n <= trim_n) return n;
            std::stable_sort(v, v + n, [&](uint32_t x, uint32_t y) { return occurrence[x] > occurrence[y]; });
            return trim_n;
        };
        for (unsigned t = 0; t < T; ++t) {
            const Parsed &P = chunks[t];
            for (const ParsedLine &pl : P.lines) {
                if (!pl.ncols) continue;
                const uint32_t ntok = pl.reflexive ? pl.na : pl.na + pl.nb;
                const uint64_t begin = nodes.size();
                for (uint32_t k = 0; k < ntok; ++k) {
                    bool added;
                    const uint64_t h = P.hash[pl.tok_begin + k];
                    const uint32_t ix = interner.find_or_add(h, added);
                    if (added) {
                        g->ids.emplace_back(P.span[pl.tok_begin + k].first, P.span[pl.tok_begin + k].second);
                        g->hashes.push_back(h);
                        g->column_ids.push_back(P.column[pl.tok_begin + k]);
                        occurrence.push_back(0);
                        row_sum.push_back(0.f);
                    }
                    nodes.push_back(ix);
                }
                if (pl.reflexive)
                    for (uint32_t k = 0; k < pl.na; ++k) { const uint32_t v = nodes[begin + k]; nodes.push_back(v); }
                uint32_t *a = nodes.data() + begin, *b = a + pl.na;
                const uint32_t na = pl.na, nb = pl.nb;
                for (uint32_t k = 0; k < na; ++k) { occurrence[a[k]] += nb; row_sum[a[k]] += 1.0f / (float)nb; }
                for (uint32_t k = 0; k < nb; ++k) { occurrence[b[k]] += na; row_sum[b[k]] += 1.0f / (float)na; }
                Hyper h;
                h.begin = begin; h.na = na; h.nb = nb;
                h.value = 1.0f / (float)(na * nb);
                h.ah = (uint32_t)high_first(a, na);
                h.bh = (uint32_t)high_first(b, nb);
                hypers.push_back(h);
            }
            chunks[t] = Parsed();  // release
        }
        const size_t n = g->ids.size();
        // ---- C + D ----
        // Rows are cut into P contiguous ranges of (about) equal WORK — first-seen order puts the popular entities first, and
        // `occurrence` counts the pair updates a row takes part in.  P is chosen by the size of the job, NOT by the thread
        // count: every range accumulates into a private hash table, and the phase is bound by how well that table caches
        // (measured on 8 cores, 6.5M edges: 8 ranges 1.07 s, 32 ranges 0.54 s) — about 2^19 updates per range; the T workers
        // take ranges from a counter.  The result does not depend on P, S or T: a range drains its updates in global line
        // order whatever produced them.
        auto pair_updates = [](const Hyper &h) -> uint64_t {
            return 2ull * ((uint64_t)h.ah * h.bh + (uint64_t)h.ah * (h.nb - h.bh) + (uint64_t)(h.na - h.ah) * h.bh);
        };
        uint64_t total_updates = 0;
        for (const Hyper &h : hypers) total_updates += pair_updates(h);
        const bool small = hypers.size() < 20000 || n < 1024;
        unsigned P = 1;
        if (!small && T > 1) {     // one worker: the single table is faster than routing through buckets (2.2 s vs 2.8 s)
            const uint64_t want = (total_updates >> 19) + 1;
            P = (unsigned)std::min<uint64_t>(std::max<uint64_t>(want, T), 4096);
            if ((size_t)P > n) P = (unsigned)n;
        }
        const unsigned T2 = P;
        struct Part { std::vector<std::pair<uint64_t, float>> ent; };
        std::vector<Part> parts(T2);
        std::vector<uint32_t> bound(T2 + 1, (uint32_t)n);
        {
            uint64_t total = 0;
            for (size_t r = 0; r < n; ++r) total += occurrence[r];
            uint64_t acc = 0;
            unsigned next = 1;
            bound[0] = 0;
            for (size_t r = 0; r < n && next < T2; ++r) {
                acc += occurrence[r];
                while (next < T2 && acc * T2 >= total * next) bound[next++] = (uint32_t)(r + 1);
            }
        }
        // `count` independent tasks on T workers, handed out by a counter
        auto for_each_task = [&](unsigned count, const std::function<void(unsigned)> &fn) {
            const unsigned workers = std::min(T, count);
            if (workers <= 1) {
                for (unsigned i = 0; i < count; ++i) fn(i);
                return;
            }
            std::atomic<unsigned> next{0};
            std::vector<std::thread> pool;
            for (unsigned w = 0; w < workers; ++w)
                pool.emplace_back([&] {
                    for (unsigned i = next.fetch_add(1); i < count; i = next.fetch_add(1)) fn(i);
                });
            for (auto &th : pool) th.join();
        };
        if (T2 == 1) {
            EdgeTable edges;
            for (const Hyper &h : hypers) {
                const uint32_t *a = nodes.data() + h.begin, *b = a + h.na;
                auto combos = [&](size_t a0, size_t a1, size_t b0, size_t b1) {
                    for (size_t i = a0; i < a1; ++i)
                        for (size_t j = b0; j < b1; ++j) { edges.add(a[i], b[j], h.value); edges.add(b[j], a[i], h.value); }
                };
                combos(0, h.ah, 0, h.bh);        // high x high
                combos(0, h.ah, h.bh, h.nb);     // high x low
                combos(h.ah, h.na, 0, h.bh);     // low  x high   (low x low dropped)
            }
            auto &ent = parts[0].ent;
            ent.reserve(edges.count);
            for (size_t i = 0; i < edges.keys.size(); ++i)
                if (edges.keys[i] != EdgeTable::EMPTY) ent.emplace_back(edges.keys[i], edges.vals[i]);
            std::sort(ent.begin(), ent.end(), [](auto &x, auto &y) { return x.first < y.first; });
        } else {
            // C1: S producers (contiguous chunks of hyperedges) route every update to the bucket of the row's range;
            // C2: range p drains bucket[0][p], bucket[1][p], ... — its updates in global line order — into a private
            //     table, then sorts its rows.
            struct Upd { uint32_t r, c; float v; };
            std::vector<uint16_t> owner_of(n);
            for (unsigned t = 0; t < T2; ++t)
                for (uint32_t r = bound[t]; r < bound[t + 1]; ++r) owner_of[r] = (uint16_t)t;
            const unsigned S = T;
            std::vector<std::vector<std::vector<Upd>>> bucket(S, std::vector<std::vector<Upd>>(T2));
            const size_t nh = hypers.size();
            for_each_task(S, [&](unsigned sidx) {
                auto &mine = bucket[sidx];
                auto emit = [&](uint32_t r, uint32_t c, float v) { mine[owner_of[r]].push_back({r, c, v}); };
                for (size_t k = nh * sidx / S; k < nh * (sidx + 1) / S; ++k) {
                    const Hyper &h = hypers[k];
                    const uint32_t *a = nodes.data() + h.begin, *b = a + h.na;
                    auto combos = [&](size_t a0, size_t a1, size_t b0, size_t b1) {
                        for (size_t i = a0; i < a1; ++i)
                            for (size_t j = b0; j < b1; ++j) { emit(a[i], b[j], h.value); emit(b[j], a[i], h.value); }
                    };
                    combos(0, h.ah, 0, h.bh);
                    combos(0, h.ah, h.bh, h.nb);
                    combos(h.ah, h.na, 0, h.bh);
                }
            });
            for_each_task(T2, [&](unsigned t) {
                EdgeTable edges;
                for (unsigned sidx = 0; sidx < S; ++sidx) {
                    for (const Upd &u : bucket[sidx][t]) edges.add(u.r, u.c, u.v);
                    std::vector<Upd>().swap(bucket[sidx][t]);
                }
                auto &ent = parts[t].ent;
                ent.reserve(edges.count);
                for (size_t i = 0; i < edges.keys.size(); ++i)
                    if (edges.keys[i] != EdgeTable::EMPTY) ent.emplace_back(edges.keys[i], edges.vals[i]);
                std::sort(ent.begin(), ent.end(), [](auto &x, auto &y) { return x.first < y.first; });
            });
        }
        // reduce (sparse_matrix_builder.rs:275-343): rows ascending = parts in order
        size_t nnz = 0;
        std::vector<size_t> base(T2 + 1, 0);
        for (unsigned t = 0; t < T2; ++t) { base[t] = nnz; nnz += parts[t].ent.size(); }
        base[T2] = nnz;
        g->rowptr.assign(n + 1, 0);
        g->col.resize(nnz); g->val_left.resize(nnz); g->val_sym.resize(nnz);
        for_each_task(T2, [&](unsigned t) {
            const auto &ent = parts[t].ent;
            for (size_t k = 0; k < ent.size(); ++k) {
                const uint32_t r = (uint32_t)(ent[k].first >> 32), c = (uint32_t)ent[k].first;
                g->rowptr[r + 1]++;   // rows of different parts are disjoint
                const size_t o = base[t] + k;
                g->col[o] = c;
                const float v = ent[k].second, rs = row_sum[r], cs = row_sum[c];
                g->val_left[o] = v / rs;
                g->val_sym[o] = v / std::sqrt(rs * cs);
            }
        });
        for (size_t r = 0; r < n; ++r) g->rowptr[r + 1] += g->rowptr[r];
        return g;
    }
};

unsigned g_threads = 0;  // 0 = hardware concurrency

bool make_builder(Builder &b, const char *columns, uint32_t trim_n) {
    if (!columns) { g_err = "columns is NULL"; return false; }
    if (!parse_fields(columns, b.cols)) return false;
    if (!relation(b.cols, b.desc)) return false;
    b.trim_n = trim_n;
    unsigned hw = std::thread::hardware_concurrency();
    b.threads = g_threads ? g_threads : (hw ? (hw > 64 ? 64 : hw) : 1);
    return true;
}

// ---- bincode helpers -------------------------------------------------------------------------------------
struct Writer {
    std::vector<uint8_t> buf;
    template <class T> void put(T v) { const auto *p = reinterpret_cast<const uint8_t *>(&v); buf.insert(buf.end(), p, p + sizeof(T)); }
    void str(const std::string &s) { put<uint64_t>(s.size()); buf.insert(buf.end(), s.begin(), s.end()); }
};
// core::str::from_utf8's acceptance rule (what bincode 1.3.3 applies to every `String` it reads: deserialize_string ->
// String::from_utf8; the reference surfaces the failure as RuntimeError("Deserialization failed: ..."), src/lib.rs:468-475):
// shortest-form encodings only, no surrogates (U+D800..U+DFFF), nothing above U+10FFFF.
static bool valid_utf8(const uint8_t *p, uint64_t n) {
    uint64_t i = 0;
    while (i < n) {
        uint8_t c = p[i];
        if (c < 0x80) { ++i; continue; }
        uint8_t lo = 0x80, hi = 0xBF; unsigned more;
        if (c >= 0xC2 && c <= 0xDF) more = 1;
        else if (c == 0xE0) { more = 2; lo = 0xA0; }
        else if ((c >= 0xE1 && c <= 0xEC) || c == 0xEE || c == 0xEF) more = 2;
        else if (c == 0xED) { more = 2; hi = 0x9F; }
        else if (c == 0xF0) { more = 3; lo = 0x90; }
        else if (c >= 0xF1 && c <= 0xF3) more = 3;
        else if (c == 0xF4) { more = 3; hi = 0x8F; }
        else return false;                                   // 0x80..0xC1 (continuation / overlong lead), 0xF5..0xFF
        if (n - i <= more) return false;
        if (p[i + 1] < lo || p[i + 1] > hi) return false;
        for (unsigned k = 2; k <= more; ++k) if ((p[i + k] & 0xC0) != 0x80) return false;
        i += more + 1;
    }
    return true;
}

struct Reader {
    const uint8_t *p, *end; bool ok = true, utf8_ok = true;
    template <class T> T get() { T v{}; if ((size_t)(end - p) < sizeof(T)) { ok = false; return v; } memcpy(&v, p, sizeof(T)); p += sizeof(T); return v; }
    // a bincode `String`: u64 length, then that many bytes, which must be UTF-8
    std::string str() {
        uint64_t n = get<uint64_t>();
        if (!ok || (uint64_t)(end - p) < n) { ok = false; return {}; }
        if (!valid_utf8(p, n)) { ok = false; utf8_ok = false; return {}; }
        std::string s(reinterpret_cast<const char *>(p), n); p += n; return s;
    }
};

}  // namespace

extern "C" {

const char *cleora_host_last_error(void) { return g_err.c_str(); }

uint64_t cleora_xxh64(const void *data, uint64_t len, uint64_t seed) {
    return xxh64(static_cast<const uint8_t *>(data), len, seed);
}

int cleora_host_build_from_lines(const char *data, const uint64_t *offsets, uint64_t n_lines,
                                 const char *columns, uint32_t trim_n, cleora_hostgraph **out) {
    if (!out || (n_lines && (!data || !offsets))) { g_err = "NULL argument"; return -1; }
    Builder b;
    if (!make_builder(b, columns, trim_n)) return -1;
    std::vector<std::string_view> lines(n_lines);
    for (uint64_t i = 0; i < n_lines; ++i) lines[i] = std::string_view(data + offsets[i], offsets[i + 1] - offsets[i]);
    *out = b.build(lines);
    return 0;
}

int cleora_host_build_from_files(const char *const *paths, uint64_t n_paths, const char *columns,
                                 uint32_t trim_n, cleora_hostgraph **out) {
    if (!out || !paths) { g_err = "NULL argument"; return -1; }
    Builder b;
    if (!make_builder(b, columns, trim_n)) return -1;
    std::vector<std::string> blobs;
    std::vector<std::string_view> lines;
    blobs.reserve(n_paths);
    for (uint64_t i = 0; i < n_paths; ++i) {
        std::ifstream f(paths[i], std::ios::binary);
        if (!f) continue;  // read_file logs and skips (src/pipeline.rs:193-199)
        blobs.emplace_back((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        const std::string &blob = blobs.back();
        size_t pos = 0;
        while (pos < blob.size()) {
            size_t q = blob.find('\n', pos);
            if (q == std::string::npos) q = blob.size();
            size_t e = q;
            if (e > pos && blob[e - 1] == '\r') --e;  // BufRead::lines strips "\r\n"
            if (e > pos) lines.emplace_back(blob.data() + pos, e - pos);  // empty lines skipped (:205)
            pos = q + 1;
        }
    }
    *out = b.build(lines);
    return 0;
}

/* 0 = one worker per hardware thread (capped at 64).  The result does not depend on it. */
void cleora_host_set_threads(uint32_t n) { g_threads = n; }

void cleora_host_free(cleora_hostgraph *g) { delete g; }

int cleora_host_empty(cleora_hostgraph **out) {
    if (!out) { g_err = "NULL argument"; return -1; }
    *out = new cleora_hostgraph();
    return 0;
}

int cleora_host_sizes(const cleora_hostgraph *g, uint64_t *n, uint64_t *nnz, uint64_t *ids_bytes) {
    if (!g) { g_err = "NULL graph"; return -1; }
    if (n) *n = g->ids.size();
    if (nnz) *nnz = g->col.size();
    if (ids_bytes) { uint64_t t = 0; for (auto &s : g->ids) t += s.size(); *ids_bytes = t; }
    return 0;
}

int cleora_host_copy(const cleora_hostgraph *g, uint64_t *rowptr, uint32_t *col, float *val_left,
                     float *val_sym, float *row_sum, uint64_t *hashes, uint8_t *column_ids) {
    if (!g) { g_err = "NULL graph"; return -1; }
    auto cp = [](auto *dst, const auto &v) { if (dst && !v.empty()) memcpy(dst, v.data(), v.size() * sizeof(v[0])); };
    cp(rowptr, g->rowptr); cp(col, g->col); cp(val_left, g->val_left); cp(val_sym, g->val_sym);
    cp(row_sum, g->row_sum); cp(hashes, g->hashes); cp(column_ids, g->column_ids);
    return 0;
}

int cleora_host_copy_ids(const cleora_hostgraph *g, char *buf, uint64_t *offsets) {
    if (!g || !offsets) { g_err = "NULL argument"; return -1; }
    uint64_t pos = 0;
    for (size_t i = 0; i < g->ids.size(); ++i) {
        offsets[i] = pos;
        if (buf && !g->ids[i].empty()) memcpy(buf + pos, g->ids[i].data(), g->ids[i].size());
        pos += g->ids[i].size();
    }
    offsets[g->ids.size()] = pos;
    return 0;
}

int cleora_host_descriptor(const cleora_hostgraph *g, uint8_t *a_id, const char **a_name,
                           uint8_t *b_id, const char **b_name) {
    if (!g) { g_err = "NULL graph"; return -1; }
    if (a_id) *a_id = g->desc.a_id;
    if (b_id) *b_id = g->desc.b_id;
    if (a_name) *a_name = g->desc.a_name.c_str();
    if (b_name) *b_name = g->desc.b_name.c_str();
    return 0;
}

int cleora_host_set_ids(cleora_hostgraph *g, const char *buf, const uint64_t *offsets, uint64_t n) {
    if (!g || (n && !offsets)) { g_err = "NULL argument"; return -1; }
    // every per-entity array (row sums, hashes, column ids, rowptr) is indexed by entity: a list of another length
    // would leave them inconsistent (the reference's setter accepts it and then fails on the next propagate)
    if (n + 1 != g->rowptr.size() && !(n == 0 && g->rowptr.empty())) {
        g_err = "entity_ids must have one id per entity (" + std::to_string(g->rowptr.empty() ? 0 : g->rowptr.size() - 1) + ")";
        return -1;
    }
    g->ids.resize(n);
    g->hashes.resize(n);
    for (uint64_t i = 0; i < n; ++i) {
        g->ids[i].assign(buf + offsets[i], offsets[i + 1] - offsets[i]);
        // initialize_deterministically hashes the CURRENT id strings (src/lib.rs:75)
        g->hashes[i] = xxh64(reinterpret_cast<const uint8_t *>(g->ids[i].data()), g->ids[i].size(), 0);
    }
    return 0;
}

int cleora_host_serialize(const cleora_hostgraph *g, uint8_t **bytes, uint64_t *len) {
    if (!g || !bytes || !len) { g_err = "NULL argument"; return -1; }
    Writer w;
    w.put<uint8_t>(g->desc.a_id); w.str(g->desc.a_name); w.put<uint8_t>(g->desc.b_id); w.str(g->desc.b_name);
    w.put<uint64_t>(g->ids.size());
    for (auto &s : g->ids) w.str(s);
    w.put<uint64_t>(g->row_sum.size());
    for (float v : g->row_sum) w.put<float>(v);
    w.put<uint64_t>(g->col.size());
    for (size_t k = 0; k < g->col.size(); ++k) { w.put<uint32_t>(g->col[k]); w.put<float>(g->val_left[k]); w.put<float>(g->val_sym[k]); }
    // slices: one (start, end) per row that has edges (src/sparse_matrix_builder.rs:294-304)
    uint64_t nslices = 0;
    for (size_t r = 0; r + 1 < g->rowptr.size(); ++r) nslices += g->rowptr[r + 1] > g->rowptr[r];
    w.put<uint64_t>(nslices);
    for (size_t r = 0; r + 1 < g->rowptr.size(); ++r)
        if (g->rowptr[r + 1] > g->rowptr[r]) { w.put<uint64_t>(g->rowptr[r]); w.put<uint64_t>(g->rowptr[r + 1]); }
    w.put<uint64_t>(g->column_ids.size());
    for (uint8_t c : g->column_ids) w.put<uint8_t>(c);
    *len = w.buf.size();
    *bytes = static_cast<uint8_t *>(malloc(w.buf.size() ? w.buf.size() : 1));
    if (!*bytes) { g_err = "Serialization failed: out of memory"; return -1; }
    memcpy(*bytes, w.buf.data(), w.buf.size());
    return 0;
}

int cleora_host_deserialize(const uint8_t *bytes, uint64_t len, cleora_hostgraph **out) {
    if (!bytes || !out) { g_err = "NULL argument"; return -1; }
    Reader r{bytes, bytes + len};
    auto g = new cleora_hostgraph();
    auto fail = [&](const char *what) { delete g; g_err = std::string("Deserialization failed: ") + what; return -1; };
    g->desc.a_id = r.get<uint8_t>(); g->desc.a_name = r.str(); g->desc.b_id = r.get<uint8_t>(); g->desc.b_name = r.str();
    uint64_t n = r.get<uint64_t>();
    if (!r.utf8_ok) return fail("invalid utf-8 in the matrix descriptor");
    if (!r.ok || n > len) return fail("entity_ids");
    g->ids.resize(n); g->hashes.resize(n);
    for (uint64_t i = 0; i < n && r.ok; ++i) {
        g->ids[i] = r.str();
        g->hashes[i] = xxh64(reinterpret_cast<const uint8_t *>(g->ids[i].data()), g->ids[i].size(), 0);
    }
    if (!r.utf8_ok) return fail("invalid utf-8 sequence in an entity id");
    uint64_t ne = r.get<uint64_t>();
    if (!r.ok || ne > len) return fail("entities");
    if (ne != n) return fail("entities and entity_ids differ in length");
    g->row_sum.resize(ne);
    for (uint64_t i = 0; i < ne; ++i) g->row_sum[i] = r.get<float>();
    uint64_t nnz = r.get<uint64_t>();
    if (!r.ok || nnz > len) return fail("edges");
    g->col.resize(nnz); g->val_left.resize(nnz); g->val_sym.resize(nnz);
    for (uint64_t k = 0; k < nnz; ++k) { g->col[k] = r.get<uint32_t>(); g->val_left[k] = r.get<float>(); g->val_sym[k] = r.get<float>(); }
    uint64_t ns = r.get<uint64_t>();
    if (!r.ok || ns > len) return fail("slices");
    // The reference zips slices positionally with rows (src/embedding.rs:59-63): slice i is row i.
    g->rowptr.assign(n + 1, 0);
    uint64_t prev_end = 0;
    for (uint64_t i = 0; i < ns; ++i) {
        uint64_t s = r.get<uint64_t>(), e = r.get<uint64_t>();
        if (!r.ok || s != prev_end || e < s || e > nnz || i >= n) return fail("slices are not a partition of the edges");
        g->rowptr[i + 1] = e;
        prev_end = e;
    }
    for (uint64_t i = ns; i < n; ++i) g->rowptr[i + 1] = prev_end;
    if (prev_end != nnz) return fail("slices do not cover the edges");
    uint64_t nc = r.get<uint64_t>();
    if (!r.ok || nc > len) return fail("column_ids");
    if (nc != n) return fail("column_ids and entity_ids differ in length");
    g->column_ids.resize(nc);
    for (uint64_t i = 0; i < nc; ++i) g->column_ids[i] = r.get<uint8_t>();
    if (!r.ok) return fail("truncated input");
    // bytes behind the last field are ignored like the reference does: bincode 1.3.3's free function `deserialize` is
    // DefaultOptions + fixint + allow_trailing_bytes (src/lib.rs:470 calls exactly that function)
    for (uint32_t c : g->col) if (c >= n) return fail("edge column out of range");
    *out = g;
    return 0;
}

void cleora_host_free_bytes(uint8_t *bytes) { free(bytes); }

}  // extern "C"
