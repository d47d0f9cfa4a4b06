/*
 * cleora_host.h — C ABI of libcleora_host.so: the host-side (CPU, no HIP) half of the drop-in,
 * i.e. the parts of pycleora's Rust core that run ONCE per graph and feed the kernels:
 * entity hashing, hypergraph -> CSR construction and the pickle wire format.
 * (SURVEY.md §8f rows N1/N2.  Graph construction is CPU work in the reference too; this is not a
 * fallback for any GPU kernel.)
 *
 * Reference interfaces replaced (paths relative to the reference checkout):
 *   hash_entity                     src/entity.rs:109-114          -> cleora_xxh64
 *   parse_fields / descriptors      src/configuration.rs:19-70, src/sparse_matrix.rs:5-46
 *   build_graph_from_iterator       src/pipeline.rs:24-79,223-240  -> cleora_host_build_from_lines
 *   build_graph_from_files          src/pipeline.rs:81-221         -> cleora_host_build_from_files
 *   SparseMatrixBuffer / Reducer    src/sparse_matrix_builder.rs:153-392
 *   __getstate__/__setstate__       src/lib.rs:463-475 (bincode 1.3.3 of struct SparseMatrix)
 *
 * Determinism: the reference spreads hyperedges over N consumer threads and sums their f32
 * partials in a nondeterministic order; this builder always produces the single-consumer
 * result (hyperedges accumulated in input order), for any thread count.
 */
#ifndef CLEORA_HOST_H
#define CLEORA_HOST_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cleora_hostgraph cleora_hostgraph;

const char *cleora_host_last_error(void);

/* XXH64 (twox-hash 1.6.3 XxHash64 with the given seed; hash_entity uses seed 0). */
uint64_t cleora_xxh64(const void *data, uint64_t len, uint64_t seed);

/* lines: n_lines strings packed back to back in `data`; line i is data[offsets[i] .. offsets[i+1]).
 * columns: the column spec ("complex::reflexive::product", "user product", ...).
 * Returns 0, or -1 with cleora_host_last_error() set (bad column spec, != 1 relation, ...). */
int cleora_host_build_from_lines(const char *data, const uint64_t *offsets, uint64_t n_lines,
                                 const char *columns, uint32_t hyperedge_trim_n,
                                 cleora_hostgraph **out);
/* Files are read in the order given (empty lines skipped, unopenable files skipped like
 * pipeline.rs:193-199).  Extension checking is the caller's job (src/lib.rs:147-158). */
int cleora_host_build_from_files(const char *const *paths, uint64_t n_paths, const char *columns,
                                 uint32_t hyperedge_trim_n, cleora_hostgraph **out);
void cleora_host_free(cleora_hostgraph *g);
/* Worker threads of the builder (0 = hardware threads, max 64).  The graph is identical for any value. */
void cleora_host_set_threads(uint32_t n);

/* n = entities, nnz = stored (directed) edges, ids_bytes = total UTF-8 bytes of all entity ids. */
int cleora_host_sizes(const cleora_hostgraph *g, uint64_t *n, uint64_t *nnz, uint64_t *ids_bytes);
/* Any output pointer may be NULL.  rowptr u64[n+1], col u32[nnz], val_* f32[nnz], row_sum f32[n]
 * (Entity.row_sum), hashes u64[n], column_ids u8[n]. */
int cleora_host_copy(const cleora_hostgraph *g, uint64_t *rowptr, uint32_t *col, float *val_left,
                     float *val_sym, float *row_sum, uint64_t *hashes, uint8_t *column_ids);
/* ids packed into buf[ids_bytes], id i = buf[offsets[i] .. offsets[i+1]); offsets u64[n+1]. */
int cleora_host_copy_ids(const cleora_hostgraph *g, char *buf, uint64_t *offsets);
/* SparseMatrixDescriptor: names are returned as pointers valid until cleora_host_free. */
int cleora_host_descriptor(const cleora_hostgraph *g, uint8_t *col_a_id, const char **col_a_name,
                           uint8_t *col_b_id, const char **col_b_name);
/* Replace the entity ids (the PyO3 `entity_ids` setter, src/sparse_matrix.rs:60-61). */
int cleora_host_set_ids(cleora_hostgraph *g, const char *buf, const uint64_t *offsets, uint64_t n);

/* bincode 1.3.3 (little endian, u64 lengths, fixed-width ints) of struct SparseMatrix in field
 * order: descriptor{u8,String,u8,String}, entity_ids Vec<String>, entities Vec<{f32}>,
 * edges Vec<{u32,f32,f32}>, slices Vec<(u64,u64)>, column_ids Vec<u8>.
 * serialize: *bytes is malloc'ed, release with cleora_host_free_bytes. */
int cleora_host_serialize(const cleora_hostgraph *g, uint8_t **bytes, uint64_t *len);
int cleora_host_deserialize(const uint8_t *bytes, uint64_t len, cleora_hostgraph **out);
void cleora_host_free_bytes(uint8_t *bytes);
/* An empty matrix (SparseMatrix::new with no arguments, src/lib.rs:439-456). */
int cleora_host_empty(cleora_hostgraph **out);

#ifdef __cplusplus
}
#endif
#endif
