/*
 * sharded_embed.c — the ROW-PARTITIONED propagation loop of BASELINE.json:north_star driven by a host with no Python
 * and no torch: one process per GPU, plain C99 over the two C ABIs (include/cleora_host.h, include/cleora_hip.h).
 * It is the loop cleora_amd/sharded.py runs, written the way the reference's Rust crate would write it through the
 * extern "C" block of INTEGRATION.md:
 *
 *   every rank builds the graph (CPU), keeps a full replica of the iterate X and owns K row blocks of the CSR
 *   (block-cyclic: block j = rows [j*B, (j+1)*B), rank r owns blocks {k*P + r});
 *   per iteration, for k = 0..K-1:   SpMM + fused L2 of block (k, r) straight into its slot of X_next
 *                                    (cleora_propagate_dev on the compute stream)
 *                                    in-place all-gather of step k's P slots (cleora_allgatherv_f32_dev on the
 *                                    communication stream, ordered with cleora_stream_wait_stream) — it runs beside
 *                                    the SpMM of block k+1;
 *   the RCCL unique id travels from rank 0 to the others through a file (any out-of-band channel does).
 *
 *   sharded_embed <rank> <world> <id-file> <columns> <dim> <iterations> <out.tsv> <edges.tsv> [more files]
 *
 * Start one process per rank (rank r uses GPU r).  Rank 0 writes the embeddings (the result of embed_fast,
 * src/lib.rs:320-364, bit for bit for rows that are not split).  Exit codes as embed_file.c.
 */
#define _DEFAULT_SOURCE /* usleep */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "cleora_hip.h"
#include "cleora_host.h"

#define STEPS 2 /* row blocks per rank and iteration */

static int fail_dev(const char *what) {
    fprintf(stderr, "sharded_embed: %s: %s\n", what, cleora_last_error());
    return 3;
}

int main(int argc, char **argv) {
    if (argc < 9) {
        fprintf(stderr, "usage: sharded_embed <rank> <world> <id-file> <columns> <dim> <iterations> <out.tsv> <edges> [...]\n");
        return 1;
    }
    const int rank = atoi(argv[1]), world = atoi(argv[2]);
    const char *id_path = argv[3], *columns = argv[4];
    const uint32_t dim = (uint32_t)strtoul(argv[5], NULL, 10);
    const uint64_t iterations = strtoull(argv[6], NULL, 10);
    const char *out_path = argv[7];
    if (world < 1 || rank < 0 || rank >= world || dim == 0) { fprintf(stderr, "sharded_embed: bad rank / world / dim\n"); return 1; }

    /* 1. the graph, on every rank (deterministic: identical CSR everywhere) */
    cleora_hostgraph *hg = NULL;
    if (cleora_host_build_from_files((const char *const *)&argv[8], (uint64_t)(argc - 8), columns, 16, &hg) != 0) {
        fprintf(stderr, "sharded_embed: %s\n", cleora_host_last_error());
        return 2;
    }
    uint64_t n = 0, nnz = 0, ids_bytes = 0;
    cleora_host_sizes(hg, &n, &nnz, &ids_bytes);
    uint64_t *rowptr = malloc((n + 1) * sizeof *rowptr), *hashes = malloc((n ? n : 1) * sizeof *hashes);
    uint32_t *col = malloc((nnz ? nnz : 1) * sizeof *col);
    float *val = malloc((nnz ? nnz : 1) * sizeof *val);
    if (!rowptr || !hashes || !col || !val) return 4;
    cleora_host_copy(hg, rowptr, col, val, NULL, NULL, hashes, NULL);

    /* 2. device, communicator */
    int n_dev = 0;
    cleora_device_count(&n_dev);
    if (n_dev < 1) { fprintf(stderr, "sharded_embed: no HIP device visible; libcleora_hip has no CPU fallback\n"); return 3; }
    const int device = rank % n_dev;
    if (cleora_set_device(device) != CLEORA_OK) return fail_dev("cleora_set_device");
    unsigned char id[CLEORA_COMM_ID_BYTES];
    if (rank == 0) {
        if (cleora_comm_unique_id(id) != CLEORA_OK) return fail_dev("cleora_comm_unique_id");
        char tmp[4096];
        snprintf(tmp, sizeof tmp, "%s.tmp", id_path);
        FILE *f = fopen(tmp, "wb");
        if (!f || fwrite(id, 1, sizeof id, f) != sizeof id || fclose(f) != 0 || rename(tmp, id_path) != 0) { perror(id_path); return 4; }
    } else {
        FILE *f = NULL;
        for (int tries = 0; tries < 600 && !(f = fopen(id_path, "rb")); ++tries) usleep(100000);
        if (!f || fread(id, 1, sizeof id, f) != sizeof id) { fprintf(stderr, "sharded_embed: no unique id in %s\n", id_path); return 4; }
        fclose(f);
    }
    cleora_comm *comm = NULL;
    if (cleora_comm_create(id, rank, world, device, &comm) != CLEORA_OK) return fail_dev("cleora_comm_create");
    void *compute = NULL, *comms = NULL;
    if (cleora_stream_create(&compute) != CLEORA_OK || cleora_stream_create(&comms) != CLEORA_OK) return fail_dev("cleora_stream_create");

    /* 3. this rank's row blocks: block j = rows [j*B, (j+1)*B) of the padded row space, B a multiple of 4 */
    const uint64_t nb = (uint64_t)world * STEPS;
    uint64_t B = (n + nb - 1) / nb;
    B = (B + 3) / 4 * 4;
    if (B == 0) B = 4;
    const uint64_t n_pad = B * nb;
    cleora_graph *blocks[STEPS];
    uint64_t *brp = malloc((B + 1) * sizeof *brp);
    if (!brp) return 4;
    for (int k = 0; k < STEPS; ++k) {
        const uint64_t b0 = ((uint64_t)k * world + rank) * B;
        const uint64_t r0 = b0 < n ? b0 : n, r1 = b0 + B < n ? b0 + B : n;
        const uint64_t e0 = rowptr[r0], e1 = rowptr[r1];
        for (uint64_t i = 0; i <= B; ++i) brp[i] = (r0 + i <= r1 ? rowptr[r0 + i] : e1) - e0;   /* padding rows are empty */
        if (cleora_graph_create(device, B, n_pad, e1 - e0, brp, col + e0, val + e0, NULL, 0, 0, &blocks[k]) != CLEORA_OK)
            return fail_dev("cleora_graph_create");
    }

    /* 4. replicas of the iterate, deterministic start (initialize_deterministically, src/lib.rs:242-252) */
    const uint64_t bytes = n_pad * (uint64_t)dim * sizeof(float);
    void *bufs[2], *d_hash = NULL;
    if (cleora_alloc_iterates(blocks[0], dim, 2, bufs, NULL) != CLEORA_OK) return fail_dev("cleora_alloc_iterates");
    float *x = bufs[0], *x_next = bufs[1];
    if (cleora_malloc((n ? n : 1) * sizeof(uint64_t), &d_hash) != CLEORA_OK) return fail_dev("cleora_malloc");
    cleora_memset(x, 0, bytes, compute);
    cleora_memset(x_next, 0, bytes, compute);
    cleora_memcpy_h2d(d_hash, hashes, n * sizeof(uint64_t), compute);
    if (n && cleora_init_dev(d_hash, n, dim, 0, x, dim, compute) != CLEORA_OK) return fail_dev("cleora_init_dev");

    /* 5. the loop */
    uint64_t offsets[4097];
    if (world > 4096) return 1;
    for (uint64_t it = 0; it < iterations; ++it) {
        for (int k = 0; k < STEPS; ++k) {
            const uint64_t g0 = (uint64_t)k * world * B, mine = g0 + (uint64_t)rank * B;
            if (cleora_propagate_dev(blocks[k], CLEORA_LEFT, x, dim, dim, x_next + mine * dim, dim, CLEORA_F_L2NORM, 0.0f,
                                     x + mine * dim, NULL, NULL, compute) != CLEORA_OK)
                return fail_dev("cleora_propagate_dev");
            /* the exchange of step k starts when block (k, r) is written and runs beside the SpMM of block k+1 */
            if (cleora_stream_wait_stream(comms, compute) != CLEORA_OK) return fail_dev("cleora_stream_wait_stream");
            for (int r = 0; r <= world; ++r) offsets[r] = (g0 + (uint64_t)r * B) * dim;
            if (cleora_allgatherv_f32_dev(comm, x_next, offsets, comms) != CLEORA_OK) return fail_dev("cleora_allgatherv_f32_dev");
        }
        if (cleora_stream_wait_stream(compute, comms) != CLEORA_OK) return fail_dev("cleora_stream_wait_stream");
        float *t = x; x = x_next; x_next = t;
    }

    /* 6. rank 0 writes the result */
    int rc = 0;
    if (rank == 0) {
        float *emb = malloc((n ? n : 1) * (size_t)dim * sizeof *emb);
        char *ids = malloc(ids_bytes ? ids_bytes : 1);
        uint64_t *id_off = malloc((n + 1) * sizeof *id_off);
        if (!emb || !ids || !id_off) return 4;
        if (cleora_memcpy_d2h(emb, x, n * (uint64_t)dim * sizeof(float), compute) != CLEORA_OK) return fail_dev("cleora_memcpy_d2h");
        cleora_host_copy_ids(hg, ids, id_off);
        FILE *f = fopen(out_path, "w");
        if (!f) { perror(out_path); return 4; }
        for (uint64_t i = 0; i < n; ++i) {
            fwrite(ids + id_off[i], 1, (size_t)(id_off[i + 1] - id_off[i]), f);
            fputc('\t', f);
            for (uint32_t c = 0; c < dim; ++c) fprintf(f, c ? " %.9g" : "%.9g", (double)emb[i * (uint64_t)dim + c]);
            fputc('\n', f);
        }
        if (fclose(f) != 0) { perror(out_path); rc = 4; }
        free(emb); free(ids); free(id_off);
    } else {
        cleora_stream_sync(compute);
    }
    int algo = -1;
    cleora_comm_get_allgather(comm, &algo);
    fprintf(stderr, "sharded_embed: rank %d of %d: %llu entities, %llu iterations, all-gather: %s\n", rank, world, (unsigned long long)n,
            (unsigned long long)iterations,
            world == 1 ? "none (one rank)" : algo == CLEORA_ALLGATHER_P2P ? "send/recv mesh (CLEORA_ALLGATHER=p2p)" : "ncclAllGather (ring)");
    for (int k = 0; k < STEPS; ++k) cleora_graph_destroy(blocks[k]);
    cleora_comm_destroy(comm);
    cleora_stream_destroy(compute);
    cleora_stream_destroy(comms);
    cleora_free(bufs[0]); cleora_free(bufs[1]); cleora_free(d_hash);
    cleora_host_free(hg);
    free(rowptr); free(hashes); free(col); free(val); free(brp);
    return rc;
}
