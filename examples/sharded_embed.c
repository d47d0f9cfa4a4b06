/*
 * sharded_embed.c — the ROW-PARTITIONED propagation loop of BASELINE.json:north_star driven by a host with no Python
 * and no torch: one process per GPU, plain C99 over the two C ABIs (include/cleora_host.h, include/cleora_hip.h).
 * It is what the reference's Rust crate would do through the extern "C" block of INTEGRATION.md — three calls:
 *
 *   cleora_comm_create / cleora_comm_create_local   the communicator (RCCL over xGMI, or the peer-direct hipIpc transport)
 *   cleora_sharded_create                           this rank's row blocks of the CSR on its GPU (every rank builds the same
 *                                                   graph on its host: the build is deterministic)
 *   cleora_embed_sharded                            the loop: per iteration and block, SpMM + fused L2 straight into the
 *                                                   block's slot of the next replica, in-place all-gather of the step's slots on
 *                                                   a communication stream beside the next block's SpMM (csrc/sharded.hip);
 *                                                   with --whiten the default embed() loop (pycleora/__init__.py:109-117)
 *   the unique id travels from rank 0 to the others through a file (any out-of-band channel does).
 *
 *   sharded_embed [--local] [--whiten] <rank> <world> <id-file> <columns> <dim> <iterations> <out.tsv> <edges.tsv> [more files]
 *
 * Start one process per rank (rank r uses GPU r mod #GPUs).  --local: the peer-direct transport without RCCL (several ranks
 * may then share one GPU).  Rank 0 writes the embeddings (without --whiten: the result of embed_fast, src/lib.rs:320-364, bit
 * for bit for rows that are not split).  Exit codes as embed_file.c.
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
    int local = 0, whiten = 0;
    while (argc > 1 && argv[1][0] == '-' && argv[1][1] == '-') {
        if (strcmp(argv[1], "--local") == 0) local = 1;
        else if (strcmp(argv[1], "--whiten") == 0) whiten = 1;
        else { fprintf(stderr, "sharded_embed: unknown option %s\n", argv[1]); return 1; }
        ++argv; --argc;
    }
    if (argc < 9) {
        fprintf(stderr, "usage: sharded_embed [--local] [--whiten] <rank> <world> <id-file> <columns> <dim> <iterations> <out.tsv> <edges> [...]\n");
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
        if ((local ? cleora_comm_local_id(id) : cleora_comm_unique_id(id)) != CLEORA_OK) return fail_dev("cleora_comm_unique_id");
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
    if ((local ? cleora_comm_create_local(id, rank, world, device, &comm) : cleora_comm_create(id, rank, world, device, &comm)) != CLEORA_OK)
        return fail_dev("cleora_comm_create");

    /* 3. this rank's row blocks (block-cyclic: P * STEPS contiguous blocks, rank r owns blocks {k * P + r}) */
    cleora_sharded *sh = NULL;
    if (cleora_sharded_create(comm, device, n, nnz, rowptr, col, val, NULL, 0, STEPS, CLEORA_BALANCE_AUTO, &sh) != CLEORA_OK)
        return fail_dev("cleora_sharded_create");
    cleora_sharded_info info;
    cleora_sharded_get_info(sh, &info);

    /* 4. the replica of the iterate, deterministic start (initialize_deterministically, src/lib.rs:242-252); rows >= n zero */
    const uint64_t bytes = info.n_pad * (uint64_t)dim * sizeof(float);
    void *xv = NULL, *d_hash = NULL;
    if (cleora_malloc(bytes, &xv) != CLEORA_OK || cleora_malloc((n ? n : 1) * sizeof(uint64_t), &d_hash) != CLEORA_OK) return fail_dev("cleora_malloc");
    float *x = xv;
    cleora_memset(x, 0, bytes, NULL);
    cleora_memcpy_h2d(d_hash, hashes, n * sizeof(uint64_t), NULL);
    if (n && cleora_init_dev(d_hash, n, dim, 0, x, dim, NULL) != CLEORA_OK) return fail_dev("cleora_init_dev");

    /* 5. the loop: one call */
    uint64_t ran = 0;
    if (cleora_embed_sharded(sh, x, CLEORA_LEFT, dim, iterations, 0.0f, 0.0f, whiten ? CLEORA_F_WHITEN : 0u, &ran) != CLEORA_OK)
        return fail_dev("cleora_embed_sharded");
    /* 6. rank 0 writes the result */
    int rc = 0;
    if (rank == 0) {
        float *emb = malloc((n ? n : 1) * (size_t)dim * sizeof *emb);
        char *ids = malloc(ids_bytes ? ids_bytes : 1);
        uint64_t *id_off = malloc((n + 1) * sizeof *id_off);
        if (!emb || !ids || !id_off) return 4;
        if (cleora_memcpy_d2h(emb, x, n * (uint64_t)dim * sizeof(float), NULL) != CLEORA_OK) return fail_dev("cleora_memcpy_d2h");
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
    }
    int algo = -1;
    cleora_comm_get_allgather(comm, &algo);
    fprintf(stderr, "sharded_embed: rank %d of %d: %llu entities, %llu rows in %u blocks (%s-balanced), %llu iterations%s, all-gather: %s\n", rank, world,
            (unsigned long long)n, (unsigned long long)info.local_rows, info.steps, info.balance == CLEORA_BALANCE_NNZ ? "work" : "row",
            (unsigned long long)ran, whiten ? " of the whitened loop" : "",
            world == 1 ? "none (one rank)" : algo == CLEORA_ALLGATHER_PEER ? "peer-direct stores (hipIpc)" : algo == CLEORA_ALLGATHER_P2P ? "send/recv mesh" : "ncclAllGather (ring)");
    cleora_sharded_destroy(sh);
    cleora_comm_destroy(comm);
    cleora_free(xv); cleora_free(d_hash);
    cleora_host_free(hg);
    free(rowptr); free(hashes); free(col); free(val);
    return rc;
}
