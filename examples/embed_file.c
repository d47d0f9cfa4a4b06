/*
 * embed_file.c — a host with no Python and no torch: plain C99 over the two C ABIs
 * (include/cleora_host.h, include/cleora_hip.h).  It is what the reference's Rust crate would do through
 * the extern "C" block of INTEGRATION.md: build the graph from edge files on the CPU
 * (SparseMatrix::from_files, src/lib.rs:137-173), hand the CSR to the device library, run the
 * device-resident embed loop (embed_fast, src/lib.rs:320-364; with --whiten the default path of
 * pycleora.embed(), pycleora/__init__.py:97-127) and write `entity_id<TAB>v0 v1 ...` lines.
 *
 *   embed_file [--whiten] [--symmetric] [--devices 0,1,...] <columns> <dim> <iterations> <out.tsv> <edges.tsv> [more files]
 *
 * --devices: the same loop with the graph row-partitioned over the listed devices INSIDE this process (cleora_multi_*,
 * csrc/multi.hip: a host thread per device, peer-direct all-gather of the iterate between iterations); ids may repeat.
 *
 * Exit codes: 0 ok, 1 usage, 2 graph construction failed, 3 device library failed (e.g. no GPU: there is
 * no CPU fallback), 4 I/O.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "cleora_hip.h"
#include "cleora_host.h"

static int fail_dev(const char *what) {
    fprintf(stderr, "embed_file: %s: %s\n", what, cleora_last_error());
    return 3;
}

int main(int argc, char **argv) {
    int whiten = 0, kind = CLEORA_LEFT, a = 1;
    int devices[64];
    uint32_t n_devices = 0;
    while (a < argc && strncmp(argv[a], "--", 2) == 0) {
        if (strcmp(argv[a], "--whiten") == 0) whiten = 1;
        else if (strcmp(argv[a], "--symmetric") == 0) kind = CLEORA_SYMMETRIC;
        else if (strcmp(argv[a], "--devices") == 0 && a + 1 < argc) {
            for (char *tok = strtok(argv[++a], ","); tok && n_devices < 64; tok = strtok(NULL, ",")) devices[n_devices++] = atoi(tok);
            if (n_devices == 0) { fprintf(stderr, "embed_file: --devices needs a list like 0,1,2\n"); return 1; }
        }
        else { fprintf(stderr, "embed_file: unknown option %s\n", argv[a]); return 1; }
        ++a;
    }
    if (argc - a < 5) {
        fprintf(stderr, "usage: embed_file [--whiten] [--symmetric] [--devices 0,1,...] <columns> <dim> <iterations> <out.tsv> <edges> [...]\n");
        return 1;
    }
    const char *columns = argv[a];
    const uint32_t dim = (uint32_t)strtoul(argv[a + 1], NULL, 10);
    const uint64_t iterations = strtoull(argv[a + 2], NULL, 10);
    const char *out_path = argv[a + 3];
    const char *const *files = (const char *const *)&argv[a + 4];
    const uint64_t n_files = (uint64_t)(argc - a - 4);
    if (dim == 0) { fprintf(stderr, "embed_file: dim must be positive\n"); return 1; }

    /* 1. graph on the host (CPU work in the reference too) */
    cleora_hostgraph *hg = NULL;
    if (cleora_host_build_from_files(files, n_files, columns, 16, &hg) != 0) {
        fprintf(stderr, "embed_file: %s\n", cleora_host_last_error());
        return 2;
    }
    uint64_t n = 0, nnz = 0, ids_bytes = 0;
    cleora_host_sizes(hg, &n, &nnz, &ids_bytes);
    uint64_t *rowptr = malloc((n + 1) * sizeof *rowptr), *hashes = malloc((n ? n : 1) * sizeof *hashes);
    uint32_t *col = malloc((nnz ? nnz : 1) * sizeof *col);
    float *val_left = malloc((nnz ? nnz : 1) * sizeof *val_left), *val_sym = malloc((nnz ? nnz : 1) * sizeof *val_sym);
    char *ids = malloc(ids_bytes ? ids_bytes : 1);
    uint64_t *id_off = malloc((n + 1) * sizeof *id_off);
    float *emb = malloc((n ? n : 1) * (size_t)dim * sizeof *emb);
    if (!rowptr || !hashes || !col || !val_left || !val_sym || !ids || !id_off || !emb) {
        fprintf(stderr, "embed_file: out of host memory\n");
        return 4;
    }
    cleora_host_copy(hg, rowptr, col, val_left, val_sym, NULL, hashes, NULL);
    cleora_host_copy_ids(hg, ids, id_off);
    fprintf(stderr, "embed_file: %llu entities, %llu stored edges\n", (unsigned long long)n, (unsigned long long)nnz);

    /* 2. device: CSR upload, the whole loop in HBM, one download */
    int n_dev = 0;
    if (cleora_device_count(&n_dev) != CLEORA_OK) return fail_dev("cleora_device_count");
    if (n_dev < 1) {
        fprintf(stderr, "embed_file: no HIP device visible; libcleora_hip has no CPU fallback\n");
        return 3;
    }
    uint64_t ran = 0;
    if (n_devices > 0) {
        /* one process, several devices: the row partition behind one handle (SURVEY 8b B2's graph handle with device_ids) */
        cleora_multi *m = NULL;
        if (cleora_multi_create(devices, n_devices, n, nnz, rowptr, col, val_left, val_sym, 0, CLEORA_BALANCE_AUTO, &m) != CLEORA_OK)
            return fail_dev("cleora_multi_create");
        if (n > 0 && cleora_multi_embed(m, hashes, NULL, kind, dim, iterations, 0, 0.0f, 0.0f, whiten ? CLEORA_F_WHITEN : 0u, emb, &ran) != CLEORA_OK)
            return fail_dev("cleora_multi_embed");
        cleora_multi_destroy(m);
    } else {
        cleora_graph *g = NULL;
        if (cleora_graph_create(0, n, n, nnz, rowptr, col, val_left, val_sym, 0, 0, &g) != CLEORA_OK)
            return fail_dev("cleora_graph_create");
        if (n > 0 && cleora_embed(g, hashes, NULL, kind, dim, iterations, 0, 0.0f, 0.0f, whiten ? CLEORA_F_WHITEN : 0u,
                                  emb, &ran) != CLEORA_OK)
            return fail_dev("cleora_embed");
        cleora_graph_destroy(g);
    }

    /* 3. output */
    FILE *f = fopen(out_path, "w");
    if (!f) { perror(out_path); return 4; }
    for (uint64_t i = 0; i < n; ++i) {
        fwrite(ids + id_off[i], 1, (size_t)(id_off[i + 1] - id_off[i]), f);
        fputc('\t', f);
        for (uint32_t c = 0; c < dim; ++c) fprintf(f, c ? " %.9g" : "%.9g", (double)emb[i * (uint64_t)dim + c]);
        fputc('\n', f);
    }
    if (fclose(f) != 0) { perror(out_path); return 4; }
    fprintf(stderr, "embed_file: %llu iterations, wrote %s\n", (unsigned long long)ran, out_path);
    cleora_host_free(hg);
    free(rowptr); free(hashes); free(col); free(val_left); free(val_sym); free(ids); free(id_off); free(emb);
    return 0;
}
