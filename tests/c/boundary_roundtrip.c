/*
 * boundary_roundtrip.c — a plain C program on the drop-in boundary: includes include/xsi_hip.h, links
 * libxsi_hip.so, writes an .xsi with xsi_writer_* (what XsiFactoryInterface::append / finalize_file do,
 * include/xsi_factory.hpp:38-46), reads it back with xsi_accessor_get_genotypes (malloc-on-NULL like
 * Accessor::get_genotypes, include/accessor.hpp:58-67) and checks every value.  Also the honest way to
 * time the per-line boundary: no interpreter between the caller and the library.
 *
 *   usage: boundary_roundtrip <out.xsi> [n_samples] [n_lines] [block_len] [zstd_level]
 * Prints one line: "ok lines=.. haps=.. write_cells_per_s=.. read_cells_per_s=.." ; exit code 0 on success.
 */
#define _POSIX_C_SOURCE 199309L /* clock_gettime */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "xsi_hip.h"

static uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

/* genotype of (line, haplotype): founder-copy pattern with a per-line allele frequency, phased */
static int32_t gt_value(uint64_t line, uint32_t h, uint32_t n_haps) {
    const uint64_t r = mix64(line * 0x9E3779B97F4A7C15ull + 12345);
    const uint32_t freq = (uint32_t)(r % 997);                 /* per mille-ish */
    const uint64_t f = mix64((line >> 9) * 1315423911ull + (h % 61)) ^ mix64(h * 2654435761ull + line);
    const int alt = (uint32_t)(f % 1000) < freq % 500;
    (void)n_haps;
    return ((alt + 1) << 1) | (h & 1);
}

static double now_s(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

#define CHECK(call)                                                              \
    do {                                                                         \
        long long _rc = (long long)(call);                                       \
        if (_rc < 0) {                                                           \
            fprintf(stderr, "%s failed: %s\n", #call, xsi_hip_last_error());     \
            return 1;                                                            \
        }                                                                        \
    } while (0)

int main(int argc, char** argv) {
    if (argc < 2) {
        fprintf(stderr, "usage: %s out.xsi [n_samples] [n_lines] [block_len] [zstd_level]\n", argv[0]);
        return 2;
    }
    const uint32_t n_samples = argc > 2 ? (uint32_t)atoi(argv[2]) : 2504;
    const uint64_t n_lines = argc > 3 ? (uint64_t)atoll(argv[3]) : 20000;
    const uint32_t block_len = argc > 4 ? (uint32_t)atoi(argv[4]) : 8192;
    const uint32_t n_haps = 2 * n_samples;
    if (xsi_hip_abi_version() != XSI_HIP_ABI_VERSION) return 3;

    xsi_hip_ctx* ctx = NULL;
    CHECK(xsi_hip_ctx_create(&ctx, 0, NULL));
    xsi_encode_params p;
    memset(&p, 0, sizeof(p));
    p.n_samples = n_samples;
    p.block_len = block_len;
    p.mac_threshold = xsi_mac_threshold(n_samples, 2, 0.001);
    p.zstd_level = argc > 5 ? (uint32_t)atoi(argv[5]) : 0; /* --zstd --zl <level>: the outer block layer (interfaces.hpp:288-315) */
    /* rows are generated up front so that the timed loops contain the boundary calls only */
    int32_t* rows = (int32_t*)malloc((size_t)n_lines * n_haps * sizeof(int32_t));
    if (!rows) return 4;
    for (uint64_t l = 0; l < n_lines; ++l)
        for (uint32_t h = 0; h < n_haps; ++h) rows[l * n_haps + h] = gt_value(l, h, n_haps);
    {
        const int32_t* first[3] = {rows, rows + n_haps, rows + 2 * (size_t)n_haps};
        const uint32_t ngt3[3] = {n_haps, n_haps, n_haps};
        const int32_t dp = xsi_default_phased(first, ngt3, n_lines < 3 ? (uint32_t)n_lines : 3, n_samples);
        CHECK(dp);
        p.default_phased = dp;
    }
    char** names = (char**)malloc(sizeof(char*) * n_samples);
    for (uint32_t i = 0; i < n_samples; ++i) {
        names[i] = (char*)malloc(16);
        snprintf(names[i], 16, "S%u", i);
    }
    xsi_writer* w = NULL;
    CHECK(xsi_writer_open(&w, ctx, argv[1], &p, (const char* const*)names));
    double t0 = now_s();
    for (uint64_t l = 0; l < n_lines; ++l) CHECK(xsi_writer_append(w, rows + l * n_haps, n_haps, 2));
    CHECK(xsi_writer_finalize(w, 0));
    const double t_w = now_s() - t0;
    xsi_writer_close(w);

    /* the same file through the zero-copy form (rows written into the writer's own staging), byte for byte;
     * then once more with an in-place producer whose source is one cache-resident row: the rate of the
     * boundary itself, without the DRAM-to-DRAM copy of the per-line append */
    double t_zc = 0.0, t_hot = 0.0;
    {
        char path2[1024];
        snprintf(path2, sizeof(path2), "%s.zc", argv[1]);
        CHECK(xsi_writer_open(&w, ctx, path2, &p, (const char* const*)names));
        t0 = now_s();
        for (uint64_t l = 0; l < n_lines; ++l) {
            int32_t* dst = xsi_writer_row_buffer(w);
            if (!dst) {
                fprintf(stderr, "xsi_writer_row_buffer failed: %s\n", xsi_hip_last_error());
                return 1;
            }
            memcpy(dst, rows + l * n_haps, (size_t)n_haps * 4);
            CHECK(xsi_writer_commit_row(w, n_haps, 2));
        }
        CHECK(xsi_writer_finalize(w, 0));
        t_zc = now_s() - t0;
        xsi_writer_close(w);
        FILE* fa = fopen(argv[1], "rb");
        FILE* fb = fopen(path2, "rb");
        if (!fa || !fb) return 7;
        int same = 1;
        for (;;) {
            unsigned char ba[65536], bb[65536];
            const size_t na = fread(ba, 1, sizeof(ba), fa), nb = fread(bb, 1, sizeof(bb), fb);
            if (na != nb || memcmp(ba, bb, na)) same = 0;
            if (na < sizeof(ba) || !same) break;
        }
        fclose(fa);
        fclose(fb);
        remove(path2);
        if (!same) {
            fprintf(stderr, "zero-copy file differs from the append file\n");
            return 8;
        }
        CHECK(xsi_writer_open(&w, ctx, path2, &p, (const char* const*)names));
        t0 = now_s();
        for (uint64_t l = 0; l < n_lines; ++l) {
            int32_t* dst = xsi_writer_row_buffer(w);
            if (!dst) return 1;
            memcpy(dst, rows + (l & 1) * n_haps, (size_t)n_haps * 4); /* two hot rows */
            CHECK(xsi_writer_commit_row(w, n_haps, 2));
        }
        CHECK(xsi_writer_finalize(w, 0));
        t_hot = now_s() - t0;
        xsi_writer_close(w);
        remove(path2);
    }

    if (xsi_file_num_samples(argv[1]) != (int64_t)n_samples) {
        fprintf(stderr, "xsi_file_num_samples mismatch\n");
        return 5;
    }
    xsi_accessor* a = NULL;
    CHECK(xsi_accessor_open(&a, ctx, argv[1]));
    if (xsi_accessor_num_samples(a) != n_samples || strcmp(xsi_accessor_sample_name(a, n_samples - 1), names[n_samples - 1])) return 6;
    void* gt = NULL; /* allocated by the first call, like bcf_get_genotypes / Accessor::get_genotypes */
    int ngt_arr = 0;
    xsi_bm_state bm;
    xsi_bm_init(&bm);
    uint64_t bad = 0;
    /* every value of every line first (not timed: the comparison reads the 4 N-byte source row from DRAM, which
     * costs more than the call it checks) */
    for (uint64_t l = 0; l < n_lines; ++l) {
        const int64_t pos = xsi_bm_next(&bm, block_len, 2);
        CHECK(pos);
        const int64_t n = xsi_accessor_get_genotypes(a, 2, (uint64_t)pos, &gt, &ngt_arr);
        CHECK(n);
        if ((uint32_t)n != n_haps || (uint32_t)ngt_arr != n_haps || memcmp(gt, rows + l * n_haps, (size_t)n_haps * 4)) ++bad;
    }
    /* then the rate of the calls themselves, with the same light check as the view loop below */
    xsi_bm_init(&bm);
    t0 = now_s();
    for (uint64_t l = 0; l < n_lines; ++l) {
        const int64_t pos = xsi_bm_next(&bm, block_len, 2);
        const int64_t n = xsi_accessor_get_genotypes(a, 2, (uint64_t)pos, &gt, &ngt_arr);
        CHECK(n);
        const int32_t* g = (const int32_t*)gt;
        if ((uint32_t)n != n_haps || g[0] != rows[l * n_haps] || g[n_haps - 1] != rows[l * n_haps + n_haps - 1]) ++bad;
    }
    const double t_r = now_s() - t0;
    free(gt);
    /* the same through the view (no copy into a caller array) */
    xsi_bm_init(&bm);
    t0 = now_s();
    for (uint64_t l = 0; l < n_lines; ++l) {
        const int64_t pos = xsi_bm_next(&bm, block_len, 2);
        const int32_t* v = NULL;
        const int64_t n = xsi_accessor_genotypes_view(a, 2, (uint64_t)pos, &v);
        CHECK(n);
        if ((uint32_t)n != n_haps || v[0] != rows[l * n_haps] || v[n_haps - 1] != rows[l * n_haps + n_haps - 1]) ++bad;
    }
    const double t_v = now_s() - t0;
    /* and through the batch call into a page-locked array of the accessor's: 256 consecutive lines per call, the rows
     * stored by the compose kernels themselves (no window, no host copy) */
    enum { BATCH = 256 };
    int32_t* rows_b = NULL;
    CHECK(xsi_accessor_alloc_array(a, (uint64_t)BATCH * n_haps, &rows_b));
    CHECK(xsi_accessor_register_array(a, rows_b, (uint64_t)BATCH * n_haps));
    uint32_t na_b[BATCH];
    uint64_t pos_b[BATCH];
    xsi_bm_init(&bm);
    t0 = now_s();
    for (uint64_t l0 = 0; l0 < n_lines; l0 += BATCH) {
        const uint64_t m = n_lines - l0 < BATCH ? n_lines - l0 : BATCH;
        for (uint64_t k = 0; k < m; ++k) {
            na_b[k] = 2;
            pos_b[k] = (uint64_t)xsi_bm_next(&bm, block_len, 2);
        }
        const int64_t n = xsi_accessor_get_genotypes_batch(a, m, na_b, pos_b, rows_b, n_haps, NULL);
        CHECK(n);
        if ((uint64_t)n != m * n_haps) ++bad;
        for (uint64_t k = 0; k < m; k += 37) {
            const int32_t* g = rows_b + k * n_haps;
            if (g[0] != rows[(l0 + k) * n_haps] || g[n_haps - 1] != rows[(l0 + k) * n_haps + n_haps - 1]) ++bad;
        }
    }
    const double t_b = now_s() - t0;
    CHECK(xsi_accessor_free_array(a, rows_b));
    xsi_accessor_close(a);
    xsi_hip_ctx_destroy(ctx);
    const double cells = (double)n_lines * n_haps;
    printf("%s lines=%llu haps=%u bad_lines=%llu write_cells_per_s=%.4g zero_copy_write_cells_per_s=%.4g "
           "zero_copy_hot_source_cells_per_s=%.4g read_cells_per_s=%.4g view_read_cells_per_s=%.4g batch_read_cells_per_s=%.4g\n",
           bad ? "MISMATCH" : "ok", (unsigned long long)n_lines, n_haps, (unsigned long long)bad, cells / t_w, cells / t_zc,
           cells / t_hot, cells / t_r, cells / t_v, cells / t_b);
    return bad ? 7 : 0;
}
