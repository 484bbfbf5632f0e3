/*
 * xsi_oracle.c — CPU ORACLE (test infrastructure, NOT product code).
 * See xsi_oracle.h for the rules and for how parity is pinned.
 *
 * Every function cites the reference file:line whose behaviour it restates
 * (paths relative to /root/reference).  Reference quirks are kept on purpose
 *  because the contract is byte-identical .xsi and identical
 * decoded int32 rows.
 */
#include "xsi_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* small helpers                                                       */
/* ------------------------------------------------------------------ */
typedef struct {
    uint8_t* p;
    size_t n, cap;
} buf_t;

static int buf_reserve(buf_t* b, size_t extra) {
    if (b->n + extra <= b->cap) return 0;
    size_t nc = b->cap ? b->cap : 256;
    while (nc < b->n + extra) nc *= 2;
    uint8_t* np = (uint8_t*)realloc(b->p, nc);
    if (!np) return -1;
    b->p = np;
    b->cap = nc;
    return 0;
}
static int buf_put(buf_t* b, const void* src, size_t n) {
    if (buf_reserve(b, n)) return -1;
    if (n) memcpy(b->p + b->n, src, n);
    b->n += n;
    return 0;
}
static int buf_u16(buf_t* b, uint16_t v) { return buf_put(b, &v, 2); }
static int buf_u32(buf_t* b, uint32_t v) { return buf_put(b, &v, 4); }
static int buf_zero(buf_t* b, size_t n) {
    if (buf_reserve(b, n)) return -1;
    memset(b->p + b->n, 0, n);
    b->n += n;
    return 0;
}
static void buf_free(buf_t* b) {
    free(b->p);
    b->p = NULL;
    b->n = b->cap = 0;
}
/* A_T value (2 or 4 bytes, little endian). */
static int buf_at(buf_t* b, uint32_t v, int aet) {
    if (aet == 2) return buf_u16(b, (uint16_t)v);
    return buf_u32(b, v);
}

/* htslib vcf.h macros */
static inline int32_t gt_allele(int32_t v) { return (v >> 1) - 1; }
static inline int gt_is_missing(int32_t v) { return (v >> 1) ? 0 : 1; }
static inline int gt_is_phased(int32_t v) { return v & 1; }
static inline int32_t gt_unphased(int32_t allele) { return (allele + 1) << 1; }

/* ------------------------------------------------------------------ */
/* WAH16 (wah.hpp)                                                     */
/* ------------------------------------------------------------------ */
#define WAH_BITS 15u
#define WAH_HIGH 0x8000u
#define WAH_ONE 0x4000u
#define WAH_MAXC 0x3FFFu
#define WAH_ALLSET 0x7FFFu

typedef struct {
    uint16_t zeros, ones; /* not_set_counter / all_set_counter */
} wah_state;

/* process_wah_word, wah.hpp:376-429 */
static int wah_push_group(buf_t* out, wah_state* s, uint16_t word) {
    if (word == 0) {
        if (s->ones) {
            if (buf_u16(out, (uint16_t)(WAH_HIGH | WAH_ONE | s->ones))) return -1;
            s->ones = 0;
        }
        if (s->zeros == WAH_MAXC) {
            if (buf_u16(out, (uint16_t)(0xFFFFu & ~WAH_ONE))) return -1; /* 0xBFFF */
            s->zeros = 0;
        }
        s->zeros++;
    } else if (word == WAH_ALLSET) {
        if (s->zeros) {
            if (buf_u16(out, (uint16_t)(WAH_HIGH | s->zeros))) return -1;
            s->zeros = 0;
        }
        if (s->ones == WAH_MAXC) {
            if (buf_u16(out, 0xFFFFu)) return -1;
            s->ones = 0;
        }
        s->ones++;
    } else {
        if (s->ones) {
            if (buf_u16(out, (uint16_t)(WAH_HIGH | WAH_ONE | s->ones))) return -1;
            s->ones = 0;
        }
        if (s->zeros) {
            if (buf_u16(out, (uint16_t)(WAH_HIGH | s->zeros))) return -1;
            s->zeros = 0;
        }
        if (buf_u16(out, word)) return -1;
    }
    return 0;
}
/* trailing flush, wah.hpp:330-335 / 491-497 / 567-573: zeros then ones */
static int wah_flush(buf_t* out, wah_state* s) {
    if (s->zeros && buf_u16(out, (uint16_t)(WAH_HIGH | s->zeros))) return -1;
    if (s->ones && buf_u16(out, (uint16_t)(WAH_HIGH | WAH_ONE | s->ones))) return -1;
    s->zeros = s->ones = 0;
    return 0;
}

/* Predicate kinds used by the encoders below. */
enum {
    PRED_ALLELE,  /* DefaultPred wah.hpp:431-435: allele(gt) == arg */
    PRED_MISSING, /* MissingPred gt_block.hpp:76-81 */
    PRED_EOV,     /* EndOfVectorPred gt_block.hpp:82-87 / RawPred :88-92 with raw = vector_end */
    PRED_PHASE    /* NonDefaultPhasingPred gt_block.hpp:93-100 (needs index) */
};
static inline int pred_eval(int kind, size_t idx, int32_t v, int32_t arg) {
    switch (kind) {
        case PRED_ALLELE: return gt_allele(v) == arg;
        case PRED_MISSING: return gt_is_missing(v) || v == XO_INT32_MISSING;
        case PRED_EOV: return v == XO_INT32_VECTOR_END;
        default: return (idx & 1) && (gt_is_phased(v) != arg);
    }
}

/* wah_encode2_with_size, both forms (wah.hpp:441-501 unpermuted with index,
 * wah.hpp:506-578 permuted through a).  perm == NULL means identity. */
static int wah_encode_pred(buf_t* out, const int32_t* gt, const uint32_t* perm, size_t size, int kind,
                           int32_t arg) {
    wah_state s = {0, 0};
    size_t b = 0;
    while (b < size) {
        uint16_t word = 0;
        for (unsigned j = 0; j < WAH_BITS; ++j, ++b) {
            if (b < size) {
                size_t src = perm ? perm[b] : b;
                if (pred_eval(kind, b, gt[src], arg)) word |= (uint16_t)(1u << j);
            }
        }
        if (wah_push_group(out, &s, word)) return -1;
    }
    return wah_flush(out, &s);
}

/* wah_encode2(std::vector<bool>&), wah.hpp:238-342 */
static int wah_encode_bits_buf(buf_t* out, const uint8_t* bits, size_t n) {
    wah_state s = {0, 0};
    size_t b = 0;
    while (b < n) {
        uint16_t word = 0;
        for (unsigned j = 0; j < WAH_BITS; ++j, ++b)
            if (b < n && bits[b]) word |= (uint16_t)(1u << j);
        if (wah_push_group(out, &s, word)) return -1;
    }
    return wah_flush(out, &s);
}

size_t xo_wah_encode_bits(const uint8_t* bits01, size_t n, uint16_t* out) {
    buf_t b = {0};
    if (wah_encode_bits_buf(&b, bits01, n)) {
        buf_free(&b);
        return 0;
    }
    size_t words = b.n / 2;
    memcpy(out, b.p, b.n);
    buf_free(&b);
    return words;
}

/* wah2_extract_template, wah.hpp:177-223 */
size_t xo_wah_extract(const uint16_t* wah, size_t n, uint8_t* bits01, size_t* ones) {
    size_t pos = 0, w = 0, cnt = 0;
    while (pos < n) {
        uint16_t word = wah[w++];
        if (word & WAH_HIGH) {
            size_t len = (size_t)(word & WAH_MAXC) * WAH_BITS;
            uint8_t v = (word & WAH_ONE) ? 1 : 0;
            /* the reference writes the whole run into a vector sized n+15; a run can only
             * overshoot by < 15 bits for streams this encoder produces, clamp for safety */
            size_t stop = pos + len;
            for (size_t i = pos; i < stop && i < n + WAH_BITS; ++i) bits01[i] = v;
            if (v) cnt += len;
            pos = stop;
        } else {
            for (unsigned j = 0; j < WAH_BITS; ++j) {
                bits01[pos + j] = (uint8_t)((word >> j) & 1);
                cnt += (word >> j) & 1;
            }
            pos += WAH_BITS;
        }
    }
    if (ones) *ones = cnt;
    return w;
}

/* wah2_advance_pointer, wah.hpp:158-174 */
static size_t wah_skip(const uint16_t* wah, size_t n) {
    size_t pos = 0, w = 0;
    while (pos < n) {
        uint16_t word = wah[w++];
        pos += (word & WAH_HIGH) ? (size_t)(word & WAH_MAXC) * WAH_BITS : WAH_BITS;
    }
    return w;
}

/* ------------------------------------------------------------------ */
/* PBWT (internal_gt_record.hpp:32-59; gt_block.hpp:124-136)           */
/* ------------------------------------------------------------------ */
void xo_pbwt_sort(uint32_t* a, uint32_t* b, size_t n_a, const int32_t* gt, int32_t alt, uint32_t ratio) {
    size_t u = 0, v = 0;
    for (size_t i = 0; i < n_a; ++i) {
        uint32_t h = a[i];
        if (gt_allele(gt[h / ratio]) != alt)
            a[u++] = h;
        else
            b[v++] = h;
    }
    memcpy(a + u, b, v * sizeof(uint32_t));
}

/* haploid_rearrangement_from_diploid, interfaces.hpp:318-333 */
static void haploid_rearrangement(const uint32_t* a, size_t n_a, uint32_t* a1) {
    size_t k = 0;
    for (size_t i = 0; i < n_a; ++i)
        if ((a[i] & 1) == 0) a1[k++] = a[i] / 2;
}

/* ------------------------------------------------------------------ */
/* GT block dictionary                                                 */
/* ------------------------------------------------------------------ */
enum {
    KEY_BCF_LINES = 0x0,
    KEY_BINARY_LINES = 0x1,
    KEY_MAX_LINE_PLOIDY = 0x2,
    KEY_DEFAULT_PHASING = 0x3,
    KEY_WEIRDNESS_STRATEGY = 0x4,
    KEY_LINE_SORT = 0x10,
    KEY_LINE_SELECT = 0x11,
    KEY_LINE_HAPLOID = 0x12,
    KEY_LINE_MISSING = 0x16,
    KEY_LINE_NON_UNIFORM_PHASING = 0x17,
    KEY_LINE_END_OF_VECTORS = 0x18,
    KEY_MATRIX_WAH = 0x20,
    KEY_MATRIX_SPARSE = 0x21,
    KEY_MATRIX_MISSING = 0x26,
    KEY_MATRIX_NON_UNIFORM_PHASING = 0x27,
    KEY_MATRIX_END_OF_VECTORS = 0x28,
    KEY_MATRIX_MISSING_SPARSE = 0x36,
    KEY_MATRIX_END_OF_VECTORS_SPARSE = 0x38,
};
#define VAL_UNDEFINED 0xFFFFFFFFu
#define WS_PBWT_WAH 0u
#define WS_WAH 1u
#define WS_SPARSE 2u

/* Order in which libstdc++'s std::unordered_map<uint32_t,uint32_t> iterates after the
 * insertion sequence of GtBlock::fill_dictionary (gt_block.hpp:464-510); recorded from the
 * reference built with GLIBCXX_3.4.30 — SURVEY.md §9.4.  Index = m | e<<1 | p<<2 | h<<3. */
static const uint8_t DICT_ORDER[16][19] = {
    /* m e p h */
    /*0000*/ {9, 0x21, 0x20, 0x11, 0x04, 0x10, 0x03, 0x02, 0x01, 0x00},
    /*1000*/ {12, 0x26, 0x16, 0x21, 0x20, 0x11, 0x04, 0x10, 0x03, 0x36, 0x02, 0x01, 0x00},
    /*0100*/ {12, 0x18, 0x21, 0x20, 0x38, 0x11, 0x04, 0x10, 0x03, 0x02, 0x28, 0x01, 0x00},
    /*1100*/ {15, 0x38, 0x28, 0x00, 0x01, 0x02, 0x36, 0x10, 0x11, 0x03, 0x20, 0x04, 0x21, 0x16, 0x26, 0x18},
    /*0010*/ {11, 0x17, 0x21, 0x20, 0x11, 0x04, 0x10, 0x03, 0x02, 0x01, 0x27, 0x00},
    /*1010*/ {14, 0x27, 0x00, 0x01, 0x02, 0x36, 0x10, 0x11, 0x03, 0x20, 0x04, 0x21, 0x16, 0x26, 0x17},
    /*0110*/ {14, 0x27, 0x00, 0x01, 0x28, 0x02, 0x10, 0x11, 0x38, 0x03, 0x20, 0x04, 0x21, 0x18, 0x17},
    /*1110*/ {17, 0x27, 0x17, 0x38, 0x28, 0x00, 0x01, 0x02, 0x36, 0x10, 0x11, 0x03, 0x20, 0x04, 0x21, 0x16, 0x26, 0x18},
    /*0001*/ {10, 0x12, 0x21, 0x20, 0x11, 0x04, 0x10, 0x03, 0x02, 0x01, 0x00},
    /*1001*/ {13, 0x12, 0x26, 0x16, 0x21, 0x20, 0x11, 0x04, 0x10, 0x03, 0x36, 0x02, 0x01, 0x00},
    /*0101*/ {13, 0x12, 0x18, 0x21, 0x20, 0x38, 0x11, 0x04, 0x10, 0x03, 0x02, 0x28, 0x01, 0x00},
    /*1101*/ {16, 0x12, 0x38, 0x28, 0x00, 0x01, 0x02, 0x36, 0x10, 0x11, 0x03, 0x20, 0x04, 0x21, 0x16, 0x26, 0x18},
    /*0011*/ {12, 0x12, 0x17, 0x21, 0x20, 0x11, 0x04, 0x10, 0x03, 0x02, 0x01, 0x27, 0x00},
    /*1011*/ {15, 0x12, 0x27, 0x00, 0x01, 0x02, 0x36, 0x10, 0x11, 0x03, 0x20, 0x04, 0x21, 0x16, 0x26, 0x17},
    /*0111*/ {15, 0x12, 0x27, 0x00, 0x01, 0x28, 0x02, 0x10, 0x11, 0x38, 0x03, 0x20, 0x04, 0x21, 0x18, 0x17},
    /*1111*/ {18, 0x12, 0x27, 0x17, 0x38, 0x28, 0x00, 0x01, 0x02, 0x36, 0x10, 0x11, 0x03, 0x20, 0x04, 0x21, 0x16, 0x26, 0x18},
};

/* ------------------------------------------------------------------ */
/* Writer                                                              */
/* ------------------------------------------------------------------ */
struct xo_writer {
    uint32_t n_samples, block_len, mac_thr;
    int32_t default_phased;
    uint32_t strat;
    int aet_block;  /* A_T bytes of the GT block encoder: n_samples <= 65535 (xsi_factory.hpp:424-427) */
    int aet_header; /* header.aet_bytes: n_samples*2 <= 65535 (gt_compressor_new.hpp:177-187) */
    char** names;

    buf_t file;
    uint64_t* indices;
    size_t n_indices, cap_indices;
    uint64_t entry_counter, variant_counter;

    /* current block (GtBlock state, gt_block.hpp:410-459, 681-686) */
    uint32_t bcf_lines, binary_lines, max_vector_length;
    int missing_found, eov_found, phase_found, haploid_found;
    uint32_t *a, *b, *a1;
    uint32_t* a_weird;  /* a_weirdness, gt_block.hpp:171,180: sorted by WS_PBWT_WAH only */
    buf_t is_wah;       /* one byte per binary line */
    buf_t has_missing;  /* one byte per BCF line */
    buf_t has_eov;      /* one byte per BCF line */
    buf_t has_phase;    /* one byte per BCF line */
    buf_t haploid;      /* one byte per BCF line (quirk §9.6.2) */
    buf_t n_alts;       /* u32 per BCF line */
    buf_t wah, sparse, miss_sparse, miss_wah, eov_sparse, eov_wah, phase_wah;
    uint64_t* allele_counts;
    size_t cap_allele_counts;
};

static void block_reset(xo_writer* w) {
    w->bcf_lines = w->binary_lines = 0;
    w->max_vector_length = 1; /* gt_block.hpp:168 */
    w->missing_found = w->eov_found = w->phase_found = w->haploid_found = 0;
    for (uint32_t i = 0; i < 2 * w->n_samples; ++i) w->a[i] = i; /* iota, gt_block.hpp:179 */
    for (uint32_t i = 0; i < 2 * w->n_samples; ++i) w->a_weird[i] = i; /* :180 */
    w->is_wah.n = w->has_missing.n = w->has_eov.n = w->has_phase.n = w->haploid.n = w->n_alts.n = 0;
    w->wah.n = w->sparse.n = w->miss_sparse.n = w->miss_wah.n = w->eov_sparse.n = w->eov_wah.n = 0;
    w->phase_wah.n = 0;
}

xo_writer* xo_writer_new(uint32_t n_samples, uint32_t block_len, uint32_t mac_thr, int32_t default_phased,
                         int wah_encode_missing, const char* const* sample_names) {
    /* A_T mismatch window: the reference's GtBlock<uint16_t> keeps
     * `std::vector<uint16_t> a(2*n_samples)` (gt_block.hpp:171,179), which wraps modulo 65536 for
     * 32768 <= n_samples <= 65535, under a header that says A_T = 4 bytes: it cannot decode its own
     * output there.  This restatement keeps `a` in uint32_t, so it would NOT be bug-compatible in the
     * window; it refuses it instead (as the product does, XSI_ERR_UNSUPPORTED). */
    if (n_samples >= 32768u && n_samples <= 65535u) return NULL;
    xo_writer* w = (xo_writer*)calloc(1, sizeof(*w));
    if (!w) return NULL;
    w->n_samples = n_samples;
    w->block_len = block_len;
    w->mac_thr = mac_thr;
    w->default_phased = default_phased;
    /* gt_block.hpp:174-176, 417: WS_SPARSE, or WS_WAH with --wah-encode-missing.  wah_encode_missing == 2 (tests only)
     * selects WS_PBWT_WAH, the version-4 default that the current CLI cannot choose any more but GtBlock still
     * implements (gt_block.hpp:340-395) and the reader still decodes (accessor_internals_new.hpp:300-340, 503-533) */
    w->strat = wah_encode_missing == 2 ? WS_PBWT_WAH : (wah_encode_missing ? WS_WAH : WS_SPARSE);
    w->aet_block = (n_samples <= 65535u) ? 2 : 4;
    w->aet_header = ((uint64_t)n_samples * 2 <= 65535u) ? 2 : 4;
    size_t na = (size_t)2 * n_samples;
    w->a = (uint32_t*)malloc((na ? na : 1) * sizeof(uint32_t));
    w->b = (uint32_t*)malloc((na ? na : 1) * sizeof(uint32_t));
    w->a1 = (uint32_t*)malloc((na ? na : 1) * sizeof(uint32_t));
    w->a_weird = (uint32_t*)malloc((na ? na : 1) * sizeof(uint32_t));
    w->names = (char**)calloc(n_samples ? n_samples : 1, sizeof(char*));
    if (!w->a || !w->b || !w->a1 || !w->a_weird || !w->names) {
        xo_writer_free(w);
        return NULL;
    }
    for (uint32_t i = 0; i < n_samples; ++i) {
        const char* s = sample_names ? sample_names[i] : "";
        w->names[i] = (char*)malloc(strlen(s) + 1);
        if (!w->names[i]) {
            xo_writer_free(w);
            return NULL;
        }
        strcpy(w->names[i], s);
    }
    /* provisional header: 256 bytes, rewritten at finalize (xsi_factory.hpp:468-511) */
    if (buf_zero(&w->file, 256)) {
        xo_writer_free(w);
        return NULL;
    }
    block_reset(w);
    return w;
}

void xo_writer_free(xo_writer* w) {
    if (!w) return;
    if (w->names)
        for (uint32_t i = 0; i < w->n_samples; ++i) free(w->names[i]);
    free(w->names);
    free(w->a);
    free(w->a_weird);
    free(w->b);
    free(w->a1);
    free(w->indices);
    free(w->allele_counts);
    buf_free(&w->file);
    buf_free(&w->is_wah);
    buf_free(&w->has_missing);
    buf_free(&w->has_eov);
    buf_free(&w->has_phase);
    buf_free(&w->haploid);
    buf_free(&w->n_alts);
    buf_free(&w->wah);
    buf_free(&w->sparse);
    buf_free(&w->miss_sparse);
    buf_free(&w->miss_wah);
    buf_free(&w->eov_sparse);
    buf_free(&w->eov_wah);
    buf_free(&w->phase_wah);
    free(w);
}
void xo_free(void* p) { free(p); }

/* Sparse<T,Pred> ctor + write_to_stream (block.hpp:54-99): count (MSB = negated) then indices. */
static int sparse_emit(buf_t* out, const int32_t* gt, int32_t ngt, int kind, int32_t arg, int negated_flag,
                       int aet) {
    size_t count_pos = out->n;
    if (buf_at(out, 0, aet)) return -1;
    uint32_t count = 0;
    for (int32_t i = 0; i < ngt; ++i) {
        if (pred_eval(kind, (size_t)i, gt[i], arg)) {
            if (buf_at(out, (uint32_t)i, aet)) return -1; /* truncated to A_T like push_back into vector<T> */
            count++;
        }
    }
    if (aet == 2) {
        uint16_t c = (uint16_t)count;
        if (negated_flag) c |= 0x8000u;
        memcpy(out->p + count_pos, &c, 2);
    } else {
        uint32_t c = count;
        if (negated_flag) c |= 0x80000000u;
        memcpy(out->p + count_pos, &c, 4);
    }
    return 0;
}

/* reindex_binary_vector_from_bcf_to_binary_lines, gt_block.hpp:650-666 */
static uint8_t* reindex_flags(const xo_writer* w, const buf_t* per_bcf) {
    uint8_t* r = (uint8_t*)calloc(w->binary_lines ? w->binary_lines : 1, 1);
    if (!r) return NULL;
    size_t off = 0;
    const uint32_t* n_alts = (const uint32_t*)w->n_alts.p;
    for (uint32_t i = 0; i < w->bcf_lines; ++i) {
        if (off < w->binary_lines) r[off] = per_bcf->p[i];
        off++;
        for (uint32_t k = 1; k < n_alts[i]; ++k) off++;
    }
    return r;
}

/* GtBlock::write_to_stream + IBinaryBlock::write_to_file (gt_block.hpp:185-204, 512-647;
 * interfaces.hpp:176-268), zstd off. */
static int block_flush(xo_writer* w) {
    buf_t* f = &w->file;
    if (w->n_indices == w->cap_indices) {
        size_t nc = w->cap_indices ? w->cap_indices * 2 : 16;
        uint64_t* ni = (uint64_t*)realloc(w->indices, nc * sizeof(uint64_t));
        if (!ni) return -1;
        w->indices = ni;
        w->cap_indices = nc;
    }
    w->indices[w->n_indices++] = (uint64_t)f->n; /* xsi_factory.hpp:533, 554 */

    /* outer dictionary: {KEY_GT_ENTRY=256 -> 16} */
    if (buf_u32(f, 0xFFFFFFFFu) || buf_u32(f, 1) || buf_u32(f, 256) || buf_u32(f, 16)) return -1;

    size_t gt_start = f->n;
    int idx = (w->missing_found ? 1 : 0) | (w->eov_found ? 2 : 0) | (w->phase_found ? 4 : 0) |
              (w->haploid_found ? 8 : 0);
    const uint8_t* order = DICT_ORDER[idx];
    uint32_t n_keys = order[0];
    if (buf_u32(f, 0xFFFFFFFFu) || buf_u32(f, n_keys)) return -1;
    size_t dict_pos = f->n;
    if (buf_zero(f, (size_t)n_keys * 8)) return -1;

    uint32_t vals[0x40];
    for (int i = 0; i < 0x40; ++i) vals[i] = VAL_UNDEFINED;
    vals[KEY_BCF_LINES] = w->bcf_lines;
    vals[KEY_BINARY_LINES] = w->binary_lines;
    vals[KEY_MAX_LINE_PLOIDY] = w->max_vector_length;
    vals[KEY_DEFAULT_PHASING] = (uint32_t)w->default_phased;
    vals[KEY_WEIRDNESS_STRATEGY] = w->strat;

    vals[KEY_LINE_SORT] = (uint32_t)(f->n - gt_start);
    if (wah_encode_bits_buf(f, w->is_wah.p, w->binary_lines)) return -1;
    vals[KEY_LINE_SELECT] = vals[KEY_LINE_SORT];
    vals[KEY_MATRIX_WAH] = (uint32_t)(f->n - gt_start);
    if (buf_put(f, w->wah.p, w->wah.n)) return -1;
    vals[KEY_MATRIX_SPARSE] = (uint32_t)(f->n - gt_start);
    if (buf_put(f, w->sparse.p, w->sparse.n)) return -1;

    if (w->missing_found) {
        uint8_t* v = reindex_flags(w, &w->has_missing);
        if (!v) return -1;
        vals[KEY_LINE_MISSING] = (uint32_t)(f->n - gt_start);
        int rc = wah_encode_bits_buf(f, v, w->binary_lines);
        free(v);
        if (rc) return -1;
        if (w->strat != WS_SPARSE) {
            vals[KEY_MATRIX_MISSING] = (uint32_t)(f->n - gt_start);
            if (buf_put(f, w->miss_wah.p, w->miss_wah.n)) return -1;
        } else {
            vals[KEY_MATRIX_MISSING_SPARSE] = (uint32_t)(f->n - gt_start);
            if (buf_put(f, w->miss_sparse.p, w->miss_sparse.n)) return -1;
        }
    }
    if (w->eov_found) {
        uint8_t* v = reindex_flags(w, &w->has_eov);
        if (!v) return -1;
        vals[KEY_LINE_END_OF_VECTORS] = (uint32_t)(f->n - gt_start);
        int rc = wah_encode_bits_buf(f, v, w->binary_lines);
        free(v);
        if (rc) return -1;
        if (w->strat != WS_SPARSE) {
            vals[KEY_MATRIX_END_OF_VECTORS] = (uint32_t)(f->n - gt_start);
            if (buf_put(f, w->eov_wah.p, w->eov_wah.n)) return -1;
        } else {
            vals[KEY_MATRIX_END_OF_VECTORS_SPARSE] = (uint32_t)(f->n - gt_start);
            if (buf_put(f, w->eov_sparse.p, w->eov_sparse.n)) return -1;
        }
    }
    if (w->phase_found) {
        uint8_t* v = reindex_flags(w, &w->has_phase);
        if (!v) return -1;
        vals[KEY_LINE_NON_UNIFORM_PHASING] = (uint32_t)(f->n - gt_start);
        int rc = wah_encode_bits_buf(f, v, w->binary_lines);
        free(v);
        if (rc) return -1;
        vals[KEY_MATRIX_NON_UNIFORM_PHASING] = (uint32_t)(f->n - gt_start);
        if (buf_put(f, w->phase_wah.p, w->phase_wah.n)) return -1;
    }
    if (w->haploid_found) {
        vals[KEY_LINE_HAPLOID] = (uint32_t)(f->n - gt_start);
        /* one bit per BCF line (gt_block.hpp:219-224, 639-642) */
        if (wah_encode_bits_buf(f, w->haploid.p, w->bcf_lines)) return -1;
    }

    for (uint32_t i = 0; i < n_keys; ++i) {
        uint32_t key = order[1 + i];
        memcpy(f->p + dict_pos + (size_t)i * 8, &key, 4);
        memcpy(f->p + dict_pos + (size_t)i * 8 + 4, &vals[key], 4);
    }
    /* pad block to 4 bytes (interfaces.hpp:254-263) */
    while (f->n % 4)
        if (buf_zero(f, 1)) return -1;
    block_reset(w);
    return 0;
}

/* XsiFactoryExt::append -> GtBlock::encode_line (xsi_factory.hpp:513-539; gt_block.hpp:207-406) */
int xo_writer_append(xo_writer* w, const int32_t* gt, int32_t ngt, int32_t n_allele) {
    if (!w || !gt || n_allele < 2 || w->n_samples == 0) return -1;
    if (ngt <= 0 || (uint32_t)ngt % w->n_samples) return -1;
    const uint32_t ploidy = (uint32_t)ngt / w->n_samples;
    if (ploidy != 1 && ploidy != 2) return -2; /* "PLOIDY ERROR", gt_block.hpp:313-316 */

    /* check_flush_block, xsi_factory.hpp:527-539 */
    if (w->entry_counter && (w->entry_counter % w->block_len) == 0)
        if (block_flush(w)) return -1;

    /* ---- scan_genotypes, gt_block.hpp:207-269 ---- */
    if (ploidy > w->max_vector_length) w->max_vector_length = ploidy;
    uint8_t hap_flag = (ploidy == 1);
    if (hap_flag) w->haploid_found = 1;
    if (buf_put(&w->haploid, &hap_flag, 1)) return -1;
    if ((size_t)n_allele > w->cap_allele_counts) {
        uint64_t* na = (uint64_t*)realloc(w->allele_counts, (size_t)n_allele * sizeof(uint64_t));
        if (!na) return -1;
        w->allele_counts = na;
        w->cap_allele_counts = (size_t)n_allele;
    }
    memset(w->allele_counts, 0, (size_t)n_allele * sizeof(uint64_t));
    uint8_t l_missing = 0, l_eov = 0, l_phase = 0;
    for (uint32_t i = 0; i < w->n_samples; ++i) {
        for (uint32_t j = 0; j < ploidy; ++j) {
            int32_t v = gt[(size_t)i * ploidy + j];
            if (j && gt_is_phased(v) != w->default_phased) l_phase = 1;
            if (gt_is_missing(v) || v == XO_INT32_MISSING) {
                l_missing = 1;
            } else if (v == XO_INT32_VECTOR_END) {
                l_eov = 1;
            } else {
                int32_t al = gt_allele(v);
                if (al < 0 || al >= n_allele) return -3; /* "Unknown allele error !" */
                w->allele_counts[al]++;
            }
        }
    }
    if (l_missing) w->missing_found = 1;
    if (l_eov) w->eov_found = 1;
    if (l_phase) w->phase_found = 1;
    uint32_t n_alt = (uint32_t)n_allele - 1;
    if (buf_put(&w->has_missing, &l_missing, 1) || buf_put(&w->has_eov, &l_eov, 1) ||
        buf_put(&w->has_phase, &l_phase, 1) || buf_put(&w->n_alts, &n_alt, 4))
        return -1;

    /* ---- per ALT allele, gt_block.hpp:292-328 ---- */
    const size_t n_a = (size_t)2 * w->n_samples;
    for (int32_t alt = 1; alt < n_allele; ++alt) {
        uint64_t cnt = w->allele_counts[alt];
        uint64_t minor = cnt < (uint64_t)ngt - cnt ? cnt : (uint64_t)ngt - cnt;
        uint8_t flag;
        if (minor > w->mac_thr) {
            flag = 1;
            if (ploidy == 1) {
                haploid_rearrangement(w->a, n_a, w->a1);
                if (wah_encode_pred(&w->wah, gt, w->a1, (size_t)ngt, PRED_ALLELE, alt)) return -1;
                xo_pbwt_sort(w->a, w->b, n_a, gt, alt, 2); /* pbwt_sort1 */
            } else {
                if (wah_encode_pred(&w->wah, gt, w->a, (size_t)ngt, PRED_ALLELE, alt)) return -1;
                xo_pbwt_sort(w->a, w->b, n_a, gt, alt, 1);
            }
        } else {
            flag = 0;
            int32_t sparse_allele = (cnt == minor) ? alt : 0;
            if (sparse_emit(&w->sparse, gt, ngt, PRED_ALLELE, sparse_allele, sparse_allele == 0, w->aet_block))
                return -1;
        }
        if (buf_put(&w->is_wah, &flag, 1)) return -1;
        w->binary_lines++;
    }

    /* ---- side channels, gt_block.hpp:330-401 ---- */
    if (l_missing && sparse_emit(&w->miss_sparse, gt, ngt, PRED_MISSING, 0, 0, w->aet_block)) return -1;
    if (l_eov && sparse_emit(&w->eov_sparse, gt, ngt, PRED_EOV, 0, 0, w->aet_block)) return -1;
    if (w->strat == WS_WAH) {
        /* a_weirdness stays identity (only WS_PBWT_WAH sorts it), so the permuted encoder
         * degenerates to natural order, for haploid lines too (a1 of iota = iota) */
        if (l_missing && wah_encode_pred(&w->miss_wah, gt, NULL, (size_t)ngt, PRED_MISSING, 0)) return -1;
        if (l_eov && wah_encode_pred(&w->eov_wah, gt, NULL, (size_t)ngt, PRED_EOV, 0)) return -1;
    }
    if (w->strat == WS_PBWT_WAH && (l_missing || l_eov)) {
        /* gt_block.hpp:340-395: the lines go through a_weirdness, which is then partitioned by "missing or end of
         * vector" (WeirdnessPred).  A fully haploid line would be encoded through a1 and leave a_weirdness unsorted
         * (:350, :380-384) while the reader indexes a_weird directly (accessor_internals_new.hpp:313): refused here */
        if ((uint32_t)ngt != 2 * w->n_samples) return -1;
        if (l_missing && wah_encode_pred(&w->miss_wah, gt, w->a_weird, (size_t)ngt, PRED_MISSING, 0)) return -1;
        if (l_eov && wah_encode_pred(&w->eov_wah, gt, w->a_weird, (size_t)ngt, PRED_EOV, 0)) return -1;
        size_t u = 0, v = 0;
        for (size_t i = 0; i < (size_t)ngt; ++i) {
            const int32_t g = gt[w->a_weird[i]];
            const int weird = (g == XO_INT32_VECTOR_END) || ((g >> 1) == 0);
            if (!weird) w->a_weird[u++] = w->a_weird[i];
            else w->b[v++] = w->a_weird[i];
        }
        memcpy(w->a_weird + u, w->b, v * sizeof(uint32_t));
    }
    if (l_phase && wah_encode_pred(&w->phase_wah, gt, NULL, (size_t)ngt, PRED_PHASE, w->default_phased)) return -1;

    w->bcf_lines++;
    w->variant_counter += n_alt;
    w->entry_counter++;
    return 0;
}

int xo_writer_append_rows(xo_writer* w, const int32_t* gt, size_t n_rows, size_t stride, int32_t ngt, int32_t n_allele) {
    for (size_t r = 0; r < n_rows; ++r) {
        int rc = xo_writer_append(w, gt + r * stride, ngt, n_allele);
        if (rc) return rc;
    }
    return 0;
}

static void put_le(uint8_t* p, uint64_t v, int bytes) {
    for (int i = 0; i < bytes; ++i) p[i] = (uint8_t)(v >> (8 * i));
}

/* XsiFactoryExt::finalize_file, xsi_factory.hpp:543-606; header_t compression.hpp:40-104 */
int xo_writer_finalize(xo_writer* w, uint32_t max_ploidy, uint8_t** out, size_t* out_len) {
    if (!w || !out || !out_len) return -1;
    buf_t* f = &w->file;
    if (w->bcf_lines)
        if (block_flush(w)) return -1;
    while (f->n % 8)
        if (buf_zero(f, 1)) return -1;
    uint64_t indices_offset = f->n;
    if (buf_put(f, w->indices, w->n_indices * sizeof(uint64_t))) return -1;
    uint64_t samples_offset = f->n;
    for (uint32_t i = 0; i < w->n_samples; ++i)
        if (buf_put(f, w->names[i], strlen(w->names[i]) + 1)) return -1;

    uint8_t* h = f->p;
    memset(h, 0, 256);
    put_le(h + 0, 0xaabbccddu, 4);
    put_le(h + 4, 0xfeed1767u, 4);
    put_le(h + 8, 5, 4);
    h[12] = (uint8_t)max_ploidy;
    h[13] = 4; /* ind_bytes = sizeof(uint32_t), stale */
    h[14] = (uint8_t)w->aet_header;
    h[15] = 2;
    h[16] = (uint8_t)((w->default_phased & 1) << 2); /* bool default_phased : 1 at bit 2 */
    h[17] = 1;                                       /* iota_ppa=1, no_sort=0, zstd=0 */
    put_le(h + 32, (uint64_t)w->n_samples * max_ploidy, 8);
    put_le(h + 40, w->variant_counter, 8);
    put_le(h + 48, 0, 4);
    put_le(h + 52, 1, 4);
    put_le(h + 56, w->block_len, 4);
    put_le(h + 60, (uint32_t)((w->entry_counter + (uint32_t)w->block_len - 1) / (uint32_t)w->block_len), 4);
    put_le(h + 64, 256, 8);
    put_le(h + 72, indices_offset, 8);
    put_le(h + 80, samples_offset, 8);
    put_le(h + 88, 0xFFFFFFFFu, 4);
    put_le(h + 92, 0xFFFFFFFFu, 4);
    put_le(h + 96, w->mac_thr, 4);
    put_le(h + 100, w->entry_counter, 8);
    put_le(h + 108, 0, 4);
    put_le(h + 112, w->n_samples, 8);
    put_le(h + 252, 0xfeed1767u, 4);

    *out = f->p;
    *out_len = f->n;
    f->p = NULL;
    f->n = f->cap = 0;
    return 0;
}

/* ------------------------------------------------------------------ */
/* Reader                                                              */
/* ------------------------------------------------------------------ */
typedef struct {
    int present;
    uint32_t val;
} dict_ent;

struct xo_reader {
    const uint8_t* file;
    size_t len;
    uint32_t version;
    int aet;
    uint64_t hap_samples, num_samples, indices_offset, n_blocks;

    /* DecompressPointerGTBlock state */
    int64_t cur_block;
    const uint8_t* gt_block;
    size_t N_SAMPLES, N_HAPS;
    uint32_t bcf_lines, binary_lines;
    int32_t default_phasing;
    uint32_t strat;
    uint8_t *is_wah, *l_missing, *l_eov, *l_phase, *haploid; /* per binary line (+15 slack) */
    int has_missing_vec, has_eov_vec, has_phase_vec;
    int has_weirdness, has_phase;
    const uint8_t *wah_origin, *sparse_origin, *miss_wah_origin, *miss_sparse_origin, *eov_wah_origin,
        *eov_sparse_origin, *phase_origin;
    const uint8_t *wah_p, *sparse_p, *miss_wah_p, *miss_sparse_p, *eov_wah_p, *eov_sparse_p, *phase_p;
    size_t pos, weird_pos, phase_pos;
    uint32_t *a, *b, *a1;
    uint32_t* a_weird; /* accessor_internals_new.hpp:62,146 */
    uint8_t *y, *y2, *x;
    uint8_t* y3;
    size_t ones;
    int sparse_negated;
    uint32_t* sparse;
    size_t n_sparse;
    uint64_t* allele_counts;
    size_t cap_allele_counts;
};

static uint64_t get_le(const uint8_t* p, int bytes) {
    uint64_t v = 0;
    for (int i = 0; i < bytes; ++i) v |= (uint64_t)p[i] << (8 * i);
    return v;
}

xo_reader* xo_reader_open(const uint8_t* file, size_t len) {
    if (!file || len < 256) return NULL;
    if (get_le(file + 0, 4) != 0xaabbccddu) return NULL;
    if (get_le(file + 4, 4) != 0xfeed1767u || get_le(file + 252, 4) != 0xfeed1767u) return NULL;
    uint32_t version = (uint32_t)get_le(file + 8, 4);
    if (version != 4 && version != 5) return NULL; /* accessor_internals_new.hpp:782-785 */
    if (file[12] == 0) return NULL;                /* "Ploidy in header is set to 0" :814-817 */
    if (file[17] & 4) return NULL;                 /* zstd layer is not restated here */
    int aet = file[14];
    if (aet != 2 && aet != 4) return NULL; /* accessor.cpp:62-80 */
    xo_reader* r = (xo_reader*)calloc(1, sizeof(*r));
    if (!r) return NULL;
    r->file = file;
    r->len = len;
    r->version = version;
    r->aet = aet;
    r->hap_samples = get_le(file + 32, 8);
    r->num_samples = get_le(file + 112, 8);
    r->indices_offset = get_le(file + 72, 8);
    uint64_t samples_offset = get_le(file + 80, 8);
    r->n_blocks = (samples_offset - r->indices_offset) / (version >= 5 ? 8 : 4);
    r->cur_block = -1;
    /* accessor_internals_new.hpp:53 */
    r->N_SAMPLES = (size_t)r->num_samples;
    r->N_HAPS = r->N_SAMPLES ? r->N_SAMPLES * 2 : (size_t)r->hap_samples;
    size_t n = r->N_HAPS + 16;
    r->a = (uint32_t*)malloc(n * sizeof(uint32_t));
    r->b = (uint32_t*)malloc(n * sizeof(uint32_t));
    r->a1 = (uint32_t*)malloc(n * sizeof(uint32_t));
    r->y = (uint8_t*)calloc(n, 1);
    r->y2 = (uint8_t*)calloc(n, 1);
    r->y3 = (uint8_t*)calloc(n, 1);
    r->a_weird = (uint32_t*)malloc(n * sizeof(uint32_t));
    r->x = (uint8_t*)calloc(n, 1);
    r->sparse = (uint32_t*)malloc(n * sizeof(uint32_t));
    if (!r->a || !r->b || !r->a1 || !r->y || !r->y2 || !r->y3 || !r->a_weird || !r->x || !r->sparse) {
        xo_reader_close(r);
        return NULL;
    }
    return r;
}

void xo_reader_close(xo_reader* r) {
    if (!r) return;
    free(r->a);
    free(r->b);
    free(r->a1);
    free(r->y);
    free(r->y2);
    free(r->y3);
    free(r->a_weird);
    free(r->x);
    free(r->sparse);
    free(r->is_wah);
    free(r->l_missing);
    free(r->l_eov);
    free(r->l_phase);
    free(r->haploid);
    free(r->allele_counts);
    free(r);
}
uint64_t xo_reader_hap_samples(const xo_reader* r) { return r->hap_samples; }
uint64_t xo_reader_num_samples(const xo_reader* r) { return r->num_samples; }
const uint64_t* xo_reader_allele_counts(const xo_reader* r) { return r->allele_counts; }

/* read_dictionary (interfaces.hpp:77-90) into a small direct-indexed table (keys < 0x40). */
static void read_dict(const uint8_t* p, dict_ent* d) {
    for (int i = 0; i < 0x40; ++i) d[i].present = 0;
    uint32_t n = (uint32_t)get_le(p + 4, 4);
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t k = (uint32_t)get_le(p + 8 + (size_t)i * 8, 4);
        uint32_t v = (uint32_t)get_le(p + 12 + (size_t)i * 8, 4);
        if (k < 0x40) {
            d[k].present = 1;
            d[k].val = v;
        }
    }
}

/* fill_bool_vector_from_1d_dict_key, accessor_internals_new.hpp:591-604 */
static int load_flag_vector(const xo_reader* r, const dict_ent* d, uint32_t key, uint8_t** dst, size_t n) {
    free(*dst);
    *dst = NULL;
    if (!d[key].present || d[key].val == VAL_UNDEFINED) return 0;
    *dst = (uint8_t*)calloc(n + 16, 1);
    if (!*dst) return 0;
    xo_wah_extract((const uint16_t*)(r->gt_block + d[key].val), n, *dst, NULL);
    return 1;
}
static const uint8_t* dict_ptr(const xo_reader* r, const dict_ent* d, uint32_t key) {
    if (!d[key].present || d[key].val == VAL_UNDEFINED) return NULL;
    return r->gt_block + d[key].val;
}

static void reader_reset(xo_reader* r) { /* reset(), accessor_internals_new.hpp:386-405 */
    for (size_t i = 0; i < r->N_HAPS; ++i) r->a[i] = (uint32_t)i;
    for (size_t i = 0; i < r->N_HAPS; ++i) r->a_weird[i] = (uint32_t)i; /* :394 */
    r->pos = r->weird_pos = r->phase_pos = 0;
    r->wah_p = r->wah_origin;
    r->sparse_p = r->sparse_origin;
    r->miss_wah_p = r->miss_wah_origin;
    r->miss_sparse_p = r->miss_sparse_origin;
    r->eov_wah_p = r->eov_wah_origin;
    r->eov_sparse_p = r->eov_sparse_origin;
    r->phase_p = r->phase_origin;
}

/* set_block_ptr + DecompressPointerGTBlock ctor (accessor_internals_new.hpp:830-893, 52-148) */
static int reader_set_block(xo_reader* r, uint64_t block_id) {
    if (block_id >= r->n_blocks) return -1;
    uint64_t off = (r->version >= 5) ? get_le(r->file + r->indices_offset + block_id * 8, 8)
                                     : get_le(r->file + r->indices_offset + block_id * 4, 4);
    if (off + 16 > r->len) return -1;
    const uint8_t* blk = r->file + off;
    uint32_t n_outer = (uint32_t)get_le(blk + 4, 4);
    const uint8_t* gt = NULL;
    for (uint32_t i = 0; i < n_outer; ++i)
        if (get_le(blk + 8 + (size_t)i * 8, 4) == 256) gt = blk + get_le(blk + 12 + (size_t)i * 8, 4);
    if (!gt) return -1;
    r->gt_block = gt;
    r->cur_block = (int64_t)block_id;
    dict_ent d[0x40];
    read_dict(gt, d);
    if (!d[KEY_BCF_LINES].present || !d[KEY_BINARY_LINES].present) return -1;
    r->bcf_lines = d[KEY_BCF_LINES].val;
    r->binary_lines = d[KEY_BINARY_LINES].val;
    uint32_t dp = d[KEY_DEFAULT_PHASING].present ? d[KEY_DEFAULT_PHASING].val : 0;
    r->default_phasing = (dp == 1) ? 1 : 0; /* :77-81, net effect */
    r->strat = d[KEY_WEIRDNESS_STRATEGY].present ? d[KEY_WEIRDNESS_STRATEGY].val : WS_PBWT_WAH;
    size_t nb = r->binary_lines;
    load_flag_vector(r, d, KEY_LINE_SELECT, &r->is_wah, nb);
    if (!r->is_wah) return -1;
    /* KEY_LINE_SORT == KEY_LINE_SELECT for every file this format's writer produces */
    r->has_missing_vec = load_flag_vector(r, d, KEY_LINE_MISSING, &r->l_missing, nb);
    r->has_eov_vec = load_flag_vector(r, d, KEY_LINE_END_OF_VECTORS, &r->l_eov, nb);
    r->has_weirdness = r->has_missing_vec || r->has_eov_vec;
    r->has_phase_vec = load_flag_vector(r, d, KEY_LINE_NON_UNIFORM_PHASING, &r->l_phase, nb);
    r->has_phase = r->has_phase_vec;
    if (!load_flag_vector(r, d, KEY_LINE_HAPLOID, &r->haploid, nb)) {
        r->haploid = (uint8_t*)calloc(nb + 16, 1);
        if (!r->haploid) return -1;
    }
    r->wah_origin = dict_ptr(r, d, KEY_MATRIX_WAH);
    r->sparse_origin = dict_ptr(r, d, KEY_MATRIX_SPARSE);
    r->miss_wah_origin = dict_ptr(r, d, KEY_MATRIX_MISSING);
    r->miss_sparse_origin = dict_ptr(r, d, KEY_MATRIX_MISSING_SPARSE);
    r->eov_wah_origin = dict_ptr(r, d, KEY_MATRIX_END_OF_VECTORS);
    r->eov_sparse_origin = dict_ptr(r, d, KEY_MATRIX_END_OF_VECTORS_SPARSE);
    r->phase_origin = dict_ptr(r, d, KEY_MATRIX_NON_UNIFORM_PHASING);
    reader_reset(r);
    return 0;
}

static inline size_t cur_n_haps(const xo_reader* r, size_t line) {
    return r->haploid[line] ? r->N_SAMPLES : r->N_HAPS;
}

/* sparse_extract / sparse_advance_pointer, accessor_internals_new.hpp:619-653 */
static const uint8_t* sparse_read(xo_reader* r, const uint8_t* p, int extract, uint32_t* dst, size_t* n_dst,
                                  int set_ones) {
    uint32_t num = (uint32_t)get_le(p, r->aet);
    uint32_t msb = (r->aet == 2) ? 0x8000u : 0x80000000u;
    p += r->aet;
    int neg = (num & msb) != 0;
    num &= ~msb;
    if (extract) {
        for (uint32_t i = 0; i < num; ++i) dst[i] = (uint32_t)get_le(p + (size_t)i * r->aet, r->aet);
        *n_dst = num;
    }
    p += (size_t)num * r->aet;
    if (set_ones) {
        r->sparse_negated = neg;
        size_t n = cur_n_haps(r, r->pos);
        r->ones = neg ? n - num : num;
    }
    return p;
}

/* bool_pbwt_sort, gt_block.hpp:124-136 */
static void bool_pbwt(uint32_t* a, uint32_t* b, const uint8_t* y, size_t n) {
    size_t u = 0, v = 0;
    for (size_t i = 0; i < n; ++i) {
        if (!y[i])
            a[u++] = a[i];
        else
            b[v++] = a[i];
    }
    memcpy(a + u, b, v * sizeof(uint32_t));
}

/* update_a_if_needed + private_pbwt_sort, accessor_internals_new.hpp:548-589 */
static void update_a(xo_reader* r) {
    if (!r->is_wah[r->pos]) return; /* sorting == wah */
    if (r->haploid[r->pos]) {
        haploid_rearrangement(r->a, r->N_HAPS, r->a1);
        for (size_t i = 0; i < r->N_SAMPLES; ++i) r->x[r->a1[i]] = r->y[i];
        size_t u = 0, v = 0;
        for (size_t i = 0; i < r->N_SAMPLES * 2; ++i) {
            if (!r->x[r->a[i] / 2])
                r->a[u++] = r->a[i];
            else
                r->b[v++] = r->a[i];
        }
        memcpy(r->a + u, r->b, v * sizeof(uint32_t));
    } else {
        bool_pbwt(r->a, r->b, r->y, r->N_HAPS);
    }
}

/* weirdness_advance, accessor_internals_new.hpp:478-537 (all three strategies; WS_PBWT_WAH, the version-4 default,
 * has no fixture of the reference's to pin it: restated from the source alone) */
static void weirdness_advance(xo_reader* r, size_t steps, size_t n) {
    for (size_t i = 0; i < steps; ++i) {
        int m = r->has_missing_vec && r->l_missing[r->weird_pos];
        int e = r->has_eov_vec && r->l_eov[r->weird_pos];
        if (r->strat == WS_SPARSE) {
            if (m) r->miss_sparse_p = sparse_read(r, r->miss_sparse_p, 0, NULL, NULL, 0);
            if (e) r->eov_sparse_p = sparse_read(r, r->eov_sparse_p, 0, NULL, NULL, 0);
        } else if (r->strat == WS_PBWT_WAH && (m || e)) {
            /* :492-533: the lines are extracted and a_weird is partitioned by "missing or end of vector"; on a fully
             * haploid line the sort is commented out in the reference (:508-510, :516-518, :525-527) */
            memset(r->y2, 0, r->N_HAPS);
            memset(r->y3, 0, r->N_HAPS);
            if (m) r->miss_wah_p += 2 * xo_wah_extract((const uint16_t*)r->miss_wah_p, n, r->y2, NULL);
            if (e) r->eov_wah_p += 2 * xo_wah_extract((const uint16_t*)r->eov_wah_p, n, r->y3, NULL);
            if (!r->haploid[r->weird_pos]) {
                size_t u = 0, v = 0;
                for (size_t k = 0; k < r->N_HAPS; ++k) {
                    if (!(r->y2[k] | r->y3[k])) r->a_weird[u++] = r->a_weird[k];
                    else r->b[v++] = r->a_weird[k];
                }
                memcpy(r->a_weird + u, r->b, v * sizeof(uint32_t));
            }
        } else {
            if (m) r->miss_wah_p += 2 * wah_skip((const uint16_t*)r->miss_wah_p, n);
            if (e) r->eov_wah_p += 2 * wah_skip((const uint16_t*)r->eov_wah_p, n);
        }
        r->weird_pos++;
    }
}
/* phase_advance, :539-546 */
static void phase_advance(xo_reader* r, size_t steps, size_t n) {
    for (size_t i = 0; i < steps; ++i) {
        if (r->has_phase_vec && r->l_phase[r->phase_pos])
            r->phase_p += 2 * wah_skip((const uint16_t*)r->phase_p, n);
        r->phase_pos++;
    }
}

/* seek, accessor_internals_new.hpp:154-196 */
static void reader_seek_line(xo_reader* r, size_t position) {
    if (r->pos == position) return;
    if (r->pos > position) reader_reset(r);
    while (r->pos < position) {
        size_t n = cur_n_haps(r, r->pos);
        if (r->is_wah[r->pos]) {
            r->wah_p += 2 * xo_wah_extract((const uint16_t*)r->wah_p, n, r->y, NULL);
        } else {
            size_t ns;
            r->sparse_p = sparse_read(r, r->sparse_p, 1, r->sparse, &ns, 1);
        }
        update_a(r);
        if (r->has_weirdness) weirdness_advance(r, 1, n);
        if (r->has_phase) phase_advance(r, 1, n);
        r->pos++;
    }
}

static int reader_seek(xo_reader* r, uint64_t bm) { /* :722-738 */
    uint64_t block_id = (bm & 0xFFFFFFFFu) >> 15;
    uint32_t offset = (uint32_t)(bm & 0x7FFF);
    if (r->cur_block < 0 || (uint64_t)r->cur_block != block_id)
        if (reader_set_block(r, block_id)) return -1;
    if (offset >= r->binary_lines) return -1;
    reader_seek_line(r, offset);
    return 0;
}

static int ensure_counts(xo_reader* r, uint32_t n_alleles) {
    if (n_alleles > r->cap_allele_counts) {
        uint64_t* na = (uint64_t*)realloc(r->allele_counts, (size_t)n_alleles * sizeof(uint64_t));
        if (!na) return -1;
        r->allele_counts = na;
        r->cap_allele_counts = n_alleles;
    }
    return 0;
}

/* fill_genotype_array_advance, accessor_internals_new.hpp:198-384 */
int64_t xo_reader_fill_genotype_array(xo_reader* r, int32_t* gt, size_t gt_size, uint32_t n_alleles, uint64_t bm) {
    if (!r || !gt || n_alleles < 2) return -1;
    if (reader_seek(r, bm)) return -1;
    if (ensure_counts(r, n_alleles)) return -1;
    if (r->pos + (n_alleles - 1) > r->binary_lines) return -1;
    const size_t N = cur_n_haps(r, r->pos);
    if (gt_size < N) return -1;
    const size_t start = r->pos;
    const int32_t DP = r->default_phasing;
    size_t total_alt = 0, n_missing = 0, n_eovs = 0;

    if (!r->is_wah[r->pos]) {
        size_t ns;
        r->sparse_p = sparse_read(r, r->sparse_p, 1, r->sparse, &ns, 1);
        int32_t default_gt = r->sparse_negated ? 1 : 0, sparse_gt = r->sparse_negated ? 0 : 1;
        for (size_t i = 0; i < N; ++i) gt[i] = gt_unphased(default_gt) | ((int32_t)(i & 1) & DP);
        for (size_t k = 0; k < ns; ++k) {
            size_t i = r->sparse[k];
            gt[i] = gt_unphased(sparse_gt) | ((int32_t)(i & 1) & DP);
        }
    } else {
        r->wah_p += 2 * xo_wah_extract((const uint16_t*)r->wah_p, N, r->y, &r->ones);
        if (r->haploid[r->pos]) {
            haploid_rearrangement(r->a, r->N_HAPS, r->a1);
            for (size_t i = 0; i < N; ++i) gt[r->a1[i]] = gt_unphased(r->y[i]);
        } else {
            for (size_t i = 0; i < N; ++i) gt[r->a[i]] = gt_unphased(r->y[i]) | ((int32_t)(r->a[i] & 1) & DP);
        }
    }
    r->allele_counts[1] = r->ones;
    total_alt = r->ones;
    update_a(r);
    r->pos++;

    for (uint32_t alt = 2; alt < n_alleles; ++alt) {
        if (!r->is_wah[r->pos]) {
            size_t ns;
            r->sparse_p = sparse_read(r, r->sparse_p, 1, r->sparse, &ns, 1);
            if (r->sparse_negated) {
                for (size_t i = 0; i < N; ++i)
                    if (gt_allele(gt[i]) == 0) gt[i] = gt_unphased((int32_t)alt) | ((int32_t)(i & 1) & DP);
                for (size_t k = 0; k < ns; ++k) {
                    size_t i = r->sparse[k];
                    if (gt_allele(gt[i]) == (int32_t)alt) gt[i] = gt_unphased(0) | ((int32_t)(i & 1) & DP);
                }
            } else {
                for (size_t k = 0; k < ns; ++k) {
                    size_t i = r->sparse[k];
                    gt[i] = gt_unphased((int32_t)alt) | ((int32_t)(i & 1) & DP);
                }
            }
        } else {
            r->wah_p += 2 * xo_wah_extract((const uint16_t*)r->wah_p, N, r->y, &r->ones);
            if (r->haploid[r->pos]) {
                haploid_rearrangement(r->a, r->N_HAPS, r->a1);
                for (size_t i = 0; i < N; ++i)
                    if (r->y[i]) gt[r->a1[i]] = gt_unphased(r->y[i]); /* sic: allele 1, :269 */
            } else {
                for (size_t i = 0; i < N; ++i)
                    if (r->y[i]) gt[r->a[i]] = gt_unphased((int32_t)alt) | ((int32_t)(r->a[i] & 1) & DP);
            }
        }
        r->allele_counts[alt] = r->ones;
        total_alt += r->ones;
        update_a(r);
        r->pos++;
    }

    if (r->has_weirdness) {
        if (r->has_missing_vec && r->l_missing[start]) {
            if (r->strat == WS_SPARSE) {
                size_t ns;
                (void)sparse_read(r, r->miss_sparse_p, 1, r->sparse, &ns, 0);
                n_missing = ns;
                for (size_t k = 0; k < ns; ++k) {
                    size_t i = r->sparse[k];
                    gt[i] = XO_GT_MISSING | ((int32_t)(i & 1) & DP);
                }
            } else { /* WS_WAH / WS_PBWT_WAH through a_weird (the identity for WS_WAH), :308-318 */
                (void)xo_wah_extract((const uint16_t*)r->miss_wah_p, N, r->y2, &n_missing);
                for (size_t i = 0; i < N; ++i)
                    if (r->y2[i]) {
                        const size_t idx = r->a_weird[i];
                        gt[idx] = XO_GT_MISSING | ((int32_t)(idx & 1) & DP);
                    }
            }
        }
        if (r->has_eov_vec && r->l_eov[start]) {
            if (r->strat == WS_SPARSE) {
                size_t ns;
                (void)sparse_read(r, r->eov_sparse_p, 1, r->sparse, &ns, 0);
                n_eovs = ns;
                for (size_t k = 0; k < ns; ++k) gt[r->sparse[k]] = XO_INT32_VECTOR_END;
            } else {
                (void)xo_wah_extract((const uint16_t*)r->eov_wah_p, N, r->y2, &n_eovs);
                for (size_t i = 0; i < N; ++i)
                    if (r->y2[i]) gt[r->a_weird[i]] = XO_INT32_VECTOR_END; /* :331-337 */
            }
        }
        weirdness_advance(r, n_alleles - 1, N);
    }
    if (r->has_phase) {
        if (r->has_phase_vec && r->l_phase[start]) {
            (void)xo_wah_extract((const uint16_t*)r->phase_p, N, r->y2, NULL);
            for (size_t i = 0; i < N; ++i)
                if (r->y2[i] && gt[i] != XO_INT32_VECTOR_END) gt[i] ^= (int32_t)(i & 1);
        }
        phase_advance(r, n_alleles - 1, N);
    }
    r->allele_counts[0] = N - (total_alt + n_missing + n_eovs);
    return (int64_t)N;
}

int64_t xo_reader_fill_rows(xo_reader* r, int32_t* gt, size_t stride, uint64_t first_line, size_t n_rows,
                            uint32_t block_len) {
    for (size_t i = 0; i < n_rows; ++i) {
        const uint64_t line = first_line + i;
        const uint64_t bm = ((line / block_len) << 15) | (line % block_len);
        int64_t n = xo_reader_fill_genotype_array(r, gt + i * stride, stride, 2, bm);
        if (n < 0) return n;
    }
    return (int64_t)n_rows;
}

/* fill_allele_counts_advance, accessor_internals_new.hpp:407-438 */
int xo_reader_fill_allele_counts(xo_reader* r, uint32_t n_alleles, uint64_t bm) {
    if (!r || n_alleles < 2) return -1;
    if (reader_seek(r, bm)) return -1;
    if (ensure_counts(r, n_alleles)) return -1;
    if (r->pos + (n_alleles - 1) > r->binary_lines) return -1;
    const size_t N = cur_n_haps(r, r->pos);
    size_t total_alt = 0;
    for (uint32_t alt = 1; alt < n_alleles; ++alt) {
        if (r->is_wah[r->pos]) {
            r->wah_p += 2 * xo_wah_extract((const uint16_t*)r->wah_p, N, r->y, &r->ones);
        } else {
            size_t ns;
            r->sparse_p = sparse_read(r, r->sparse_p, 1, r->sparse, &ns, 1);
        }
        update_a(r);
        r->pos++;
        r->allele_counts[alt] = r->ones;
        total_alt += r->ones;
    }
    r->allele_counts[0] = N - total_alt;
    /* the reference does not advance the weirdness / phase cursors here (:407-438) */
    return 0;
}
