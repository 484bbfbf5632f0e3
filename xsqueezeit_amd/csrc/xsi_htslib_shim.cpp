// xsi_htslib_shim.cpp — the htslib-facing layer above libxsi_hip.so's C ABI, compiled ONLY where htslib is
// installed (this image has none: no <htslib/vcf.h>, no libhts).  With htslib present it
// exports, under their reference names, the symbols an existing HTSLIB caller links against:
//
//   c_xcf_new / c_xcf_add_readers / c_xcf_update_readers / c_xcf_sample_name / c_xcf_nsamples /
//   __c__xcf__get__genotypes__void / c_xcf_delete            (include/c_api.h:38-93, c_api.cpp:37-85; the table of
//                                                             readers behind them: xsi_mixed_vcf.cpp:46-106)
//   xsi_compress_bcf(in, out)                                 the -c fill loop: BcfTraversal::traverse
//                                                             (bcf_traversal.cpp:3-16) feeding XsiFactoryInterface::
//                                                             append, plus the variant-only BCF with the BM field
//                                                             (xcf.cpp:641-714)
//
// and nothing else: every genotype goes through xsi_writer_* / xsi_accessor_* of include/xsi_hip.h.  Without
// htslib the file compiles to xsi_htslib_shim_available() == 0, so the library's symbol set says which it is.
// It is untested here (it cannot be compiled in this image); it is kept small for that reason.
#include "../../include/xsi_hip.h"

extern "C" int xsi_htslib_shim_available(void);

#if defined(__has_include)
#if __has_include(<htslib/vcf.h>) && __has_include(<htslib/synced_bcf_reader.h>)
#define XSI_HAVE_HTSLIB 1
#endif
#endif

#ifndef XSI_HAVE_HTSLIB

extern "C" int xsi_htslib_shim_available(void) { return 0; }

#else

#include <htslib/synced_bcf_reader.h>
#include <htslib/vcf.h>
#include <sys/stat.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

extern "C" int xsi_htslib_shim_available(void) { return 1; }

namespace {

// the .xsi that belongs to a variant-only BCF: its "##XSI=<basename>" header line next to the BCF, else the name
// with "_var.bcf" cut off (accessor.hpp:89-111)
bool xsi_path_of(const char* bcf_path, const bcf_hdr_t* hdr, std::string& out) {
    std::string p(bcf_path ? bcf_path : "");
    if (hdr) {
        bcf_hrec_t* h = bcf_hdr_get_hrec(hdr, BCF_HL_GEN, "XSI", nullptr, nullptr);
        if (h && h->value) {
            const size_t slash = p.find_last_of('/');
            out = (slash == std::string::npos ? std::string(".") : p.substr(0, slash)) + "/" + h->value;
            return true;
        }
    }
    const size_t pos = p.find("_var.bcf");
    if (pos == std::string::npos) return false;
    out = p.substr(0, pos);
    return true;
}
bool file_exists(const std::string& p) {
    struct stat st;
    return ::stat(p.c_str(), &st) == 0;
}

struct Entry {
    bool is_xsi = false;
    xsi_accessor* acc = nullptr;
    int32_t* bm = nullptr;  // bcf_get_format_int32 scratch (Accessor::values, accessor.hpp:121-122)
    int nbm = 0;
};
struct Xcf {
    xsi_hip_ctx* ctx = nullptr;
    std::vector<Entry> entries;
    ~Xcf() {
        for (auto& e : entries) {
            if (e.acc) xsi_accessor_close(e.acc);
            free(e.bm);
        }
        if (ctx) xsi_hip_ctx_destroy(ctx);
    }
};

}  // namespace

extern "C" {

typedef void* c_xcf;

c_xcf* c_xcf_new(void) {
    Xcf* x = new (std::nothrow) Xcf();
    if (x && xsi_hip_ctx_create(&x->ctx, 0, nullptr) != XSI_OK) {
        fprintf(stderr, "c_xcf_new: %s\n", xsi_hip_last_error());
        delete x;
        x = nullptr;
    }
    return reinterpret_cast<c_xcf*>(x);
}

void c_xcf_add_readers(c_xcf* xp, bcf_srs_t* readers) {
    Xcf* x = reinterpret_cast<Xcf*>(xp);
    if (!x || !readers) return;
    for (int i = 0; i < readers->nreaders; ++i) {
        if ((size_t)i >= x->entries.size()) x->entries.resize((size_t)i + 1);
        Entry& e = x->entries[(size_t)i];
        if (e.acc) xsi_accessor_close(e.acc);
        e.acc = nullptr;
        e.is_xsi = false;
        std::string xsi;
        if (xsi_path_of(readers->readers[i].fname, readers->readers[i].header, xsi) && file_exists(xsi)) {
            if (xsi_accessor_open(&e.acc, x->ctx, xsi.c_str()) == XSI_OK)
                e.is_xsi = true;
            else
                fprintf(stderr, "c_xcf_add_readers: %s\n", xsi_hip_last_error());
        }
    }
}

void c_xcf_update_readers(c_xcf* x, bcf_srs_t* readers) { c_xcf_add_readers(x, readers); }  // c_api.cpp:50-52

const char* c_xcf_sample_name(c_xcf* xp, int reader_id, const bcf_hdr_t* hdr, int sample_id) {
    Xcf* x = reinterpret_cast<Xcf*>(xp);
    if (x && reader_id >= 0 && (size_t)reader_id < x->entries.size() && x->entries[(size_t)reader_id].is_xsi)
        return xsi_accessor_sample_name(x->entries[(size_t)reader_id].acc, (uint64_t)sample_id);
    return hdr->samples[sample_id];
}

int c_xcf_nsamples(const char* fname) {
    // the .xsi's header when there is one (no device needed), else the BCF's sample count (c_api.cpp:58-76)
    bcf_srs_t* sr = bcf_sr_init();
    if (!bcf_sr_add_reader(sr, fname)) {
        bcf_sr_destroy(sr);
        return 0;
    }
    std::string xsi;
    int n = -1;
    if (xsi_path_of(fname, sr->readers[0].header, xsi) && file_exists(xsi)) n = (int)xsi_file_num_samples(xsi.c_str());
    if (n < 0) n = bcf_hdr_nsamples(sr->readers[0].header);
    bcf_sr_destroy(sr);
    return n;
}

int __c__xcf__get__genotypes__void(c_xcf* xp, int reader_id, const bcf_hdr_t* hdr, bcf1_t* line, void** dst, int* ndst) {
    Xcf* x = reinterpret_cast<Xcf*>(xp);
    if (!x || reader_id < 0 || (size_t)reader_id >= x->entries.size() || !x->entries[(size_t)reader_id].is_xsi)
        return bcf_get_genotypes(hdr, line, dst, ndst);  // not an xsi reader: xsi_mixed_vcf.cpp:93-99
    Entry& e = x->entries[(size_t)reader_id];
    // Accessor::position_from_bm_entry (accessor.hpp:37-46): the record's BM value
    if (bcf_unpack(line, BCF_UN_ALL)) fprintf(stderr, "bcf_unpack error\n");
    if (bcf_get_format_int32(hdr, line, "BM", &e.bm, &e.nbm) < 1) {
        fprintf(stderr, "Failed to retrieve binary matrix index position (BM key)\n");
        return -1;
    }
    const int64_t r = xsi_accessor_get_genotypes(e.acc, line->n_allele, (uint64_t)(uint32_t)e.bm[0], dst, ndst);
    if (r < 0) fprintf(stderr, "c_xcf_get_genotypes: %s\n", xsi_hip_last_error());
    return (int)r;
}

void c_xcf_delete(c_xcf* x) { delete reinterpret_cast<Xcf*>(x); }

// The -c fill loop (GtCompressorStream over BcfTraversal, gt_compressor_new.hpp:84-142, bcf_traversal.cpp:3-16):
// every record's genotypes go to xsi_writer_append; the variant-only BCF keeps the record without its samples'
// fields except one int32 FORMAT value per record, BM = block << 15 | binary-line offset (xcf.cpp:641-714).
// maf and block_len as the CLI's --maf / --variant-block-length (xsqueezeit.hpp:36-93).  0 on success.
int xsi_compress_bcf(const char* in_bcf, const char* out_xsi, double maf, uint32_t block_len, uint32_t zstd_level) {
    bcf_srs_t* sr = bcf_sr_init();
    if (!sr || !bcf_sr_add_reader(sr, in_bcf)) {
        if (sr) bcf_sr_destroy(sr);
        return XSI_ERR_IO;
    }
    bcf_hdr_t* hdr = sr->readers[0].header;
    const uint32_t n_samples = (uint32_t)bcf_hdr_nsamples(hdr);
    std::vector<const char*> names(hdr->samples, hdr->samples + n_samples);
    const std::string var_path = std::string(out_xsi) + "_var.bcf";
    htsFile* fp = hts_open(var_path.c_str(), "wb");
    bcf_hdr_t* vh = bcf_hdr_dup(hdr);
    const char* base = strrchr(out_xsi, '/');
    bcf_hdr_append(vh, (std::string("##XSI=") + (base ? base + 1 : out_xsi)).c_str());
    bcf_hdr_append(vh, "##FORMAT=<ID=BM,Number=1,Type=Integer,Description=\"Position in GT Binary Matrix\">");
    bcf_hdr_set_samples(vh, nullptr, 0);     // the variant file carries one pseudo sample holding BM
    bcf_hdr_add_sample(vh, "BIN_MATRIX_POS");
    bcf_hdr_sync(vh);
    int rc = (fp && bcf_hdr_write(fp, vh) == 0) ? XSI_OK : XSI_ERR_IO;
    xsi_hip_ctx* ctx = nullptr;
    xsi_writer* w = nullptr;
    int32_t* gt = nullptr;
    int ngt_cap = 0;
    xsi_bm_state bm;
    xsi_bm_init(&bm);
    std::vector<bcf1_t*> head;  // the first three records decide the default phase (seek_default_phased, xcf.cpp:811-836)
    std::vector<std::vector<int32_t>> head_gt;
    auto open_writer = [&](uint32_t first_ploidy) -> int {
        std::vector<const int32_t*> rows;
        std::vector<uint32_t> ngts;
        for (auto& g : head_gt) {
            rows.push_back(g.data());
            ngts.push_back((uint32_t)g.size());
        }
        xsi_encode_params p;
        p.n_samples = n_samples;
        p.block_len = block_len;
        p.mac_threshold = xsi_mac_threshold(n_samples, first_ploidy, maf);
        p.default_phased = xsi_default_phased(rows.data(), ngts.data(), (uint32_t)rows.size(), n_samples);
        p.wah_encode_missing = 0;
        p.zstd_level = zstd_level;
        if (xsi_hip_ctx_create(&ctx, 0, nullptr)) return XSI_ERR_HIP;
        return xsi_writer_open(&w, ctx, out_xsi, &p, names.data());
    };
    auto emit = [&](bcf1_t* rec, const int32_t* g, int n) -> int {
        int r = xsi_writer_append(w, g, (uint32_t)n, rec->n_allele);
        if (r) return r;
        const int64_t pos = xsi_bm_next(&bm, block_len, rec->n_allele);
        if (pos < 0) return (int)pos;
        bcf1_t* v = bcf_dup(rec);
        bcf_unpack(v, BCF_UN_ALL);
        bcf_subset(hdr, v, 0, nullptr);  // drop the samples' fields
        bcf_translate(vh, hdr, v);
        int32_t bmv = (int32_t)pos;
        bcf_update_format_int32(vh, v, "BM", &bmv, 1);
        r = bcf_write1(fp, vh, v) == 0 ? XSI_OK : XSI_ERR_IO;
        bcf_destroy(v);
        return r;
    };
    while (rc == XSI_OK && bcf_sr_next_line(sr)) {
        bcf1_t* rec = bcf_sr_get_line(sr, 0);
        bcf_unpack(rec, BCF_UN_STR);
        const int n = bcf_get_genotypes(hdr, rec, &gt, &ngt_cap);
        if (n <= 0) {
            rc = XSI_ERR_FORMAT;
            break;
        }
        if (!w) {
            head.push_back(bcf_dup(rec));
            head_gt.emplace_back(gt, gt + n);
            if (head.size() < 3) continue;
            rc = open_writer((uint32_t)head_gt[0].size() / n_samples);
            for (size_t i = 0; rc == XSI_OK && i < head.size(); ++i) rc = emit(head[i], head_gt[i].data(), (int)head_gt[i].size());
            continue;
        }
        rc = emit(rec, gt, n);
    }
    if (rc == XSI_OK && !w && !head.empty()) {  // fewer than three records in the file
        rc = open_writer((uint32_t)head_gt[0].size() / n_samples);
        for (size_t i = 0; rc == XSI_OK && i < head.size(); ++i) rc = emit(head[i], head_gt[i].data(), (int)head_gt[i].size());
    }
    if (rc == XSI_OK && w) rc = xsi_writer_finalize(w, 0);
    for (auto* r : head) bcf_destroy(r);
    if (w) xsi_writer_close(w);
    if (ctx) xsi_hip_ctx_destroy(ctx);
    free(gt);
    if (fp) hts_close(fp);
    bcf_hdr_destroy(vh);
    bcf_sr_destroy(sr);
    return rc;
}

}  // extern "C"

#endif  // XSI_HAVE_HTSLIB
