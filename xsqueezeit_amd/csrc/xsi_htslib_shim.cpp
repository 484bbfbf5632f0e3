// xsi_htslib_shim.cpp — the htslib-facing layer above libxsi_hip.so's C ABI.
//
// Built with a body only on request (make HTSLIB=1 -> -DXSI_HAVE_HTSLIB -lhts): this image has no htslib, so the
// default library carries the stub at the bottom (xsi_htslib_shim_available() == 0, xsi_compress_bcf /
// xsi_decompress_bcf return XSI_ERR_UNSUPPORTED).  What CAN be checked here is checked: tests/test_host.py compiles
// this file with -fsyntax-only -DXSI_HAVE_HTSLIB against declaration-only prototypes of the htslib functions it uses
// (tests/cxx/htslib_decls/), and tests/test_shim_mock.py builds it against a small working stand-in for those ~35 calls
// (tests/cxx/mini_hts/: GT-only VCF text, test infrastructure) and RUNS both fill loops and the c_xcf_* table on the
// GPU: -c of the reference's fixtures == the oracle's files, -x -Ov == the input text, -s / -r / -t / -Ox.
// It has never run against htslib itself: BASELINE configs[0] (a real BCF2 file through -c / -x) stays untested.
//
// With htslib it exports, under their reference names, the symbols an existing HTSLIB caller links against
//
//   c_xcf_new / c_xcf_add_readers / c_xcf_update_readers / c_xcf_sample_name / c_xcf_nsamples /
//   __c__xcf__get__genotypes__void / c_xcf_delete            (include/c_api.h:38-93, c_api.cpp:37-85; the table of
//                                                             readers behind them: xsi_mixed_vcf.cpp:46-106)
// and the two fill loops of the CLI:
//
//   xsi_compress_bcf    -c: the variant-only BCF with the BM field (replace_samples_by_pos_in_binary_matrix,
//                           xcf.cpp:641-714) + BcfTraversal::traverse (bcf_traversal.cpp:3-16) feeding
//                           XsiFactoryInterface::append (gt_compressor_new.hpp:84-142)
//   xsi_decompress_bcf  -x: NewDecompressor (gt_decompressor_new.hpp:113-124 reader set-up with -r/-R/-t, :157-206
//                           decompress_inner_loop, :209-238 sample selection, :275-320 record update + write,
//                           :241-273 the -Ox re-encode, :432-543 output header)
//
// Every genotype goes through xsi_writer_* / xsi_accessor_* of include/xsi_hip.h; nothing here computes.
#include "../../include/xsi_hip.h"

#ifdef XSI_HAVE_HTSLIB

#include <htslib/hts.h>
#include <htslib/synced_bcf_reader.h>
#include <htslib/vcf.h>
#include <sys/stat.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

extern "C" int xsi_htslib_shim_available(void) { return 1; }

namespace {

const char* const VAR_EXT = "_var.bcf";  // XSI_BCF_VAR_EXTENSION

std::string base_name(const std::string& p) {
    const size_t slash = p.find_last_of('/');
    return slash == std::string::npos ? p : p.substr(slash + 1);
}

// the .xsi that belongs to a variant-only BCF: its "##XSI=<basename>" header line next to the BCF, else the name
// with "_var.bcf" cut off (Accessor::get_filename_from_variant_file, accessor.hpp:89-111)
bool xsi_path_of(const char* bcf_path, bcf_hdr_t* hdr, std::string& out) {
    std::string p(bcf_path ? bcf_path : "");
    if (hdr) {
        bcf_hrec_t* h = bcf_hdr_get_hrec(hdr, BCF_HL_GEN, "XSI", nullptr, nullptr);
        if (h && h->value) {
            const size_t slash = p.find_last_of('/');
            out = (slash == std::string::npos ? std::string(".") : p.substr(0, slash)) + "/" + h->value;
            return true;
        }
    }
    const size_t pos = p.find(VAR_EXT);
    if (pos == std::string::npos) return false;
    out = p.substr(0, pos);
    return true;
}
bool file_exists(const std::string& p) {
    struct stat st;
    return ::stat(p.c_str(), &st) == 0;
}
// create_index_file (xcf.cpp:39-60): a CSI index (min_shift 14) next to the variant BCF; -r / -R need it
int build_index(const std::string& bcf_path) {
    const int r = bcf_index_build3(bcf_path.c_str(), nullptr, 14, 1);
    if (r == 0) return XSI_OK;
    fprintf(stderr, r == -2 ? "index: failed to open %s\n" : r == -3 ? "index: %s is in a format that cannot be usefully indexed\n"
                                                                    : "index: failed to create index for %s\n", bcf_path.c_str());
    return XSI_ERR_IO;
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

// Accessor::position_from_bm_entry (accessor.hpp:37-46).  <0 on error.
int64_t bm_of_record(const bcf_hdr_t* hdr, bcf1_t* line, int32_t** scratch, int* n_scratch) {
    if (bcf_unpack(line, BCF_UN_ALL)) fprintf(stderr, "bcf_unpack error\n");
    if (bcf_get_format_int32(hdr, line, "BM", scratch, n_scratch) < 1) {
        fprintf(stderr, "Failed to retrieve binary matrix index position (BM key)\n");
        return -1;
    }
    return (int64_t)(uint32_t)(*scratch)[0];
}

// one synced reader over `path`, optionally restricted (initialize_bcf_file_reader[_with_region/_with_target],
// xcf.cpp:39-149)
bcf_srs_t* open_reader(const char* path, const char* regions, int regions_is_file, const char* targets) {
    bcf_srs_t* sr = bcf_sr_init();
    if (!sr) return nullptr;
    if (regions && *regions) {
        sr->require_index = 1;
        if (bcf_sr_set_regions(sr, regions, regions_is_file) < 0) {
            fprintf(stderr, "Failed to read the regions: %s\n", regions);
            bcf_sr_destroy(sr);
            return nullptr;
        }
    } else if (targets && *targets) {
        if (bcf_sr_set_targets(sr, targets, 0, 0) < 0) {
            fprintf(stderr, "Failed to read the targets: %s\n", targets);
            bcf_sr_destroy(sr);
            return nullptr;
        }
    }
    if (!bcf_sr_add_reader(sr, path)) {
        fprintf(stderr, "Failed to read from %s\n", path);
        bcf_sr_destroy(sr);
        return nullptr;
    }
    return sr;
}

// -s "A,B,C" / "^A,B" (NewDecompressor::enable_select_samples, gt_decompressor_new.hpp:322-366): listed samples in
// the order of the option, or every sample that is not listed in file order; unknown names are ignored
std::vector<uint32_t> select_samples(const std::vector<std::string>& all, const char* option) {
    std::string opt(option);
    const bool inverse = !opt.empty() && opt[0] == '^';
    if (inverse) opt.erase(0, 1);
    std::vector<std::string> listed;
    std::stringstream ss(opt);
    for (std::string tok; std::getline(ss, tok, ',');)
        if (!tok.empty()) listed.push_back(tok);
    std::vector<uint32_t> use;
    if (inverse) {
        for (size_t i = 0; i < all.size(); ++i)
            if (std::find(listed.begin(), listed.end(), all[i]) == listed.end()) use.push_back((uint32_t)i);
    } else {
        for (const auto& s : listed) {
            auto it = std::find(all.begin(), all.end(), s);
            if (it != all.end()) use.push_back((uint32_t)(it - all.begin()));
        }
    }
    return use;
}

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
    if (sample_id < 0) return nullptr;
    if (x && reader_id >= 0 && (size_t)reader_id < x->entries.size() && x->entries[(size_t)reader_id].is_xsi)
        return xsi_accessor_sample_name(x->entries[(size_t)reader_id].acc, (uint64_t)sample_id);
    if (!hdr || sample_id >= bcf_hdr_nsamples(hdr)) return nullptr;
    return hdr->samples[sample_id];
}

int c_xcf_nsamples(const char* fname) {
    // the .xsi's header when there is one (no device needed), else the BCF's sample count (c_api.cpp:58-76)
    bcf_srs_t* sr = bcf_sr_init();
    if (!sr) return 0;
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
    const int64_t bm = bm_of_record(hdr, line, &e.bm, &e.nbm);
    if (bm < 0) return -1;
    const int64_t r = xsi_accessor_get_genotypes(e.acc, line->n_allele, (uint64_t)bm, dst, ndst);
    if (r < 0) fprintf(stderr, "c_xcf_get_genotypes: %s\n", xsi_hip_last_error());
    return (int)r;
}

void c_xcf_delete(c_xcf* x) { delete reinterpret_cast<Xcf*>(x); }

// ---------------------------------------------------------------------------------------------------------------------
// -c.  Two passes over the input like the reference's two threads (xsqueezeit.cpp:120-148), one after the other:
//  1. the variant-only BCF: the reader drops every sample (bcf_hdr_set_samples(NULL)), the output header gets the one
//     pseudo sample BIN_MATRIX_POS, FORMAT/BM and ##XSI=<basename>, every record leaves with n_sample = 1 and
//     BM = block << 15 | binary-line offset; written "wz" as the reference writes it (xcf.cpp:641-714);
//  2. the genotypes: bcf_get_genotypes per record -> xsi_writer_append; MAC threshold from the first record's ploidy,
//     default phase from the first three records (gt_compressor_new.hpp:84-109, xcf.cpp:811-836).
static int write_variant_bcf(const char* in_bcf, const std::string& var_path, const char* out_xsi, uint32_t block_len) {
    bcf_srs_t* sr = open_reader(in_bcf, nullptr, 0, nullptr);
    if (!sr) return XSI_ERR_IO;
    htsFile* fp = hts_open(var_path.c_str(), "wz");
    if (!fp) {
        bcf_sr_destroy(sr);
        return XSI_ERR_IO;
    }
    int rc = XSI_OK;
    bcf_hdr_t* hdr = nullptr;
    if (bcf_hdr_set_samples(sr->readers[0].header, nullptr, 0) < 0) {
        fprintf(stderr, "xsi_compress_bcf: %s: could not drop the sample columns from the header copy\n", in_bcf);
        rc = XSI_ERR_FORMAT;
    }
    if (rc == XSI_OK && !(hdr = bcf_hdr_dup(sr->readers[0].header))) rc = XSI_ERR_IO;
    if (rc == XSI_OK) {
        if (bcf_hdr_add_sample(hdr, "BIN_MATRIX_POS") < 0 ||
            bcf_hdr_append(hdr, "##FORMAT=<ID=BM,Number=1,Type=Integer,Description=\"Position in GT Binary Matrix\">") < 0 ||
            bcf_hdr_append(hdr, (std::string("##XSI=") + base_name(out_xsi)).c_str()) < 0)
            rc = XSI_ERR_FORMAT;
        else if (bcf_hdr_sync(hdr) < 0)
            fprintf(stderr, "xsi_compress_bcf: header of %s did not re-synchronise after the BM additions (continuing)\n", var_path.c_str());
    }
    if (rc == XSI_OK && bcf_hdr_write(fp, hdr) < 0) {
        fprintf(stderr, "xsi_compress_bcf: %s: header write refused by htslib\n", var_path.c_str());
        rc = XSI_ERR_IO;
    }
    xsi_bm_state bm;
    xsi_bm_init(&bm);
    while (rc == XSI_OK && bcf_sr_next_line(sr)) {
        bcf1_t* rec = bcf_dup(bcf_sr_get_line(sr, 0));
        if (!rec) {
            rc = XSI_ERR_IO;
            break;
        }
        bcf_unpack(rec, BCF_UN_STR);
        rec->n_sample = 1;
        const int64_t pos = xsi_bm_next(&bm, block_len, rec->n_allele);
        if (pos < 0) {
            rc = (int)pos;
        } else {
            int32_t v = (int32_t)pos;
            if (bcf_update_format_int32(hdr, rec, "BM", &v, 1) < 0 || bcf_write1(fp, hdr, rec) < 0) rc = XSI_ERR_IO;
        }
        bcf_destroy(rec);
    }
    if (hts_close(fp) < 0 && rc == XSI_OK) rc = XSI_ERR_IO;
    if (hdr) bcf_hdr_destroy(hdr);
    bcf_sr_destroy(sr);
    return rc;
}

int xsi_compress_bcf(const char* in_bcf, const char* out_xsi, double maf, uint32_t block_len, uint32_t zstd_level) {
    if (!in_bcf || !out_xsi || !block_len) return XSI_ERR_ARG;
    int rc = write_variant_bcf(in_bcf, std::string(out_xsi) + VAR_EXT, out_xsi, block_len);
    if (rc == XSI_OK) rc = build_index(std::string(out_xsi) + VAR_EXT);  // xsqueezeit.cpp:127
    if (rc) return rc;
    bcf_srs_t* sr = open_reader(in_bcf, nullptr, 0, nullptr);
    if (!sr) return XSI_ERR_IO;
    bcf_hdr_t* hdr = sr->readers[0].header;
    const uint32_t n_samples = (uint32_t)bcf_hdr_nsamples(hdr);
    if (!n_samples) {
        bcf_sr_destroy(sr);
        return XSI_ERR_FORMAT;
    }
    std::vector<const char*> names(hdr->samples, hdr->samples + n_samples);
    xsi_hip_ctx* ctx = nullptr;
    xsi_writer* w = nullptr;
    int32_t* gt = nullptr;
    int ngt_cap = 0;
    struct Held {
        std::vector<int32_t> gt;
        uint32_t n_allele;
    };
    std::vector<Held> head;  // the first three records decide the default phase (seek_default_phased, xcf.cpp:811-836)
    auto open_writer = [&]() -> int {
        std::vector<const int32_t*> rows;
        std::vector<uint32_t> ngts;
        for (auto& h : head) {
            rows.push_back(h.gt.data());
            ngts.push_back((uint32_t)h.gt.size());
        }
        const int32_t dp = xsi_default_phased(rows.data(), ngts.data(), (uint32_t)rows.size(), n_samples);
        if (dp < 0) return (int)dp;
        xsi_encode_params p;
        p.n_samples = n_samples;
        p.block_len = block_len;
        p.mac_threshold = xsi_mac_threshold(n_samples, (uint32_t)head[0].gt.size() / n_samples, maf);
        p.default_phased = dp;
        p.wah_encode_missing = 0;
        p.zstd_level = zstd_level;
        int r = xsi_hip_ctx_create(&ctx, 0, nullptr);
        if (r) return r;
        r = xsi_writer_open(&w, ctx, out_xsi, &p, names.data());
        for (size_t i = 0; r == XSI_OK && i < head.size(); ++i)
            r = xsi_writer_append(w, head[i].gt.data(), (uint32_t)head[i].gt.size(), head[i].n_allele);
        return r;
    };
    while (rc == XSI_OK && bcf_sr_next_line(sr)) {
        bcf1_t* rec = bcf_sr_get_line(sr, 0);
        bcf_unpack(rec, BCF_UN_STR);
        const int n = bcf_get_genotypes(hdr, rec, &gt, &ngt_cap);
        if (n <= 0) {
            fprintf(stderr, "Failed to get genotypes (line %lld)\n", (long long)rec->pos + 1);
            rc = XSI_ERR_FORMAT;
            break;
        }
        if (!w) {
            head.push_back(Held{std::vector<int32_t>(gt, gt + n), rec->n_allele});
            if (head.size() == 3) rc = open_writer();
            continue;
        }
        rc = xsi_writer_append(w, gt, (uint32_t)n, rec->n_allele);
    }
    if (rc == XSI_OK && !w && !head.empty()) rc = open_writer();  // fewer than three records in the file
    if (rc == XSI_OK && w) rc = xsi_writer_finalize(w, 0);
    if (rc && rc != XSI_ERR_FORMAT) fprintf(stderr, "xsi_compress_bcf: %s\n", xsi_hip_last_error());
    if (w) xsi_writer_close(w);
    if (ctx) xsi_hip_ctx_destroy(ctx);
    free(gt);
    bcf_sr_destroy(sr);
    return rc;
}

// ---------------------------------------------------------------------------------------------------------------------
// -x.  NewDecompressor::decompress (gt_decompressor_new.hpp): walk the variant BCF (all of it, or -r / -R / -t), fetch
// each record's genotypes by its BM value, put them back into the record and write it.
int xsi_decompress_bcf(const char* in_xsi, const char* out_path, const xsi_decompress_options* opt_in) {
    if (!in_xsi || !out_path) return XSI_ERR_ARG;
    xsi_decompress_options opt;
    memset(&opt, 0, sizeof(opt));
    if (opt_in) opt = *opt_in;
    const char otype = opt.output_type ? opt.output_type : 'b';
    const bool to_xsi = otype == 'x';
    const char* flags = "wb";  // create_output_file, :432-470
    if (!strcmp(out_path, "-") && opt.fast_pipe)
        flags = "wbu";
    else if (otype == 'u')
        flags = "wbu";
    else if (otype == 'z')
        flags = "wz";
    else if (otype == 'v')
        flags = "w";
    else if (otype != 'b' && otype != 'x')
        fprintf(stderr, "Unrecognized output type : %c\nWill default to BCF\n", otype);

    xsi_hip_ctx* ctx = nullptr;
    xsi_accessor* acc = nullptr;
    xsi_writer* w = nullptr;
    bcf_srs_t* sr = nullptr;
    htsFile* fp = nullptr;
    bcf_hdr_t* hdr = nullptr;
    int32_t *bm = nullptr, *genotypes = nullptr;
    int nbm = 0;
    int rc = xsi_hip_ctx_create(&ctx, 0, nullptr);
    if (rc == XSI_OK) rc = xsi_accessor_open(&acc, ctx, in_xsi);
    std::vector<std::string> sample_list;
    std::vector<uint32_t> use;
    bool select = false;
    uint64_t hap_samples = 0;
    if (rc == XSI_OK) {
        hap_samples = xsi_accessor_hap_samples(acc);
        const uint64_t ns = xsi_accessor_num_samples(acc);
        for (uint64_t i = 0; i < ns; ++i) sample_list.push_back(xsi_accessor_sample_name(acc, i));
        if (opt.samples && *opt.samples) {
            use = select_samples(sample_list, opt.samples);
            select = true;
        } else {
            for (uint32_t i = 0; i < (uint32_t)ns; ++i) use.push_back(i);
        }
        if (use.empty()) {  // decompress_checks, :417-424
            fprintf(stderr, "No samples to extract\n");
            rc = XSI_ERR_ARG;
        }
    }
    if (rc == XSI_OK && select) rc = xsi_accessor_set_sample_subset(acc, use.data(), (uint32_t)use.size());
    if (rc == XSI_OK) {
        // twice the sample count: room for a diploid line of a file whose header says ploidy 1
        genotypes = (int32_t*)malloc(sizeof(int32_t) * (size_t)(2 * sample_list.size() > hap_samples ? 2 * sample_list.size() : hap_samples) + 8);
        if (!genotypes) rc = XSI_ERR_ARG;
    }
    const std::string var_in = std::string(in_xsi) + VAR_EXT;
    if (rc == XSI_OK && !file_exists(var_in)) {
        fprintf(stderr, "File %s is missing and required to decompress the .xsi\n", var_in.c_str());
        rc = XSI_ERR_IO;
    }
    if (rc == XSI_OK && !file_exists(var_in + ".csi")) {  // xsqueezeit.cpp:174-178
        fprintf(stderr, "Index for %s is missing, reindexing now...\n", var_in.c_str());
        rc = build_index(var_in);
    }
    if (rc == XSI_OK && !(sr = open_reader(var_in.c_str(), opt.regions, opt.regions_is_file, opt.targets))) rc = XSI_ERR_IO;
    std::string bcf_out(out_path);
    if (to_xsi) bcf_out += VAR_EXT;
    if (rc == XSI_OK && !(fp = hts_open(bcf_out.c_str(), flags))) {
        fprintf(stderr, "Could not open %s\n", bcf_out.c_str());
        rc = XSI_ERR_IO;
    }
    bcf_hdr_t* vhdr = rc == XSI_OK ? sr->readers[0].header : nullptr;
    uint32_t block_len = 8192;
    if (rc == XSI_OK && !(hdr = bcf_hdr_dup(vhdr))) rc = XSI_ERR_IO;
    if (rc == XSI_OK) {
        bcf_hdr_remove(hdr, BCF_HL_GEN, "XSI");
        if (to_xsi) {
            // the new file: ##XSI of the new name, BM stays; the factory gets the (selected) sample names and the old
            // file's block length, default phase and zstd flag (:471-500)
            if (bcf_hdr_append(hdr, (std::string("##XSI=") + base_name(out_path)).c_str()) < 0) rc = XSI_ERR_FORMAT;
            uint8_t h[256];
            FILE* f = fopen(in_xsi, "rb");
            if (!f || fread(h, 1, 256, f) != 256) rc = XSI_ERR_IO;
            if (f) fclose(f);
            if (rc == XSI_OK) {
                uint32_t ss_rate;
                memcpy(&ss_rate, h + 56, 4);
                block_len = ss_rate ? ss_rate : 8192;
                const uint32_t ploidy = h[12];
                xsi_encode_params p;
                p.n_samples = (uint32_t)use.size();
                p.block_len = block_len;
                p.mac_threshold = xsi_mac_threshold((uint32_t)use.size(), ploidy, opt.maf);
                p.default_phased = (h[16] & 4u) ? 1 : 0;
                p.wah_encode_missing = 0;
                p.zstd_level = (opt.zstd_level || (h[17] & 4u)) ? (opt.zstd_level ? opt.zstd_level : 7u) : 0u;
                std::vector<const char*> names;
                for (uint32_t i : use) names.push_back(sample_list[i].c_str());
                rc = xsi_writer_open(&w, ctx, out_path, &p, names.data());
            }
        } else {
            bcf_hdr_remove(hdr, BCF_HL_FMT, "BM");
            if (bcf_hdr_set_samples(hdr, nullptr, 0) < 0) {
                fprintf(stderr, "Failed to remove samples from header for %s\n", bcf_out.c_str());
                rc = XSI_ERR_FORMAT;
            }
            for (size_t i = 0; rc == XSI_OK && i < use.size(); ++i)
                if (bcf_hdr_add_sample(hdr, sample_list[use[i]].c_str()) < 0) rc = XSI_ERR_FORMAT;
        }
    }
    if (rc == XSI_OK) {
        bcf_hdr_add_sample(hdr, nullptr);  // to update internal structures (:519-520)
        if (bcf_hdr_sync(hdr) < 0) fprintf(stderr, "bcf_hdr_sync() failed ...\n");
        const bool is_vcf = otype == 'v' || otype == 'z';
        if (!(is_vcf && opt.no_header) && bcf_hdr_write(fp, hdr) < 0) {
            fprintf(stderr, "Could not write header to file %s\n", bcf_out.c_str());
            rc = XSI_ERR_IO;
        }
    }
    // decompress_inner_loop (:157-206)
    xsi_bm_state newbm;
    xsi_bm_init(&newbm);
    std::vector<int32_t> ac;
    const uint64_t gt_cap = 2 * sample_list.size() > hap_samples ? 2 * sample_list.size() : hap_samples;
    while (rc == XSI_OK && bcf_sr_next_line(sr)) {
        bcf1_t* rec = bcf_sr_get_line(sr, 0);
        const int64_t pos = bm_of_record(vhdr, rec, &bm, &nbm);
        if (pos < 0) {
            rc = XSI_ERR_FORMAT;
            break;
        }
        const uint32_t n_alt = rec->n_allele ? rec->n_allele - 1u : 0u;
        int64_t n;  // values of this line: samples x the line's ploidy
        if (select) {
            ac.assign(n_alt ? n_alt : 1u, 0);
            n = xsi_accessor_fill_selected_genotypes(acc, genotypes, gt_cap, rec->n_allele, (uint64_t)pos, ac.data());
        } else {
            n = xsi_accessor_fill_genotype_array(acc, genotypes, gt_cap, rec->n_allele, (uint64_t)pos);
        }
        if (n < 0) {
            rc = (int)n;
            break;
        }
        const int64_t line_ploidy = n / (int64_t)use.size();
        if (line_ploidy < 1 || line_ploidy > 2) {
            fprintf(stderr, line_ploidy < 1 ? "Detected ploidy of 0 !\n" : "Cannot handle ploidy above 2 !\n");
            rc = XSI_ERR_FORMAT;
            break;
        }
        if (select) {  // recompute AC / AN as bcftools view -s does (:225-236, :297-303)
            // (results not checked, as in the reference: a file whose header defines no AC / AN keeps none)
            int32_t an = (int32_t)n;
            bcf_update_info_int32(hdr, rec, "AC", ac.data(), (int)n_alt);
            bcf_update_info_int32(hdr, rec, "AN", &an, 1);
        }
        if (to_xsi) {
            // update_and_write_xsi (:241-273): the record keeps BM, now pointing into the NEW file
            const int64_t np = xsi_bm_next(&newbm, block_len, rec->n_allele);
            if (np < 0) {
                rc = (int)np;
                break;
            }
            int32_t v = (int32_t)np;
            if (bcf_update_format_int32(vhdr, rec, "BM", &v, 1) < 0 || bcf_write1(fp, hdr, rec) < 0) {
                fprintf(stderr, "Failed to write record\n");
                rc = XSI_ERR_IO;
                break;
            }
            rc = xsi_writer_append(w, genotypes, (uint32_t)n, rec->n_allele);
        } else {
            // update_and_write_bcf_record (:275-320): drop BM, set GT
            if (bcf_update_format_int32(vhdr, rec, "BM", nullptr, 0) < 0 || bcf_update_genotypes(hdr, rec, genotypes, (int)n) < 0) {
                fprintf(stderr, "Failed to update genotypes\n");
                rc = XSI_ERR_FORMAT;
                break;
            }
            if (bcf_write1(fp, hdr, rec) < 0) {
                fprintf(stderr, "Failed to write record\n");
                rc = XSI_ERR_IO;
            }
        }
    }
    if (rc == XSI_OK && w) rc = xsi_writer_finalize(w, 0);
    if (rc && rc != XSI_ERR_FORMAT && rc != XSI_ERR_IO) fprintf(stderr, "xsi_decompress_bcf: %s\n", xsi_hip_last_error());
    if (w) xsi_writer_close(w);
    if (fp && hts_close(fp) < 0 && rc == XSI_OK) rc = XSI_ERR_IO;
    if (hdr) bcf_hdr_destroy(hdr);
    if (sr) bcf_sr_destroy(sr);
    free(bm);
    free(genotypes);
    if (acc) xsi_accessor_close(acc);
    if (ctx) xsi_hip_ctx_destroy(ctx);
    return rc;
}

}  // extern "C"

#else  // !XSI_HAVE_HTSLIB: the default build (this image has no htslib)

extern "C" {
int xsi_htslib_shim_available(void) { return 0; }
int xsi_compress_bcf(const char*, const char*, double, uint32_t, uint32_t) { return XSI_ERR_UNSUPPORTED; }
int xsi_decompress_bcf(const char*, const char*, const xsi_decompress_options*) { return XSI_ERR_UNSUPPORTED; }
}

#endif  // XSI_HAVE_HTSLIB
