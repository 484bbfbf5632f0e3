// TEST INFRASTRUCTURE, NOT htslib.  A small working stand-in for the ~35 htslib calls that
// xsqueezeit_amd/csrc/xsi_htslib_shim.cpp makes, so that the shim's two fill loops (-c, -x) and its c_xcf_* table can
// RUN in an image that has no htslib (tests/test_shim_mock.py builds shim + this file into one test library).
// It is written against the declaration-only headers in tests/cxx/htslib_decls/ and works on GT-only VCF TEXT: every
// "BCF" it reads or writes is VCF text, whatever the open mode says; an "index" is an empty marker file.
// What it models of the real library, because the shim depends on it:
//   * bcf_get_genotypes / bcf_get_format_int32: htslib's int32 encoding ((allele + 1) << 1 | phased, "." = 0,
//     short samples padded with bcf_int32_vector_end; the first allele carries no phase bit), realloc'd *dst, the
//     number of values as return, -1 for a tag the header does not define, -3 for one the record does not carry;
//   * bcf_update_format: the tag must be defined in the header passed in (-1 otherwise), n == 0 removes the tag,
//     line->n_sample becomes bcf_hdr_nsamples(hdr), GT goes first; bcf_update_info likewise for INFO;
//   * bcf_write: refuses a record whose n_sample is not the header's sample count;
//   * bcf_hdr_set_samples(hdr, NULL, 0): no samples (a reader's header then yields records without FORMAT fields);
//   * bcf_sr_set_regions + require_index: bcf_sr_add_reader fails with errnum = idx_load_failed when "<file>.csi" is
//     missing; regions / targets are "chr", "chr:pos", "chr:beg-end" lists, 1-based inclusive, matched on POS.
// Nothing here is shipped or linked into libxsi_hip.so.  It proves the shim's control flow and its use of the C ABI,
// not compatibility with BCF2 binary files: BASELINE configs[0] still needs a machine with htslib.
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

namespace {

const int32_t VECTOR_END = INT32_MIN + 1;
const int32_t MISSING = INT32_MIN;

std::vector<std::string> split(const std::string& s, char c) {
    std::vector<std::string> out;
    size_t a = 0;
    for (;;) {
        const size_t b = s.find(c, a);
        out.push_back(s.substr(a, b == std::string::npos ? std::string::npos : b - a));
        if (b == std::string::npos) break;
        a = b + 1;
    }
    return out;
}

struct Hdr {
    bcf_hdr_t pub;  // first: the shim sees a bcf_hdr_t*
    std::vector<std::string> meta;  // "##..." lines
    std::vector<std::string> names;
    std::vector<char*> name_ptrs;
    bool drop_samples = false;  // a reader's header after bcf_hdr_set_samples(NULL)
    std::vector<bcf_hrec_t*> hrecs;  // handed out by bcf_hdr_get_hrec, owned here
    void sync() {
        name_ptrs.clear();
        for (auto& n : names) name_ptrs.push_back(const_cast<char*>(n.c_str()));
        pub.samples = name_ptrs.empty() ? nullptr : name_ptrs.data();
        pub.n[BCF_DT_SAMPLE] = (int32_t)names.size();
    }
    bool defines(const char* cls, const char* key) const {
        const std::string pre = std::string("##") + cls + "=<ID=" + key;
        for (auto& m : meta)
            if (m.compare(0, pre.size(), pre) == 0 && (m[pre.size()] == ',' || m[pre.size()] == '>')) return true;
        return false;
    }
    ~Hdr() {
        for (auto* h : hrecs) {
            free(h->key);
            free(h->value);
            delete h;
        }
    }
};
Hdr* H(const bcf_hdr_t* h) { return reinterpret_cast<Hdr*>(const_cast<bcf_hdr_t*>(h)); }

struct Fmt {
    std::string key;
    std::vector<int32_t> v;        // n_sample * per values (GT: htslib encoding)
    std::vector<std::string> raw;  // per sample, for a field that is not a list of integers
    int per = 0;
};
struct Rec {
    bcf1_t pub;  // first
    std::string chrom, id, qual, filter;
    std::vector<std::string> alleles;
    std::vector<std::pair<std::string, std::string>> info;
    std::vector<Fmt> fmt;
    void sync() {
        pub.n_allele = (uint32_t)alleles.size();
        pub.n_info = (uint32_t)info.size();
        pub.n_fmt = (uint32_t)fmt.size();
        pub.rlen = alleles.empty() ? 0 : (hts_pos_t)alleles[0].size();
    }
};
Rec* R(const bcf1_t* r) { return reinterpret_cast<Rec*>(const_cast<bcf1_t*>(r)); }

std::vector<int32_t> parse_gt(const std::string& s) {
    std::vector<int32_t> v;
    int32_t phased = 0;
    std::string tok;
    for (size_t i = 0; i <= s.size(); ++i) {
        const char ch = i < s.size() ? s[i] : '\0';
        if (ch == '/' || ch == '|' || ch == '\0') {
            v.push_back((tok.empty() || tok == ".") ? (0 | phased) : (((atoi(tok.c_str()) + 1) << 1) | phased));
            tok.clear();
            phased = ch == '|' ? 1 : 0;
        } else {
            tok += ch;
        }
    }
    return v;
}
std::string format_gt(const int32_t* v, int per) {
    std::string s;
    for (int j = 0; j < per; ++j) {
        if (v[j] == VECTOR_END) break;
        if (j) s += (v[j] & 1) ? '|' : '/';
        if ((v[j] >> 1) == 0) s += '.';
        else s += std::to_string((v[j] >> 1) - 1);
    }
    return s.empty() ? "." : s;
}

bool parse_record(const std::string& line, const Hdr& hdr, Rec& r) {
    const std::vector<std::string> t = split(line, '\t');
    if (t.size() < 8) return false;
    r.chrom = t[0];
    r.pub.pos = atoll(t[1].c_str()) - 1;
    r.pub.rid = 0;
    r.pub.qual = 0;
    r.id = t[2];
    r.alleles.clear();
    r.alleles.push_back(t[3]);
    if (t[4] != ".")
        for (auto& a : split(t[4], ',')) r.alleles.push_back(a);
    r.qual = t[5];
    r.filter = t[6];
    r.info.clear();
    if (t[7] != ".")
        for (auto& kv : split(t[7], ';')) {
            const size_t eq = kv.find('=');
            r.info.emplace_back(kv.substr(0, eq), eq == std::string::npos ? std::string() : kv.substr(eq + 1));
        }
    r.fmt.clear();
    r.pub.n_sample = 0;
    if (!hdr.drop_samples && t.size() > 9) {
        const std::vector<std::string> keys = split(t[8], ':');
        const size_t ns = t.size() - 9;
        if (ns != hdr.names.size()) return false;
        std::vector<std::vector<std::string>> cols(ns);
        for (size_t s = 0; s < ns; ++s) cols[s] = split(t[9 + s], ':');
        for (size_t k = 0; k < keys.size(); ++k) {
            Fmt f;
            f.key = keys[k];
            std::vector<std::vector<int32_t>> per(ns);
            bool ints = true;
            for (size_t s = 0; s < ns; ++s) {
                const std::string val = k < cols[s].size() ? cols[s][k] : ".";
                if (f.key == "GT") {
                    per[s] = parse_gt(val);
                } else {
                    for (auto& x : split(val, ',')) {
                        char* end = nullptr;
                        const long q = strtol(x.c_str(), &end, 10);
                        if (x == ".") per[s].push_back(MISSING);
                        else if (end && *end == '\0' && !x.empty()) per[s].push_back((int32_t)q);
                        else ints = false;
                    }
                }
                f.raw.push_back(val);
            }
            if (ints) {
                size_t mx = 1;
                for (auto& p : per) mx = std::max(mx, p.size());
                f.per = (int)mx;
                f.v.assign(ns * mx, VECTOR_END);
                for (size_t s = 0; s < ns; ++s) std::copy(per[s].begin(), per[s].end(), f.v.begin() + s * mx);
                f.raw.clear();
            }
            r.fmt.push_back(f);
        }
        r.pub.n_sample = (uint32_t)ns;
    }
    r.sync();
    return true;
}

struct Region {
    std::string chrom;
    long long beg = 1, end = (1ll << 60);
};
bool parse_regions(const char* s, std::vector<Region>& out) {
    if (!s || !*s) return false;
    for (auto& tok : split(s, ',')) {
        if (tok.empty()) return false;
        Region g;
        const size_t c = tok.rfind(':');
        if (c == std::string::npos) {
            g.chrom = tok;
        } else {
            g.chrom = tok.substr(0, c);
            const std::string range = tok.substr(c + 1);
            const size_t d = range.find('-');
            g.beg = atoll(range.substr(0, d).c_str());
            g.end = d == std::string::npos ? g.beg : (d + 1 < range.size() ? atoll(range.substr(d + 1).c_str()) : (1ll << 60));
        }
        out.push_back(g);
    }
    return true;
}

struct Srs {
    bcf_srs_t pub;  // first
    bcf_sr_t reader;
    int has_line = 0;
    bcf1_t* buf[1] = {nullptr};
    std::ifstream in;
    Hdr* hdr = nullptr;
    Rec* cur = nullptr;
    std::vector<Region> regions, targets;
    bool use_regions = false, use_targets = false;
    std::string fname;
};
Srs* S(bcf_srs_t* s) { return reinterpret_cast<Srs*>(s); }

bool exists(const std::string& p) {
    struct stat st;
    return ::stat(p.c_str(), &st) == 0;
}

}  // namespace

struct htsFile {
    FILE* f;
};

extern "C" {

htsFile* hts_open(const char* fn, const char* mode) {
    if (!fn || !mode || mode[0] != 'w') return nullptr;  // the shim only opens files for writing (readers go through bcf_sr)
    FILE* f = !strcmp(fn, "-") ? stdout : fopen(fn, "w");
    if (!f) return nullptr;
    return new htsFile{f};
}
int hts_close(htsFile* fp) {
    if (!fp) return -1;
    const int r = fp->f == stdout ? fflush(stdout) : fclose(fp->f);
    delete fp;
    return r ? -1 : 0;
}
int bcf_index_build3(const char* fn, const char* fnidx, int min_shift, int n_threads) {
    (void)min_shift;
    (void)n_threads;
    if (!fn || !exists(fn)) return -2;
    FILE* f = fopen(fnidx ? fnidx : (std::string(fn) + ".csi").c_str(), "w");
    if (!f) return -1;
    fclose(f);
    return 0;
}

// ---- headers
bcf_hdr_t* bcf_hdr_dup(const bcf_hdr_t* hdr) {
    if (!hdr) return nullptr;
    Hdr* n = new Hdr();
    memset(&n->pub, 0, sizeof(n->pub));
    n->meta = H(hdr)->meta;
    n->names = H(hdr)->names;
    n->drop_samples = false;
    if (H(hdr)->drop_samples) n->names.clear();
    n->sync();
    return &n->pub;
}
void bcf_hdr_destroy(bcf_hdr_t* h) { delete H(h); }
int bcf_hdr_write(htsFile* fp, bcf_hdr_t* h) {
    if (!fp || !h) return -1;
    for (auto& m : H(h)->meta) fprintf(fp->f, "%s\n", m.c_str());
    fprintf(fp->f, "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO");
    if (!H(h)->names.empty()) {
        fprintf(fp->f, "\tFORMAT");
        for (auto& n : H(h)->names) fprintf(fp->f, "\t%s", n.c_str());
    }
    fprintf(fp->f, "\n");
    return 0;
}
int bcf_hdr_append(bcf_hdr_t* h, const char* line) {
    if (!h || !line || strncmp(line, "##", 2)) return -1;
    std::string s(line);
    while (!s.empty() && s.back() == '\n') s.pop_back();
    H(h)->meta.push_back(s);
    return 0;
}
int bcf_hdr_sync(bcf_hdr_t* h) {
    if (!h) return -1;
    H(h)->sync();
    return 0;
}
int bcf_hdr_set_samples(bcf_hdr_t* hdr, const char* samples, int is_file) {
    (void)is_file;
    if (!hdr) return -1;
    if (samples) return -1;  // only "no samples" is modelled
    H(hdr)->drop_samples = true;
    H(hdr)->names.clear();
    H(hdr)->sync();
    return 0;
}
int bcf_hdr_add_sample(bcf_hdr_t* hdr, const char* sample) {
    if (!hdr) return -1;
    if (sample) {
        if (std::find(H(hdr)->names.begin(), H(hdr)->names.end(), sample) != H(hdr)->names.end()) return -1;  // duplicate
        H(hdr)->names.push_back(sample);
    }
    H(hdr)->sync();
    return 0;
}
void bcf_hdr_remove(bcf_hdr_t* h, int type, const char* key) {
    if (!h || !key) return;
    const char* cls = type == BCF_HL_FMT ? "FORMAT" : type == BCF_HL_INFO ? "INFO" : type == BCF_HL_FLT ? "FILTER" : nullptr;
    std::vector<std::string> keep;
    for (auto& m : H(h)->meta) {
        bool drop;
        if (cls) {
            const std::string pre = std::string("##") + cls + "=<ID=" + key;
            drop = m.compare(0, pre.size(), pre) == 0 && (m[pre.size()] == ',' || m[pre.size()] == '>');
        } else {
            const std::string pre = std::string("##") + key + "=";
            drop = m.compare(0, pre.size(), pre) == 0;
        }
        if (!drop) keep.push_back(m);
    }
    H(h)->meta.swap(keep);
}
bcf_hrec_t* bcf_hdr_get_hrec(const bcf_hdr_t* hdr, int type, const char* key, const char* value, const char* str_class) {
    (void)value;
    (void)str_class;
    if (!hdr || type != BCF_HL_GEN || !key) return nullptr;
    const std::string pre = std::string("##") + key + "=";
    for (auto& m : H(hdr)->meta)
        if (m.compare(0, pre.size(), pre) == 0) {
            bcf_hrec_t* h = new bcf_hrec_t();
            memset(h, 0, sizeof(*h));
            h->type = BCF_HL_GEN;
            h->key = strdup(key);
            h->value = strdup(m.c_str() + pre.size());
            H(hdr)->hrecs.push_back(h);
            return h;
        }
    return nullptr;
}

// ---- records
bcf1_t* bcf_dup(bcf1_t* src) {
    if (!src) return nullptr;
    Rec* n = new Rec(*R(src));
    return &n->pub;
}
void bcf_destroy(bcf1_t* v) { delete R(v); }
int bcf_unpack(bcf1_t* b, int which) {
    (void)which;
    return b ? 0 : -1;
}
int bcf_write(htsFile* fp, bcf_hdr_t* h, bcf1_t* v) {
    if (!fp || !h || !v) return -1;
    Rec* r = R(v);
    if ((uint32_t)H(h)->names.size() != v->n_sample) {
        fprintf(stderr, "[mini_hts] Broken VCF record, the number of columns at %s:%lld does not match the number of samples (%u vs %d)\n",
                r->chrom.c_str(), (long long)v->pos + 1, (unsigned)v->n_sample, (int)H(h)->names.size());
        return -1;
    }
    std::string alt;
    for (size_t i = 1; i < r->alleles.size(); ++i) alt += (i > 1 ? "," : "") + r->alleles[i];
    std::string info;
    for (size_t i = 0; i < r->info.size(); ++i)
        info += (i ? ";" : "") + r->info[i].first + (r->info[i].second.empty() ? "" : "=" + r->info[i].second);
    fprintf(fp->f, "%s\t%lld\t%s\t%s\t%s\t%s\t%s\t%s", r->chrom.c_str(), (long long)v->pos + 1, r->id.c_str(),
            r->alleles.empty() ? "." : r->alleles[0].c_str(), alt.empty() ? "." : alt.c_str(), r->qual.c_str(),
            r->filter.c_str(), info.empty() ? "." : info.c_str());
    if (v->n_sample && !r->fmt.empty()) {
        std::string keys;
        for (size_t k = 0; k < r->fmt.size(); ++k) keys += (k ? ":" : "") + r->fmt[k].key;
        fprintf(fp->f, "\t%s", keys.c_str());
        for (uint32_t s = 0; s < v->n_sample; ++s) {
            std::string col;
            for (size_t k = 0; k < r->fmt.size(); ++k) {
                const Fmt& f = r->fmt[k];
                std::string val;
                if (!f.raw.empty()) {
                    val = s < f.raw.size() ? f.raw[s] : ".";
                } else if (f.key == "GT") {
                    val = format_gt(f.v.data() + (size_t)s * f.per, f.per);
                } else {
                    for (int j = 0; j < f.per; ++j) {
                        const int32_t x = f.v[(size_t)s * f.per + j];
                        if (x == VECTOR_END) break;
                        val += (j ? "," : "") + (x == MISSING ? std::string(".") : std::to_string(x));
                    }
                    if (val.empty()) val = ".";
                }
                col += (k ? ":" : "") + val;
            }
            fprintf(fp->f, "\t%s", col.c_str());
        }
    }
    fprintf(fp->f, "\n");
    return ferror(fp->f) ? -1 : 0;
}

int bcf_get_format_values(const bcf_hdr_t* hdr, bcf1_t* line, const char* tag, void** dst, int* ndst, int type) {
    if (!hdr || !line || !tag || !dst || !ndst || type != BCF_HT_INT) return -1;
    if (!H(hdr)->defines("FORMAT", tag)) return -1;  // no such tag in the header
    for (auto& f : R(line)->fmt)
        if (f.key == tag) {
            if (!f.raw.empty()) return -2;  // not an integer field
            const int n = (int)f.v.size();
            if (*ndst < n || !*dst) {
                void* p = realloc(*dst, sizeof(int32_t) * (size_t)n);
                if (!p) return -4;
                *dst = p;
                *ndst = n;
            }
            memcpy(*dst, f.v.data(), sizeof(int32_t) * (size_t)n);
            return n;
        }
    return -3;  // the record does not carry it
}
int bcf_update_format(const bcf_hdr_t* hdr, bcf1_t* line, const char* key, const void* values, int n, int type) {
    if (!hdr || !line || !key || type != BCF_HT_INT) return -1;
    Rec* r = R(line);
    if (!H(hdr)->defines("FORMAT", key)) return n ? -1 : 0;
    auto it = std::find_if(r->fmt.begin(), r->fmt.end(), [&](const Fmt& f) { return f.key == key; });
    if (!n) {
        if (it != r->fmt.end()) r->fmt.erase(it);
        r->sync();
        return 0;
    }
    const int ns = (int)H(hdr)->names.size();
    if (ns <= 0 || n % ns) {
        fprintf(stderr, "[mini_hts] bcf_update_format(%s): %d values for %d samples\n", key, n, ns);
        abort();  // htslib asserts here
    }
    line->n_sample = (uint32_t)ns;
    Fmt f;
    f.key = key;
    f.per = n / ns;
    f.v.assign(static_cast<const int32_t*>(values), static_cast<const int32_t*>(values) + n);
    if (it != r->fmt.end()) *it = f;
    else if (!strcmp(key, "GT")) r->fmt.insert(r->fmt.begin(), f);
    else r->fmt.push_back(f);
    r->sync();
    return 0;
}
int bcf_update_info(const bcf_hdr_t* hdr, bcf1_t* line, const char* key, const void* values, int n, int type) {
    if (!hdr || !line || !key || type != BCF_HT_INT) return -1;
    Rec* r = R(line);
    if (!H(hdr)->defines("INFO", key)) return n ? -1 : 0;
    auto it = std::find_if(r->info.begin(), r->info.end(), [&](const std::pair<std::string, std::string>& kv) { return kv.first == key; });
    if (!n) {
        if (it != r->info.end()) r->info.erase(it);
        r->sync();
        return 0;
    }
    std::string val;
    for (int i = 0; i < n; ++i) val += (i ? "," : "") + std::to_string(static_cast<const int32_t*>(values)[i]);
    if (it != r->info.end()) it->second = val;
    else r->info.emplace_back(key, val);
    r->sync();
    return 0;
}

// ---- synced reader (one reader)
bcf_srs_t* bcf_sr_init(void) {
    Srs* s = new Srs();
    memset(&s->pub, 0, sizeof(s->pub));
    memset(&s->reader, 0, sizeof(s->reader));
    s->pub.readers = &s->reader;
    s->pub.has_line = &s->has_line;
    s->reader.buffer = s->buf;
    return &s->pub;
}
void bcf_sr_destroy(bcf_srs_t* readers) {
    if (!readers) return;
    Srs* s = S(readers);
    delete s->cur;
    delete s->hdr;
    delete s;
}
int bcf_sr_set_regions(bcf_srs_t* readers, const char* regions, int is_file) {
    if (!readers || is_file) return -1;
    Srs* s = S(readers);
    if (!parse_regions(regions, s->regions)) return -1;
    s->use_regions = true;
    return 0;
}
int bcf_sr_set_targets(bcf_srs_t* readers, const char* targets, int is_file, int alleles) {
    (void)alleles;
    if (!readers || is_file) return -1;
    Srs* s = S(readers);
    if (!parse_regions(targets, s->targets)) return -1;
    s->use_targets = true;
    return 0;
}
int bcf_sr_add_reader(bcf_srs_t* readers, const char* fname) {
    if (!readers || !fname) return 0;
    Srs* s = S(readers);
    if (s->pub.nreaders) return 0;  // one reader
    s->in.open(fname);
    if (!s->in) {
        s->pub.errnum = 0;  // open_failed
        return 0;
    }
    if ((s->pub.require_index || s->use_regions) && !exists(std::string(fname) + ".csi")) {
        s->pub.errnum = 2;  // idx_load_failed
        s->in.close();
        return 0;
    }
    s->hdr = new Hdr();
    memset(&s->hdr->pub, 0, sizeof(s->hdr->pub));
    std::string line;
    bool got = false;
    while (std::getline(s->in, line)) {
        if (line.compare(0, 2, "##") == 0) {
            s->hdr->meta.push_back(line);
        } else if (line.compare(0, 6, "#CHROM") == 0) {
            const std::vector<std::string> t = split(line, '\t');
            for (size_t i = 9; i < t.size(); ++i) s->hdr->names.push_back(t[i]);
            got = true;
            break;
        } else {
            break;
        }
    }
    if (!got) {
        s->pub.errnum = 5;  // header_error
        return 0;
    }
    s->hdr->sync();
    s->fname = fname;
    s->reader.fname = s->fname.c_str();
    s->reader.header = &s->hdr->pub;
    s->pub.nreaders = 1;
    return 1;
}
int bcf_sr_next_line(bcf_srs_t* readers) {
    if (!readers) return 0;
    Srs* s = S(readers);
    s->has_line = 0;
    if (!s->pub.nreaders) return 0;
    std::string line;
    while (std::getline(s->in, line)) {
        if (line.empty()) continue;
        Rec* r = new Rec();
        memset(&r->pub, 0, sizeof(r->pub));
        if (!parse_record(line, *s->hdr, *r)) {
            delete r;
            s->pub.errnum = 8;  // vcf_parse_error
            return 0;
        }
        auto inside = [&](const std::vector<Region>& v) {
            for (auto& g : v)
                if (g.chrom == r->chrom && r->pub.pos + 1 >= g.beg && r->pub.pos + 1 <= g.end) return true;
            return false;
        };
        if ((s->use_regions && !inside(s->regions)) || (s->use_targets && !inside(s->targets))) {
            delete r;
            continue;
        }
        delete s->cur;
        s->cur = r;
        s->buf[0] = &r->pub;
        s->has_line = 1;
        return 1;
    }
    return 0;
}

}  // extern "C"
