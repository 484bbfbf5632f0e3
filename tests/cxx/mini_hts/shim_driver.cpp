// TEST INFRASTRUCTURE: walks a variant-only "BCF" (VCF text, see mini_hts.cpp) through the shim's c_xcf_* table the way
// an existing htslib program does (the reference's include/c_api.h:38-93 usage: c_xcf_new, bcf_sr_add_reader,
// c_xcf_add_readers, then per record c_xcf_get_genotypes instead of bcf_get_genotypes), so that tests/test_shim_mock.py
// can drive it through ctypes without touching htslib structs from Python.
#include <htslib/synced_bcf_reader.h>
#include <htslib/vcf.h>

#include <cstdlib>
#include <cstring>

extern "C" {
typedef void* c_xcf;
c_xcf* c_xcf_new(void);
void c_xcf_add_readers(c_xcf* x, bcf_srs_t* readers);
const char* c_xcf_sample_name(c_xcf* x, int reader_id, const bcf_hdr_t* hdr, int sample_id);
int c_xcf_nsamples(const char* fname);
int __c__xcf__get__genotypes__void(c_xcf* x, int reader_id, const bcf_hdr_t* hdr, bcf1_t* line, void** dst, int* ndst);
void c_xcf_delete(c_xcf* x);

// returns the number of records, < 0 on error; values of all records back to back in `out` (at most cap),
// their counts in `per_line` (at most cap_lines), the name of sample `sample_id` in `name`
long shim_test_cxcf_walk(const char* var_path, int32_t* out, long cap, int* per_line, long cap_lines, int sample_id,
                         char* name, int name_cap, int* n_samples) {
    *n_samples = c_xcf_nsamples(var_path);
    c_xcf* x = c_xcf_new();
    if (!x) return -1;
    bcf_srs_t* sr = bcf_sr_init();
    long rc = 0;
    if (!sr || !bcf_sr_add_reader(sr, var_path)) rc = -2;
    int32_t* gt = nullptr;
    int ngt = 0;
    long used = 0, lines = 0;
    if (!rc) {
        c_xcf_add_readers(x, sr);
        const char* nm = c_xcf_sample_name(x, 0, sr->readers[0].header, sample_id);
        if (name_cap > 0) {
            strncpy(name, nm ? nm : "", (size_t)name_cap - 1);
            name[name_cap - 1] = '\0';
        }
        while (bcf_sr_next_line(sr)) {
            bcf1_t* line = bcf_sr_get_line(sr, 0);
            const int n = __c__xcf__get__genotypes__void(x, 0, sr->readers[0].header, line, (void**)&gt, &ngt);
            if (n <= 0 || used + n > cap || lines >= cap_lines) {
                rc = -3;
                break;
            }
            memcpy(out + used, gt, sizeof(int32_t) * (size_t)n);
            used += n;
            per_line[lines++] = n;
        }
    }
    free(gt);
    if (sr) bcf_sr_destroy(sr);
    c_xcf_delete(x);
    return rc ? rc : lines;
}
}
