/* TEST INFRASTRUCTURE, NOT htslib: see hts.h in this directory.  Prototypes of the vcf.h names the shim uses. */
#ifndef XSI_TEST_HTSLIB_DECLS_VCF_H
#define XSI_TEST_HTSLIB_DECLS_VCF_H
#include <stdint.h>
#include "hts.h"
#ifdef __cplusplus
extern "C" {
#endif
#define BCF_HL_FLT 0
#define BCF_HL_INFO 1
#define BCF_HL_FMT 2
#define BCF_HL_CTG 3
#define BCF_HL_STR 4
#define BCF_HL_GEN 5
#define BCF_HT_INT 1
#define BCF_DT_ID 0
#define BCF_DT_CTG 1
#define BCF_DT_SAMPLE 2
#define BCF_UN_STR 1
#define BCF_UN_FLT 2
#define BCF_UN_INFO 4
#define BCF_UN_SHR (BCF_UN_STR | BCF_UN_FLT | BCF_UN_INFO)
#define BCF_UN_FMT 8
#define BCF_UN_ALL (BCF_UN_SHR | BCF_UN_FMT)

typedef struct bcf_hrec_t {
    int type;
    char* key;
    char* value;
    int nkeys;
    char **keys, **vals;
} bcf_hrec_t;

typedef struct bcf_hdr_t {
    int32_t n[3];
    void* id[3];
    void* dict[3];
    char** samples;
    bcf_hrec_t** hrec;
    int nhrec, dirty;
} bcf_hdr_t;

typedef struct bcf1_t {
    hts_pos_t pos;
    hts_pos_t rlen;
    int32_t rid;
    float qual;
    uint32_t n_info : 16, n_allele : 16;
    uint32_t n_fmt : 8, n_sample : 24;
} bcf1_t;

#define bcf_hdr_nsamples(hdr) (hdr)->n[BCF_DT_SAMPLE]

bcf_hdr_t* bcf_hdr_dup(const bcf_hdr_t* hdr);
void bcf_hdr_destroy(bcf_hdr_t* h);
int bcf_hdr_write(htsFile* fp, bcf_hdr_t* h);
int bcf_hdr_append(bcf_hdr_t* h, const char* line);
int bcf_hdr_sync(bcf_hdr_t* h);
int bcf_hdr_set_samples(bcf_hdr_t* hdr, const char* samples, int is_file);
int bcf_hdr_add_sample(bcf_hdr_t* hdr, const char* sample);
void bcf_hdr_remove(bcf_hdr_t* h, int type, const char* key);
bcf_hrec_t* bcf_hdr_get_hrec(const bcf_hdr_t* hdr, int type, const char* key, const char* value, const char* str_class);

bcf1_t* bcf_dup(bcf1_t* src);
void bcf_destroy(bcf1_t* v);
int bcf_unpack(bcf1_t* b, int which);
int bcf_write(htsFile* fp, bcf_hdr_t* h, bcf1_t* v);
#define bcf_write1(fp, h, v) bcf_write((fp), (h), (v))

int bcf_index_build3(const char* fn, const char* fnidx, int min_shift, int n_threads);

int bcf_get_format_values(const bcf_hdr_t* hdr, bcf1_t* line, const char* tag, void** dst, int* ndst, int type);
#define bcf_get_format_int32(hdr, line, tag, dst, ndst) bcf_get_format_values(hdr, line, tag, (void**)(dst), ndst, BCF_HT_INT)
#define bcf_get_genotypes(hdr, line, dst, ndst) bcf_get_format_values(hdr, line, "GT", (void**)(dst), ndst, BCF_HT_INT)
int bcf_update_format(const bcf_hdr_t* hdr, bcf1_t* line, const char* key, const void* values, int n, int type);
#define bcf_update_format_int32(hdr, line, key, values, n) bcf_update_format((hdr), (line), (key), (values), (n), BCF_HT_INT)
#define bcf_update_genotypes(hdr, line, gts, n) bcf_update_format((hdr), (line), "GT", (gts), (n), BCF_HT_INT)
int bcf_update_info(const bcf_hdr_t* hdr, bcf1_t* line, const char* key, const void* values, int n, int type);
#define bcf_update_info_int32(hdr, line, key, values, n) bcf_update_info((hdr), (line), (key), (values), (n), BCF_HT_INT)
#ifdef __cplusplus
}
#endif
#endif
