/* TEST INFRASTRUCTURE, NOT htslib: see hts.h in this directory.  Prototypes of the synced_bcf_reader.h names the shim
 * uses. */
#ifndef XSI_TEST_HTSLIB_DECLS_SYNCED_BCF_READER_H
#define XSI_TEST_HTSLIB_DECLS_SYNCED_BCF_READER_H
#include "hts.h"
#include "vcf.h"
#ifdef __cplusplus
extern "C" {
#endif
typedef struct bcf_sr_t {
    htsFile* file;
    const char* fname;
    bcf_hdr_t* header;
    bcf1_t** buffer;
    int nbuffer, mbuffer;
} bcf_sr_t;

typedef struct bcf_srs_t {
    int collapse;
    char* apply_filters;
    int require_index;
    int max_unpack;
    int* has_line;
    int errnum;
    bcf_sr_t* readers;
    int nreaders;
} bcf_srs_t;

bcf_srs_t* bcf_sr_init(void);
void bcf_sr_destroy(bcf_srs_t* readers);
int bcf_sr_add_reader(bcf_srs_t* readers, const char* fname);
int bcf_sr_next_line(bcf_srs_t* readers);
int bcf_sr_set_regions(bcf_srs_t* readers, const char* regions, int is_file);
int bcf_sr_set_targets(bcf_srs_t* readers, const char* targets, int is_file, int alleles);
#define bcf_sr_get_line(_readers, i) ((_readers)->has_line[i] ? ((_readers)->readers[i].buffer[0]) : (bcf1_t*)NULL)
#ifdef __cplusplus
}
#endif
#endif
