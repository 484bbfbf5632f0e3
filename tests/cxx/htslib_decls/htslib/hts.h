/* TEST INFRASTRUCTURE, NOT htslib.  Declaration-only prototypes of the few htslib (public API, hts.h) names that
 * xsqueezeit_amd/csrc/xsi_htslib_shim.cpp uses, hand-written from the API's documented signatures so that the shim
 * can be compiled with -fsyntax-only in an image that has no htslib (tests/test_host.py).  Never shipped, never
 * linked, no bodies; a build that means it uses the real <htslib/...> headers (make HTSLIB=1). */
#ifndef XSI_TEST_HTSLIB_DECLS_HTS_H
#define XSI_TEST_HTSLIB_DECLS_HTS_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef int64_t hts_pos_t;
typedef struct htsFile htsFile;
htsFile* hts_open(const char* fn, const char* mode);
int hts_close(htsFile* fp);
#ifdef __cplusplus
}
#endif
#endif
