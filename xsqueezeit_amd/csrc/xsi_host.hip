// xsi_host.hip — file-level writer / accessor (placeholder until the general path lands).
#include "../../include/xsi_hip.h"
#include "xsi_ctx.hpp"

using namespace xsi;

extern "C" {
int xsi_hip_encode_gt(xsi_hip_ctx*, const xsi_encode_params*, const int32_t*, uint64_t, uint64_t, const uint32_t*,
                      const uint32_t*, void*, uint64_t, uint64_t*, xsi_encode_result*) {
    return set_error(XSI_ERR_UNSUPPORTED, "encode_gt: not built yet");
}
int xsi_hip_decode_gt(xsi_hip_ctx*, const void*, uint64_t, uint64_t, uint64_t, const uint32_t*, uint64_t, int32_t*,
                      uint64_t, uint32_t*, uint64_t*, uint32_t) {
    return set_error(XSI_ERR_UNSUPPORTED, "decode_gt: not built yet");
}
int xsi_writer_open(xsi_writer**, xsi_hip_ctx*, const char*, const xsi_encode_params*, const char* const*) {
    return set_error(XSI_ERR_UNSUPPORTED, "writer: not built yet");
}
int xsi_writer_append(xsi_writer*, const int32_t*, uint32_t, uint32_t) { return XSI_ERR_UNSUPPORTED; }
int xsi_writer_finalize(xsi_writer*, uint32_t) { return XSI_ERR_UNSUPPORTED; }
void xsi_writer_close(xsi_writer*) {}
int xsi_accessor_open(xsi_accessor**, xsi_hip_ctx*, const char*) {
    return set_error(XSI_ERR_UNSUPPORTED, "accessor: not built yet");
}
int64_t xsi_accessor_fill_genotype_array(xsi_accessor*, int32_t*, uint64_t, uint32_t, uint64_t) { return XSI_ERR_UNSUPPORTED; }
int64_t xsi_accessor_get_genotypes(xsi_accessor*, uint32_t, uint64_t, void**, int*) { return XSI_ERR_UNSUPPORTED; }
int xsi_accessor_allele_counts(xsi_accessor*, uint64_t*, uint32_t) { return XSI_ERR_UNSUPPORTED; }
uint64_t xsi_accessor_hap_samples(const xsi_accessor*) { return 0; }
uint64_t xsi_accessor_num_samples(const xsi_accessor*) { return 0; }
const char* xsi_accessor_sample_name(const xsi_accessor*, uint64_t) { return nullptr; }
void xsi_accessor_close(xsi_accessor*) {}
}
