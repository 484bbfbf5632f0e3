// Minimal stand-in for what the INTEGRATION.md encode snippet uses of the reference's
// include/xsi_factory.hpp (XsiFactoryInterface, :38-46), include/xcf.hpp (bcf_file_reader_info_t, :51-62,
// only the fields GtBlock::encode_line reads) and include/xsqueezeit.hpp (global_app_options), written for
// this test only.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

struct bcf1_t {
    uint32_t n_allele = 2;
};
struct bcf_file_reader_info_t {
    size_t n_samples = 0;
    int* gt_arr = nullptr;
    int ngt = 0;
    bcf1_t* line = nullptr;
};
struct GlobalAppOptions {
    bool wah_encode_missing = false;
};
extern GlobalAppOptions global_app_options;

class XsiFactoryInterface {
public:
    virtual void append(const bcf_file_reader_info_t& bcf_fri) = 0;
    virtual void finalize_file(const size_t max_ploidy = 2) = 0;
    virtual ~XsiFactoryInterface() {}
};
