// xsi_common.hpp — structures shared between the host API and the gfx950 kernels.
#pragma once

#include <stdint.h>
#include <stdlib.h>

namespace xsi {

// Environment switches.  The library reads NO tuning or testing variable unless the process opted in: a stray XSI_*
// variable in the environment of an HTSLIB caller must not change kernels, thresholds or inject failures.
//   XSI_ENABLE_TUNING_ENV=1   makes the variables of README.md "Environment switches" effective (dispatch thresholds,
//                             kernel variants for A/B runs and parity tests, profiling prints)
//   XSI_ENABLE_TEST_HOOKS=1   makes the three failure-injection hooks effective (tests only)
// Both are looked up on every read, like the variables themselves (tests switch them inside one process).
inline const char* tuning_env(const char* name) {
    const char* on = ::getenv("XSI_ENABLE_TUNING_ENV");
    return (on && on[0] == '1') ? ::getenv(name) : nullptr;
}
inline const char* test_hook_env(const char* name) {
    const char* on = ::getenv("XSI_ENABLE_TEST_HOOKS");
    return (on && on[0] == '1') ? ::getenv(name) : nullptr;
}

// GT block dictionary keys (gt_block.hpp:34-60)
enum : uint32_t {
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
constexpr uint32_t VAL_UNDEFINED = 0xFFFFFFFFu;
constexpr uint32_t WS_PBWT_WAH = 0, WS_WAH = 1, WS_SPARSE = 2;
constexpr uint32_t KEY_GT_ENTRY = 256;  // interfaces.hpp:167
constexpr uint32_t BM_BLOCK_BITS = 15;  // accessor_internals.hpp:412
constexpr uint32_t MAX_BIN_PER_BLOCK = 1u << BM_BLOCK_BITS;

// A_T mismatch window of the reference: for 32768 <= n_samples <= 65535 the file
// header says 4-byte A_T (2*n_samples > 65535, gt_compressor_new.hpp:177-187) while the block encoder
// is instantiated with uint16_t (n_samples <= 65535, xsi_factory.hpp:424-427), whose prefix array
// `std::vector<uint16_t> a(2*n_samples)` wraps modulo 65536 (gt_block.hpp:171,179).  The reference
// cannot decode what it writes there, so every encode entry point refuses the window.
inline bool at_mismatch_window(uint64_t n_samples) { return n_samples >= 32768u && n_samples <= 65535u; }

// per-binary-line kind bits
constexpr uint32_t KIND_WAH = 1u;       // WAH + PBWT line (else sparse)
constexpr uint32_t KIND_NEGATED = 2u;   // sparse line lists the REF positions (MSB of the count set)
constexpr uint32_t KIND_HAPLOID = 4u;   // fully haploid BCF line (ngt == n_samples)

// flag vectors a block can carry, in the order they are written (gt_block.hpp:512-647)
enum : uint32_t { FV_IS_WAH = 0, FV_MISSING = 1, FV_EOV = 2, FV_PHASE = 3, FV_HAPLOID = 4, FV_COUNT = 5 };
constexpr uint32_t FLAG_WORDS_MAX = MAX_BIN_PER_BLOCK / 15 + 2;  // WAH16 words of a 32768-bit vector

// Everything the kernels need to know about one block of an encode batch.
struct EncBlock {
    // host-provided
    uint32_t first_bcf, n_bcf;  // BCF lines [first_bcf, first_bcf+n_bcf)
    uint32_t first_bin, n_bin;  // binary lines (one per ALT allele)
    // device-computed
    uint32_t wah_first, n_wah;        // ranks of this block's WAH lines in the batch-wide list
    uint32_t sparse_bytes;            // bytes of the sparse matrix
    uint32_t wah_words;               // words of the WAH matrix
    uint32_t has_missing, has_eov, has_phase, has_haploid, max_ploidy;
    uint32_t miss_bytes, eov_bytes, phase_words;  // side-channel matrix sizes
    uint32_t flag_len[FV_COUNT];      // words of each encoded flag vector
    uint32_t dict_idx, n_keys;
    uint32_t off_flag[FV_COUNT];      // offsets relative to the GT block start
    uint32_t off_wah, off_sparse, off_miss, off_eov, off_phase;
    uint32_t gt_bytes;                // GT block bytes
    uint32_t block_bytes;             // 16 + gt_bytes, padded to 4
    uint64_t out_off;                 // offset of the block in the blocks region
};

// One block of a decode batch, parsed from the file image.
struct DecBlock {
    uint64_t file_off;      // of the outer block
    uint64_t gt_off;        // of the GT block
    uint32_t n_bcf, n_bin;
    uint32_t max_ploidy, default_phasing, strategy;
    uint32_t off_select, off_wah, off_sparse;
    uint32_t off_line_missing, off_miss_wah, off_miss_sparse;
    uint32_t off_line_eov, off_eov_wah, off_eov_sparse;
    uint32_t off_line_phase, off_phase, off_line_haploid;
    uint32_t wah_words;     // words in the WAH matrix (off_sparse - off_wah)/2
    uint32_t first_bin;     // binary lines of earlier blocks in the batch
    uint32_t first_bcf;
    uint32_t wah_first, n_wah;      // batch-wide rank of this block's WAH lines
    uint32_t sparse_first, n_sparse;
    uint32_t error;
};

}  // namespace xsi
