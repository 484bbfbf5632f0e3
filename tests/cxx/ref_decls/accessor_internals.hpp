// Minimal stand-in for the declarations of the reference's include/accessor_internals.hpp that the
// INTEGRATION.md decode snippet touches, written for this test only (not the reference's header): the
// abstract AccessorInternals interface (accessor_internals.hpp:399-413) and InternalGtAccess (:374-397).
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

struct InternalGtAccess {
    size_t position = 0, n_alleles = 0, sparse_bytes = 0, wah_bytes = 0, a_bytes = 0;
    int32_t default_allele = 0;
    const void* a = nullptr;
    std::vector<bool> sparse;
    std::vector<void*> pointers;
};

class AccessorInternals {
public:
    virtual ~AccessorInternals() {}
    virtual size_t fill_genotype_array(int32_t* gt_arr, size_t gt_arr_size, size_t n_alleles, size_t position) = 0;
    virtual void fill_allele_counts(size_t n_alleles, size_t position) = 0;
    virtual const std::vector<size_t>& get_allele_counts() const { return allele_counts; }
    virtual InternalGtAccess get_internal_access(size_t n_alleles, size_t position) = 0;

protected:
    std::vector<size_t> allele_counts;
};
