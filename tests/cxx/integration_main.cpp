// Drives the two classes INTEGRATION.md shows (extracted from the markdown by the test into
// accessor_internals_hip.hpp / xsi_factory_hip.hpp) through their reference-side interfaces: writes a file
// with XsiFactoryInterface::append / finalize_file, reads it with AccessorInternals::fill_genotype_array.
//   usage: integration_main <out.xsi>     prints "ok" and exits 0 when every line round-trips
#include <cstdio>
#include <memory>

#include "accessor_internals_hip.hpp"
#include "xsi_factory_hip.hpp"

GlobalAppOptions global_app_options;

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    const size_t n_samples = 300, n_lines = 700, block_len = 256;
    std::vector<std::string> names;
    for (size_t i = 0; i < n_samples; ++i) names.push_back("S" + std::to_string(i));
    std::vector<std::vector<int>> rows(n_lines, std::vector<int>(2 * n_samples));
    for (size_t l = 0; l < n_lines; ++l)
        for (size_t h = 0; h < 2 * n_samples; ++h) {
            const unsigned v = (unsigned)(l * 2654435761u) ^ (unsigned)(h * 40503u);
            const int alt = ((v >> 7) % 100u) < (l % 37u);
            rows[l][h] = ((alt + 1) << 1) | (int)(h & 1);
        }
    try {
        std::unique_ptr<XsiFactoryInterface> f(new XsiFactoryHip(argv[1], block_len, 1, 1, names, false, 0));
        bcf1_t rec;
        bcf_file_reader_info_t fri;
        fri.n_samples = n_samples;
        fri.line = &rec;
        for (size_t l = 0; l < n_lines; ++l) {
            fri.gt_arr = rows[l].data();
            fri.ngt = (int)(2 * n_samples);
            f->append(fri);
        }
        f->finalize_file(2);
        f.reset();
        std::unique_ptr<AccessorInternals> a(new AccessorInternalsHip(argv[1]));
        std::vector<int32_t> gt(2 * n_samples);
        for (size_t l = 0; l < n_lines; ++l) {
            const size_t pos = ((l / block_len) << 15) | (l % block_len);
            if (a->fill_genotype_array(gt.data(), gt.size(), 2, pos) != 2 * n_samples) return 3;
            for (size_t h = 0; h < 2 * n_samples; ++h)
                if (gt[h] != rows[l][h]) return 4;
            size_t alt = 0;
            for (int v : rows[l]) alt += ((v >> 1) - 1) == 1;
            if (a->get_allele_counts()[1] != alt) return 5;
        }
        // InternalGtAccess of a record in the second block: one line, a pointer into the host image, `a` a permutation
        {
            const size_t l = block_len + block_len / 2;
            InternalGtAccess ia = a->get_internal_access(2, ((l / block_len) << 15) | (l % block_len));
            if (ia.n_alleles != 2 || ia.pointers.size() != 1 || ia.sparse.size() != 1 || !ia.pointers[0] || !ia.a) return 7;
            std::vector<char> seen(2 * n_samples, 0);
            const uint32_t* arr = (const uint32_t*)ia.a;
            for (size_t i = 0; i < 2 * n_samples; ++i) {
                if (arr[i] >= 2 * n_samples || seen[arr[i]]) return 8;
                seen[arr[i]] = 1;
            }
        }
    } catch (const char* e) {
        std::fprintf(stderr, "exception: %s\n", e);
        return 6;
    }
    std::puts("ok");
    return 0;
}
