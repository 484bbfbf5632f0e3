// Host-side code of the library under AddressSanitizer + UndefinedBehaviorSanitizer (the reference builds its own
// sanitizer variant with ADD_EXTRA=y, /root/reference/Makefile:7-10).  Built by `make -C xsqueezeit_amd/csrc asan-host`
// from csrc/xsi_pack.cpp (the writer's pack-on-append: AVX-512 / AVX2 / scalar by the CPU and XSI_PACK_ISA) and run by
// tests/test_host.py::test_host_packer_under_sanitizers once per instruction set.  Test infrastructure, not product.
//
// Every buffer is exactly as long as the contract says (gt: n values, out: ceil(n / 8) bytes, both straight from
// malloc so that ASan's red zones sit right behind them): a vector tail that reads or writes one element too far aborts.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace xsi {
bool pack_bit_row(const int32_t* gt, uint32_t n, int dp, uint8_t* out);
}

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t next_u64() {  // splitmix64
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static int check(uint32_t n, int dp) {
    int32_t* gt = static_cast<int32_t*>(malloc(sizeof(int32_t) * n));
    const uint32_t nb = (n + 7u) / 8u;
    uint8_t* out = static_cast<uint8_t*>(malloc(nb));
    uint8_t* want = static_cast<uint8_t*>(calloc(nb, 1));
    for (uint32_t i = 0; i < n; ++i) {
        const int al = (next_u64() & 7u) < 3u;
        const int ph = (i & 1u) ? dp : (int)(next_u64() & 1u);  // the first value's phase bit is not stored by the format
        gt[i] = ((al + 1) << 1) | ph;
        want[i >> 3] |= (uint8_t)(al << (i & 7u));
    }
    memset(out, 0xEE, nb);
    int bad = 0;
    if (!xsi::pack_bit_row(gt, n, dp, out)) {
        fprintf(stderr, "n=%u dp=%d: a packable row was refused\n", n, dp);
        bad = 1;
    } else if (memcmp(out, want, nb) != 0) {
        fprintf(stderr, "n=%u dp=%d: wrong bits\n", n, dp);
        bad = 1;
    }
    // rows the bit form cannot hold: refused wherever the offending value sits (head, vector body, tail)
    const uint32_t spots[5] = {0u, n / 2u, n - 1u, n >= 2u ? n - 2u : 0u, n / 3u};
    const int32_t offenders[5] = {6, 0, INT32_MIN, INT32_MIN + 1, 1};
    for (int k = 0; k < 5 && !bad; ++k) {
        const int32_t keep = gt[spots[k]];
        gt[spots[k]] = offenders[k];
        if (xsi::pack_bit_row(gt, n, dp, out)) {
            fprintf(stderr, "n=%u dp=%d: value %d at %u was packed\n", n, dp, (int)offenders[k], spots[k]);
            bad = 1;
        }
        gt[spots[k]] = keep;
    }
    if (n >= 2u && !bad) {  // a second value with the other phase
        const uint32_t pos = (n - 1u) | 1u;
        if (pos < n) {
            gt[pos] ^= 1;
            if (xsi::pack_bit_row(gt, n, dp, out)) {
                fprintf(stderr, "n=%u dp=%d: a non-default phase at %u was packed\n", n, dp, pos);
                bad = 1;
            }
        }
    }
    free(gt);
    free(out);
    free(want);
    return bad;
}

int main() {
    int bad = 0;
    for (uint32_t n = 1; n <= 1100 && !bad; ++n)
        for (int dp = 0; dp < 2; ++dp) bad |= check(n, dp);
    const uint32_t big[] = {5008u, 64976u, 65535u, 65536u + 17u, 200000u, 500000u};
    for (uint32_t n : big)
        for (int dp = 0; dp < 2 && !bad; ++dp) bad |= check(n, dp);
    if (!bad) printf("pack_asan ok\n");
    return bad;
}
