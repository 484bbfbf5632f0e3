"""Synthetic haplotype matrices — numpy mirror of k_synth_packed.

Integer-only and fully specified so host and device agree bit for bit:

    mix64        = splitmix64 finaliser
    site hash    hs = mix64(seed*0x9E3779B97F4A7C15 + site + 1)
    allele count k  = 2^e + (hs>>16 & (2^e-1)), e = (hs & 0xFFFF) mod bitlen(N-1), clipped to N-1
                      (octave-uniform, i.e. P(k) ~ 1/k: the neutral site-frequency spectrum)
    threshold    t  = floor(k * 2^32 / N)
    rare sites (k*256 < 4N): bit(h) = hi32(mix64(hs ^ (h*C1 + 1))) < t          (no LD)
    common sites: haplotype h copies founder g = mix64(seed + h*C0 + seg*C2) mod 256 over
                  segments of 4096 sites (per-haplotype phase), bit = hi32(mix64(hs ^ (g*C1 + 0x51ED))) < t,
                  flipped with probability 2^-12 (hi32(mix64(hs ^ (h*C1 + 0xABCD))) < 2^20)
"""
import numpy as np

M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
C0 = np.uint64(0x9E3779B97F4A7C15)
C1 = np.uint64(0xD1B54A32D192ED03)
C2 = np.uint64(0xC2B2AE3D27D4EB4F)
FOUNDERS = 256
SEG = 4096
MUT = 1 << 20


def mix64(z):
    z = np.asarray(z, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def synth_bits(seed, first_line, n_lines, n_haps):
    """uint8 matrix [n_lines, n_haps] of 0/1."""
    seed = np.uint64(seed)
    h = np.arange(n_haps, dtype=np.uint64)
    nb = max(1, int(n_haps - 1).bit_length())
    out = np.zeros((n_lines, n_haps), dtype=np.uint8)
    with np.errstate(over="ignore"):
        off = mix64(seed ^ (h * C1 + np.uint64(7))) % np.uint64(SEG)
        for r in range(n_lines):
            site = np.uint64(first_line + r)
            hs = mix64(seed * C0 + site + np.uint64(1))
            e = int(int(hs) & 0xFFFF) % nb
            k = (1 << e) + ((int(hs) >> 16) & ((1 << e) - 1))
            k = min(k, n_haps - 1)
            t32 = (k << 32) // n_haps
            if k * FOUNDERS < 4 * n_haps:
                rr = mix64(hs ^ (h * C1 + np.uint64(1)))
                bit = (rr >> np.uint64(32)) < np.uint64(t32)
            else:
                seg = (site + off) // np.uint64(SEG)
                g = mix64(seed + h * C0 + seg * C2) % np.uint64(FOUNDERS)
                rr = mix64(hs ^ (g * C1 + np.uint64(0x51ED)))
                bit = (rr >> np.uint64(32)) < np.uint64(t32)
                r2 = mix64(hs ^ (h * C1 + np.uint64(0xABCD)))
                bit = bit ^ ((r2 >> np.uint64(32)) < np.uint64(MUT))
            out[r] = bit
    return out


def pack_rows(bits01, row_stride_bytes):
    """[n_lines, n_haps] 0/1 -> uint8 [n_lines, row_stride_bytes], LSB-first in little-endian words."""
    n_lines, n_haps = bits01.shape
    packed = np.packbits(bits01, axis=1, bitorder="little")
    out = np.zeros((n_lines, row_stride_bytes), dtype=np.uint8)
    out[:, :packed.shape[1]] = packed
    return out


def unpack_rows(packed, n_haps):
    return np.unpackbits(packed, axis=1, bitorder="little")[:, :n_haps]


def row_stride_bytes(n_haps, align=128):
    """Rows padded to 128 B (coalesced 128-byte reads)."""
    return ((n_haps + 7) // 8 + align - 1) // align * align


def bits_to_gt(bits01, default_phased=1):
    """0/1 haplotype matrix -> htslib int32 rows for diploid bi-allelic fully called lines."""
    gt = ((bits01.astype(np.int32) + 1) << 1)
    if default_phased:
        gt[:, 1::2] |= 1
    return gt
