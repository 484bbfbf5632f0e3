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


# ---------------- BASELINE configs[4] content (SURVEY.md 8d, config 5 of its table) ----------------
INT32_VECTOR_END = -2147483647  # bcf_int32_vector_end (0x80000001)


def config4_n_allele(first_line, n_lines):
    """Alleles of each BCF line: every tenth site (site % 10 == 3) is tri-allelic."""
    site = np.arange(first_line, first_line + n_lines, dtype=np.int64)
    return np.where(site % 10 == 3, 3, 2).astype(np.uint32)


def config4_rows_device(L, ctx, torch, dev, seed, first_line, n_lines, n_haps, default_phased=1, chunk=1024):
    """htslib int32 genotype rows [n_lines, n_haps] in HBM for the mixed-ploidy / multi-allelic workload:
    ALT 1 carriers from the synthetic matrix of `seed`, on tri-allelic sites ALT 2 carriers (among the REF
    haplotypes) from the matrix of seed + 1000; 5 % of the samples (sample % 20 == 7) are "male": their second
    value is end-of-vector.  No fully haploid lines.  Second values carry `default_phased`.
    Returns (rows int32 tensor, n_allele uint32 numpy)."""
    import ctypes
    from . import binding
    stride = row_stride_bytes(n_haps)
    nal = config4_n_allele(first_line, n_lines)
    rows = torch.empty((n_lines, n_haps), dtype=torch.int32, device=dev)
    shifts = torch.arange(8, dtype=torch.uint8, device=dev)
    male = (torch.arange(n_haps // 2, device=dev) % 20) == 7
    for r0 in range(0, n_lines, chunk):
        n = min(chunk, n_lines - r0)
        planes = []
        for sd in (seed, seed + 1000):
            pk = torch.empty(n * stride, dtype=torch.uint8, device=dev)
            binding.check(L.xsi_hip_synth_packed(ctx.handle, sd, first_line + r0, n, n_haps, pk.data_ptr(), stride))
            planes.append(((pk.view(n, stride, 1) >> shifts) & 1).view(n, stride * 8)[:, :n_haps])
        tri = torch.from_numpy((nal[r0:r0 + n] == 3)).to(dev)
        allele = planes[0].to(torch.int32)
        allele = torch.where(tri[:, None] & (planes[0] == 0) & (planes[1] == 1), torch.full_like(allele, 2), allele)
        gt = (allele + 1) << 1
        gt[:, 1::2] |= int(default_phased)
        gt[:, 1::2] = torch.where(male[None, :], torch.full_like(gt[:, 1::2], INT32_VECTOR_END), gt[:, 1::2])
        rows[r0:r0 + n] = gt
        del planes, allele, gt, pk
    return rows, nal


def bm_positions(n_allele, block_len):
    """BM value of every BCF line (block << 15 | binary-line offset inside the block, xcf.cpp:685-703)."""
    nal = np.asarray(n_allele, dtype=np.int64)
    n = len(nal)
    bm = np.empty(n, dtype=np.int64)
    for b0 in range(0, n, block_len):
        k = nal[b0:b0 + block_len] - 1
        off = np.concatenate(([0], np.cumsum(k)[:-1]))
        bm[b0:b0 + block_len] = ((b0 // block_len) << 15) | off
    return bm
