#!/usr/bin/env python3
"""Randomised parity sweep, GPU path against the CPU oracle (test infrastructure, like tests/): random shapes of the
packed path (haplotypes 2 .. 140 000, block lengths, MAC thresholds) and of the general int32 path (multi-allelic,
missing, end-of-vector, phase, haploid lines).  A fixed-seed subset of it runs in the `-m gpu` suite (tests/test_gpu_stress.py: 42 cases of every kind through run_case below); the
long sweeps are run by hand (minutes, not seconds):
    gpurun -- python3 tests/stress_parity.py --seed 1 --cases 60
Prints one line per case and exits non-zero on the first mismatch."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def run_case(kind, rng, c=0, cells=24_000_000, max_lines=400, long_rows=False, tmpdir=None, t0=None):
    """One random case of `kind` ("packed", "general" or "file") drawn from `rng`: GPU bytes against the oracle's file,
    GPU decode against the source.  Returns (ok_encode_or_write, ok_decode_or_read, one-line description)."""
    import ctypes
    import gpu_util as G
    from oracle import oracle
    from test_oracle import _random_lines
    from xsqueezeit_amd import binding, synth
    L = binding.lib()
    t0 = time.time() if t0 is None else t0
    if kind == "file":
        # xsi_writer_* / xsi_accessor_*: many small blocks (several writer batches), random read order
        n = int(rng.choice([3, 20, 120, 700, 2504]))
        n_lines = int(min(max(2, 3_000_000 // (2 * n)), rng.integers(2, 2500)))
        block_len = int(rng.choice([1, 2, 5, 16, 100, 8192]))
        kw = dict(multi=bool(rng.integers(0, 2)), missing=bool(rng.integers(0, 2)), eov=bool(rng.integers(0, 2)),
                  phase=bool(rng.integers(0, 2)))
        lines = _random_lines(rng, n, n_lines, **kw)
        dp = oracle.default_phased_of(lines, n)
        names = ["s%d" % i for i in range(n)]
        mac = int(rng.choice([0, 1, 3]))
        ref = oracle.encode_file(lines, n, block_len=block_len, mac_thr=mac, default_phased=dp, sample_names=names)
        path = os.path.join(tmpdir, "f%d.xsi" % c).encode()
        p = G.params(n, block_len, mac, dp)
        w = ctypes.c_void_p()
        arr = (ctypes.c_char_p * n)(*[x.encode() for x in names])
        binding.check(L.xsi_writer_open(ctypes.byref(w), G.ctx().handle, path, ctypes.byref(p), arr))
        for gt, na in lines:
            gt = np.ascontiguousarray(gt, dtype=np.int32)
            binding.check(L.xsi_writer_append(w, gt.ctypes.data, gt.size, na))
        binding.check(L.xsi_writer_finalize(w, 0))
        L.xsi_writer_close(w)
        got = open(path, "rb").read()
        ok = got == ref
        a = ctypes.c_void_p()
        binding.check(L.xsi_accessor_open(ctypes.byref(a), G.ctx().handle, path))
        bms = []
        block = off = 0
        for i, (_, na) in enumerate(lines):
            if i and i % block_len == 0:
                block += 1
                off = 0
            bms.append((block << 15) | off)
            off += na - 1
        buf = np.zeros(2 * n, dtype=np.int32)
        order = [int(x) for x in rng.permutation(n_lines)[:400]] + list(range(min(n_lines, 200)))
        ok2 = True
        for i in order:
            na = lines[i][1]
            r = L.xsi_accessor_fill_genotype_array(a, buf.ctypes.data, buf.size, na, bms[i])
            if r != len(lines[i][0]) or not np.array_equal(buf[:r], lines[i][0]):
                ok2 = False
                break
        L.xsi_accessor_close(a)
        os.remove(path)
        return ok, ok2, ("%3d file    samples=%6d lines=%4d block=%4d thr=%4d %s bytes=%8d  write %s read %s  (%.0f s)"
                         % (c, n, n_lines, block_len, mac, "".join(k[0] for k, v in kw.items() if v) or "-", len(got),
                            "ok" if ok else "MISMATCH", "ok" if ok2 else "MISMATCH", time.time() - t0))
    if kind == "packed":
        # sizes around every kernel boundary: 12 288, 16 384, 20 000, 49 152, 65 536, 131 072
        base = int(rng.choice([2, 64, 130, 1000, 5008, 12288, 16384, 20000, 32768, 49152, 65534, 65536, 70002, 131072, 140000]))
        n_haps = max(2, base + int(rng.integers(-70, 70))) & ~1
        if 32768 * 2 <= n_haps <= 65535 * 2:   # the reference's A_T mismatch window is refused by design
            n_haps = 131072 + 2 * int(rng.integers(0, 3000))
        cells_budget = cells
        if long_rows:
            n_haps = (int(rng.choice([140000, 196608, 200000, 262144, 330000, 400002, 500000, 524288])) - 2 * int(rng.integers(0, 40))) & ~1
            cells_budget = max(cells_budget, 30_000_000)
        n_lines = int(min(max(1, cells_budget // n_haps), rng.integers(1, max_lines)))
        block_len = int(rng.choice([1, 3, 8, 16, 64, 100, 8192]))
        thr = int(rng.choice([0, 1, n_haps // 1000, n_haps // 100, n_haps // 10]))
        dens = float(rng.choice([0.0005, 0.01, 0.1, 0.5, 0.9]))
        if long_rows:
            block_len = int(rng.choice([max(1, n_lines // 2 + 1), 7, 8192]))
            thr = int(rng.choice([0, n_haps // 1000]))
            dens = rng.choice([0.002, 0.01, 0.05, 0.12, 0.3, 0.5, 0.8, 0.97, 0.999], size=(n_lines, 1))
        bits = (rng.random((n_lines, n_haps), dtype=np.float32) < dens).astype(np.uint8)
        # runs: copy founders so that PBWT produces long fills
        if rng.random() < 0.5 and n_lines > 4:
            f = rng.integers(0, 8, size=n_haps)
            fb = (rng.random((n_lines, 8)) < dens).astype(np.uint8)  # (dens: a scalar, or one value per line)
            bits = fb[:, f] ^ (rng.random((n_lines, n_haps)) < 0.002).astype(np.uint8)
        stride = synth.row_stride_bytes(n_haps)
        packed = synth.pack_rows(bits, stride)
        p = G.params(n_haps // 2, block_len, thr)
        names = ["S%d" % i for i in range(n_haps // 2)]
        ref = G.oracle_file_from_bits(bits, p, names)
        region, offsets, res = G.encode_packed(packed, n_haps, p)
        got = G.assemble_file(region, offsets, p, n_lines, n_lines, names)
        ok = got == ref
        out, counts = G.decode_packed(got, n_haps, stride)
        ok2 = bool(np.array_equal(out, packed) and np.array_equal(counts, bits.sum(1).astype(np.int32)))
        return ok, ok2, ("%3d packed  haps=%6d lines=%4d block=%4d thr=%5d dens=%.4f bytes=%8d  encode %s decode %s  (%.0f s)"
                         % (c, n_haps, n_lines, block_len, thr, float(np.mean(dens)), len(got), "ok" if ok else "MISMATCH",
                            "ok" if ok2 else "MISMATCH", time.time() - t0))
    n = int(rng.choice([3, 37, 333, 2504, 6000, 9000, 20000]))
    n_lines = int(min(max(2, 6_000_000 // (2 * n)), rng.integers(2, 300)))
    block_len = int(rng.choice([1, 4, 32, 100, 8192]))
    kw = dict(multi=bool(rng.integers(0, 2)), missing=bool(rng.integers(0, 2)), eov=bool(rng.integers(0, 2)),
              phase=bool(rng.integers(0, 2)))
    lines = _random_lines(rng, n, n_lines, **kw)
    # fully haploid lines only without multi-allelic ones in the block: the reference writes KEY_LINE_HAPLOID per BCF
    # line and reads it per binary line, such blocks do not decode back to their input anywhere
    # (tests/test_gpu_stress.py::test_haploid_flags_misaligned_by_multiallelic_lines pins that case by itself)
    if not kw["multi"] and rng.random() < 0.6:
        for i in range(0, n_lines, 5):
            al = (rng.random(n) < 0.3).astype(np.int32)
            lines[i] = (((al + 1) << 1).astype(np.int32), 2)
    dp = oracle.default_phased_of(lines, n)
    thr = int(rng.choice([0, 1, max(1, 2 * n // 100)]))
    p = G.params(n, block_len, thr, dp)
    ref = oracle.encode_file(lines, n, block_len=block_len, mac_thr=thr, default_phased=dp)
    region, offsets, res = G.encode_gt(lines, n, p)
    names = ["S%d" % i for i in range(n)]
    got = G.assemble_file(region, offsets, p, n_lines, G.num_variants(lines), names, 2)
    ok = got == ref
    nal = [na for _, na in lines]
    rows, counts = G.decode_gt(got, nal)
    ok2 = all(np.array_equal(rows[i][:len(lines[i][0])], lines[i][0]) for i in range(n_lines))
    # the same lines as a WS_PBWT_WAH file (version-4 missing-data strategy, written by the oracle): decode only;
    # not with fully haploid lines (refused)
    if ok2 and (kw["missing"] or kw["eov"]) and all(len(g) == 2 * n for g, _ in lines):
        ref_pw = oracle.encode_file(lines, n, block_len=block_len, mac_thr=thr, default_phased=dp, wah_encode_missing=2)
        rows_pw, _ = G.decode_gt(ref_pw, nal)
        ok2 = all(np.array_equal(rows_pw[i][:len(lines[i][0])], lines[i][0]) for i in range(n_lines))
    return ok, ok2, ("%3d general samples=%6d lines=%4d block=%4d thr=%4d %s bytes=%8d  encode %s decode %s  (%.0f s)"
                     % (c, n, n_lines, block_len, thr, "".join(k[0] for k, v in kw.items() if v) or "-", len(got),
                        "ok" if ok else "MISMATCH", "ok" if ok2 else "MISMATCH", time.time() - t0))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--cells", type=int, default=24_000_000, help="packed cases: haplotypes x lines at most")
    ap.add_argument("--max-lines", type=int, default=400)
    ap.add_argument("--files", action="store_true", help="every fourth case through xsi_writer_* / xsi_accessor_*")
    ap.add_argument("--long-rows", action="store_true",
                    help="packed cases at 140 000 .. 524 288 haplotypes with a density of its own per line (sparse, dense and "
                         "mostly-ones lines next to each other: both exchange forms of k_chain_rank_enc_multi and the "
                         "long-row decode, 3 to 8 workgroups per block)")
    args = ap.parse_args()
    import tempfile
    rng = np.random.default_rng(args.seed)
    t0 = time.time()
    tmpdir = tempfile.mkdtemp(prefix="xsi_stress_")
    for c in range(args.cases):
        kind = "packed" if c % 3 else "general"
        if args.files and c % 4 == 1:
            kind = "file"
        ok, ok2, line = run_case(kind, rng, c, args.cells, args.max_lines, args.long_rows, tmpdir, t0)
        print(line, flush=True)
        if not (ok and ok2):
            sys.exit(1)
    print("all %d cases ok" % args.cases)


if __name__ == "__main__":
    main()
