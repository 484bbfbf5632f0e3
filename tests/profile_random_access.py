#!/usr/bin/env python3
"""Random-access decode through the Accessor boundary (BASELINE.json configs[4] style, single GPU).

Builds a synthetic bi-allelic .xsi (same generator as bench.py), opens it with xsi_accessor_* and
times Accessor::fill_genotype_array (accessor.hpp:48-50) for
  * uniformly random BM positions, cold (every first touch of a block decodes it on the GPU),
  * the same number of random positions again, warm (decoded blocks resident in HBM),
  * contiguous windows of consecutive lines,
and, beside it, the CPU oracle's reader on a few of the same random positions (it replays the
block prefix on every seek, accessor_internals_new.hpp:154-196).  Every returned row is checked
against the input matrix.  Prints one JSON line.  Not part of bench.py's contract: a profile tool.
"""
import argparse
import ctypes
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # repo root (this script lives in tests/: it times the CPU oracle beside the GPU)
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--haps", type=int, default=200000)
    ap.add_argument("--sites", type=int, default=32 * 8192)
    ap.add_argument("--block-len", type=int, default=8192)
    ap.add_argument("--maf", type=float, default=0.001)
    ap.add_argument("--queries", type=int, default=2000)
    ap.add_argument("--windows", type=int, default=5)
    ap.add_argument("--window-len", type=int, default=1000)
    ap.add_argument("--cpu-queries", type=int, default=12)
    ap.add_argument("--seed", type=int, default=45)
    args = ap.parse_args()

    import torch
    from xsqueezeit_amd import binding, synth
    L = binding.lib()
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx = binding.Context(0, stream.cuda_stream)
    N, S, bl = args.haps, args.sites, args.block_len
    n_samples = N // 2
    thr = int(float(N) * args.maf)
    stride = synth.row_stride_bytes(N)
    p = binding.EncodeParams(n_samples, bl, thr, 1, 0, 0)
    n_blocks = (S + bl - 1) // bl
    d_bits = torch.empty(S * stride, dtype=torch.uint8, device=dev)
    binding.check(L.xsi_hip_synth_packed(ctx.handle, args.seed, 0, S, N, d_bits.data_ptr(), stride))
    cap = int(L.xsi_hip_encode_bound(ctypes.byref(p), S, S))
    d_out = torch.empty(cap, dtype=torch.uint8, device=dev)
    d_off = torch.zeros(n_blocks, dtype=torch.int64, device=dev)
    res = binding.EncodeResult()
    binding.check(L.xsi_hip_encode_packed(ctx.handle, ctypes.byref(p), d_bits.data_ptr(), S, stride, d_out.data_ptr(),
                                          cap, d_off.data_ptr(), ctypes.byref(res)))
    torch.cuda.synchronize()
    nb = res.blocks_bytes
    pad = (-(256 + nb)) % 8
    io = 256 + nb + pad
    names = b"".join(b"S%d\0" % i for i in range(n_samples))
    so = io + 8 * n_blocks
    hf = binding.HeaderFields(n_samples, 2, bl, thr, 1, 0, S, S, io, so)
    hdr = (ctypes.c_uint8 * 256)()
    binding.check(L.xsi_hip_make_header(ctypes.byref(hf), hdr))
    image = bytes(hdr) + d_out[:nb].cpu().numpy().tobytes() + b"\0" * pad + d_off.cpu().numpy().astype("<u8").tobytes() + names
    del d_out
    tmp = tempfile.NamedTemporaryFile(suffix=".xsi", delete=False)
    tmp.write(image)
    tmp.close()

    a = ctypes.c_void_p()
    binding.check(L.xsi_accessor_open(ctypes.byref(a), ctx.handle, tmp.name.encode()))
    rng = np.random.default_rng(args.seed)
    lines = rng.integers(0, S, args.queries)
    buf = np.zeros(N, dtype=np.int32)

    def expected(line):
        row = d_bits[line * stride:(line + 1) * stride].cpu().numpy()
        bits = np.unpackbits(row, bitorder="little")[:N].astype(np.int32)
        gt = (bits + 1) << 1
        gt[1::2] |= 1
        return gt

    def run(ls, check_every):
        ok = True
        t = time.perf_counter()
        for k, line in enumerate(ls):
            bm = (int(line) // bl) << 15 | (int(line) % bl)
            r = L.xsi_accessor_fill_genotype_array(a, buf.ctypes.data, buf.size, 2, bm)
            if r != N:
                raise SystemExit("fill_genotype_array failed: %s" % L.xsi_hip_last_error())
            if check_every and k % check_every == 0:
                dt_pause = time.perf_counter()
                ok = ok and bool(np.array_equal(buf, expected(int(line))))
                t += time.perf_counter() - dt_pause  # keep the check out of the timing
        return time.perf_counter() - t, ok

    t_cold, ok1 = run(lines, 97)
    u64 = ctypes.c_uint64
    cb, cby, hits, misses = u64(0), u64(0), u64(0), u64(0)
    binding.check(L.xsi_accessor_cache_stats(a, ctypes.byref(cb), ctypes.byref(cby), ctypes.byref(hits), ctypes.byref(misses)))
    t_warm, ok2 = run(rng.integers(0, S, args.queries), 97)
    starts = rng.integers(0, max(S - args.window_len, 1), args.windows)
    wl = np.concatenate([np.arange(s, s + args.window_len) for s in starts])
    t_win, ok3 = run(wl, 251)
    L.xsi_accessor_close(a)

    # CPU oracle on a few of the same random positions (fresh seek each time)
    from oracle import oracle
    rd = oracle.Reader(image)
    t = time.perf_counter()
    cpu_ok = True
    for line in lines[:args.cpu_queries]:
        bm = (int(line) // bl) << 15 | (int(line) % bl)
        gt, _ = rd.fill_genotype_array(2, bm)
        cpu_ok = cpu_ok and len(gt) == N
    t_cpu = time.perf_counter() - t
    os.unlink(tmp.name)
    out = {
        "workload": "random-access decode through xsi_accessor_fill_genotype_array, %d hap x %d sites, %d blocks, bi-allelic"
                    % (N, S, n_blocks),
        "xsi_bytes": len(image), "decoded_block_bytes_in_hbm": cby.value, "blocks_cached": cb.value,
        "cold": {"queries": int(args.queries), "seconds": t_cold, "ms_per_query": 1e3 * t_cold / args.queries,
                 "block_decodes": misses.value},
        "warm": {"queries": int(args.queries), "seconds": t_warm, "ms_per_query": 1e3 * t_warm / args.queries,
                 "cells_per_s": N * args.queries / t_warm, "int32_GBps": 4.0 * N * args.queries / t_warm / 1e9},
        "windows": {"lines": int(len(wl)), "seconds": t_win, "ms_per_line": 1e3 * t_win / len(wl),
                    "cells_per_s": N * len(wl) / t_win},
        "cpu_oracle": {"queries": int(args.cpu_queries), "seconds": t_cpu, "ms_per_query": 1e3 * t_cpu / max(args.cpu_queries, 1),
                       "cores": 1},
        "rows_match_input": bool(ok1 and ok2 and ok3), "cpu_rows_ok": bool(cpu_ok),
    }
    print(json.dumps(out))


if __name__ == "__main__":
    main()
