#!/usr/bin/env python3
"""PCIe-inclusive rate of the file-level boundary (host int32 rows in / out), for DESIGN.md §7.

xsi_writer_append takes one host int32 row per BCF line (XsiFactoryInterface::append) and
xsi_accessor_fill_genotype_array returns one (Accessor::fill_genotype_array); both cross PCIe with
4 bytes per cell.  Not a bench line: bench.py's `value` has its inputs resident in HBM.
"""
import argparse
import ctypes
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--haps", type=int, default=5008)
    ap.add_argument("--sites", type=int, default=12 * 8192)
    args = ap.parse_args()
    import torch
    from xsqueezeit_amd import binding, synth
    L = binding.lib()
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    ctx = binding.Context(0, stream.cuda_stream)
    N, S = args.haps, args.sites
    n = N // 2
    bits = synth.synth_bits(42, 0, S, N)
    gt = synth.bits_to_gt(bits, 1)  # int32 [S, N], phased
    p = binding.EncodeParams(n, 8192, int(N * 0.001), 1, 0, 0)
    path = tempfile.NamedTemporaryFile(suffix=".xsi", delete=False).name
    names = (ctypes.c_char_p * n)(*[b"S%d" % i for i in range(n)])
    w = ctypes.c_void_p()
    binding.check(L.xsi_writer_open(ctypes.byref(w), ctx.handle, path.encode(), ctypes.byref(p), names))
    t = time.perf_counter()
    for i in range(S):
        binding.check(L.xsi_writer_append(w, gt[i].ctypes.data, N, 2))
    binding.check(L.xsi_writer_finalize(w, 0))
    t_w = time.perf_counter() - t
    L.xsi_writer_close(w)
    a = ctypes.c_void_p()
    binding.check(L.xsi_accessor_open(ctypes.byref(a), ctx.handle, path.encode()))
    buf = np.zeros(N, dtype=np.int32)
    ok = True
    t = time.perf_counter()
    for i in range(S):
        r = L.xsi_accessor_fill_genotype_array(a, buf.ctypes.data, N, 2, ((i // 8192) << 15) | (i % 8192))
        if r != N:
            raise SystemExit("fill failed")
        if i % 4099 == 0:
            ok = ok and bool(np.array_equal(buf, gt[i]))
    t_r = time.perf_counter() - t
    L.xsi_accessor_close(a)
    size = os.path.getsize(path)
    os.unlink(path)
    cells = float(N) * S
    print(json.dumps({"workload": "%d hap x %d sites through xsi_writer_append / xsi_accessor_fill_genotype_array "
                                  "(host int32 rows, one call per line)" % (N, S),
                      "write_cells_per_s": cells / t_w, "write_host_GBps": 4 * cells / t_w / 1e9,
                      "read_cells_per_s": cells / t_r, "read_host_GBps": 4 * cells / t_r / 1e9,
                      "xsi_bytes": size, "rows_match": ok}))


if __name__ == "__main__":
    main()
