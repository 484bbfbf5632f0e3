#!/usr/bin/env python3
"""Rate of xsi_hip_decode_dot (phenotype dot products on decoded blocks) on the bench generator's matrix."""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--haps", type=int, default=5008)
    ap.add_argument("--sites", type=int, default=1000000)
    ap.add_argument("--pheno", type=int, default=1)
    args = ap.parse_args()
    import torch
    from xsqueezeit_amd import binding, synth
    L = binding.lib()
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx = binding.Context(0, stream.cuda_stream)
    N, S, K = args.haps, args.sites, args.pheno
    n = N // 2
    thr = int(N * 0.001)
    stride = synth.row_stride_bytes(N)
    p = binding.EncodeParams(n, 8192, thr, 1, 0, 0)
    n_blocks = (S + 8191) // 8192
    d_bits = torch.empty(S * stride, dtype=torch.uint8, device=dev)
    binding.check(L.xsi_hip_synth_packed(ctx.handle, 42, 0, S, N, d_bits.data_ptr(), stride))
    cap = int(L.xsi_hip_encode_bound(ctypes.byref(p), S, S))
    d_file = torch.empty(256 + cap + 8 * n_blocks + 64, dtype=torch.uint8, device=dev)
    d_off = torch.zeros(n_blocks, dtype=torch.int64, device=dev)
    res = binding.EncodeResult()
    binding.check(L.xsi_hip_encode_packed(ctx.handle, ctypes.byref(p), d_bits.data_ptr(), S, stride,
                                          d_file[256:].data_ptr(), cap, d_off.data_ptr(), ctypes.byref(res)))
    nb = res.blocks_bytes
    pad = (-(256 + nb)) % 8
    io = 256 + nb + pad
    so = io + 8 * n_blocks
    hf = binding.HeaderFields(n, 2, 8192, thr, 1, 0, S, S, io, so)
    hdr = (ctypes.c_uint8 * 256)()
    binding.check(L.xsi_hip_make_header(ctypes.byref(hf), hdr))
    d_file[:256] = torch.frombuffer(bytearray(hdr), dtype=torch.uint8).to(dev)
    d_file[256 + nb:io] = 0
    d_file[io:so] = d_off.view(torch.uint8)
    y = torch.randn((n, K), dtype=torch.float64, device=dev) * 10.0
    out = torch.zeros((S, K), dtype=torch.float64, device=dev)
    nbin = ctypes.c_uint64(0)

    def run():
        binding.check(L.xsi_hip_decode_dot(ctx.handle, d_file.data_ptr(), so, 0, n_blocks, y.data_ptr(), K, out.data_ptr(), S,
                                           ctypes.byref(nbin)))
    run()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 3
    # check a few lines against the input matrix
    ok = True
    yh = y.cpu().numpy()
    for line in (0, 1234, S // 2, S - 1):
        row = d_bits[line * stride:(line + 1) * stride].cpu().numpy()
        bits = np.unpackbits(row, bitorder="little")[:N].astype(np.float64)
        exp = (bits[:, None] * np.repeat(yh, 2, axis=0)).sum(0)
        ok = ok and bool(np.allclose(out[line].cpu().numpy(), exp, rtol=1e-12, atol=1e-9))
    print(json.dumps({"workload": "%d hap x %d sites, .xsi in HBM -> Sxy per line for %d phenotype(s) (decode + product)" % (N, S, K),
                      "ms": 1e3 * dt, "cells_per_s": float(N) * S / dt, "matches_input": ok}))


if __name__ == "__main__":
    main()
