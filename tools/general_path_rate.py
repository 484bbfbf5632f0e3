#!/usr/bin/env python3
"""On-device rate of the general (int32 genotype rows) entry points, xsi_hip_encode_gt /
xsi_hip_decode_gt: htslib-encoded int32 rows resident in HBM -> .xsi blocks -> int32 rows.
Bi-allelic phased rows made from the bench generator's bit matrix on the device.  Profile tool,
not a bench line (bench.py measures the packed entry points)."""
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
    ap.add_argument("--sites", type=int, default=262144)
    ap.add_argument("--steps", type=int, default=3)
    args = ap.parse_args()
    import torch
    from xsqueezeit_amd import binding, synth
    L = binding.lib()
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx = binding.Context(0, stream.cuda_stream)
    N, S = args.haps, args.sites
    n = N // 2
    thr = int(N * 0.001)
    stride = synth.row_stride_bytes(N)
    p = binding.EncodeParams(n, 8192, thr, 1, 0, 0)
    d_bits = torch.empty(S * stride, dtype=torch.uint8, device=dev)
    binding.check(L.xsi_hip_synth_packed(ctx.handle, 42, 0, S, N, d_bits.data_ptr(), stride))
    # int32 rows on the device: ((bit + 1) << 1) | phased(odd index)
    d_gt = torch.empty((S, N), dtype=torch.int32, device=dev)
    shifts = torch.arange(8, device=dev, dtype=torch.uint8)
    odd = (torch.arange(N, device=dev) & 1).to(torch.int32)
    chunk = 16384
    for r0 in range(0, S, chunk):
        rows = d_bits[r0 * stride:(r0 + chunk) * stride].view(-1, stride)
        b = ((rows.unsqueeze(-1) >> shifts) & 1).reshape(rows.shape[0], -1)[:, :N].to(torch.int32)
        d_gt[r0:r0 + rows.shape[0]] = ((b + 1) << 1) | odd
    ngt = np.full(S, N, dtype=np.uint32)
    nal = np.full(S, 2, dtype=np.uint32)
    cap = int(L.xsi_hip_encode_gt_bound(ctypes.byref(p), S, S))
    d_out = torch.empty(cap, dtype=torch.uint8, device=dev)
    n_blocks = (S + 8191) // 8192
    d_off = torch.zeros(n_blocks, dtype=torch.int64, device=dev)
    res = binding.EncodeResult()
    d_dec = torch.empty((S, N), dtype=torch.int32, device=dev)
    d_cnt = torch.zeros((S, 2), dtype=torch.int64, device=dev)
    ngt_out = np.zeros(S, dtype=np.uint32)
    d_file = torch.empty(cap + 256 + 8 * n_blocks + 64, dtype=torch.uint8, device=dev)

    wall = {"encode_gt": 0.0, "harness_between": 0.0, "decode_gt": 0.0}

    def step():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        binding.check(L.xsi_hip_encode_gt(ctx.handle, ctypes.byref(p), d_gt.data_ptr(), N, S, ngt.ctypes.data,
                                          nal.ctypes.data, d_out.data_ptr(), cap, d_off.data_ptr(), ctypes.byref(res)))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        nb = res.blocks_bytes
        pad = (-(256 + nb)) % 8
        io = 256 + nb + pad
        so = io + 8 * n_blocks
        hf = binding.HeaderFields(n, 2, 8192, thr, 1, 0, S, S, io, so)
        hdr = (ctypes.c_uint8 * 256)()
        binding.check(L.xsi_hip_make_header(ctypes.byref(hf), hdr))
        d_file[:256] = torch.frombuffer(bytearray(hdr), dtype=torch.uint8).to(dev, non_blocking=True)
        d_file[256:256 + nb] = d_out[:nb]
        d_file[256 + nb:io] = 0
        d_file[io:so] = d_off.view(torch.uint8)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        binding.check(L.xsi_hip_decode_gt(ctx.handle, d_file.data_ptr(), so, 0, n_blocks, nal.ctypes.data, S,
                                          d_dec.data_ptr(), N, ngt_out.ctypes.data, d_cnt.data_ptr(), 2))
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        wall["encode_gt"] += t1 - t0
        wall["harness_between"] += t2 - t1
        wall["decode_gt"] += t3 - t2

    step()
    torch.cuda.synchronize()
    for k in wall:
        wall[k] = 0.0
    ctx.set_timing(True)
    t = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / args.steps
    stages = {k: round(v[0] / args.steps, 4) for k, v in ctx.timing().items() if v[1]}
    ok = bool(torch.equal(d_dec, d_gt))
    cells = float(N) * S
    print(json.dumps({"workload": "%d hap x %d bi-allelic sites, int32 rows in HBM -> xsi_hip_encode_gt -> xsi_hip_decode_gt -> int32 rows"
                                  % (N, S), "ms_per_step": 1e3 * dt, "cells_per_s": cells / dt,
                      "int32_GBps_each_way": 4 * cells / dt / 1e9, "xsi_bytes": int(res.blocks_bytes), "rows_equal": ok,
                      "wall_ms_per_step": {k: round(1e3 * v / args.steps, 3) for k, v in wall.items()}, "stage_ms_per_step": stages}))


if __name__ == "__main__":
    main()
