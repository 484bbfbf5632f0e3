#!/usr/bin/env python3
"""bench.py — encode+decode round-trip throughput of the genotype-block path on MI355X.

One step = one pass of the hot path over one batch resident in HBM: the synthetic haplotype
matrix of BASELINE.json configs[1] (5008 haplotypes x 1,000,000 bi-allelic sites, MAC threshold
floor(5008*0.001)=5, 8192-line blocks) is encoded to the .xsi blocks region and decoded back to
packed bits.  N > 1: every rank runs the same shape on its own site range (weak scaling, no
data-path collective) and the compressed block streams are gathered to rank 0 over RCCL inside
the timed region (the path's one exchange step).  Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, MI355X_MICROARCH.md "Chip-level parameters"


def main():
    # Libraries (RCCL prints a version banner) write to stdout; the contract is ONE JSON line there.
    # Park the real stdout and point fd 1 at stderr until the line is ready.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--haps", type=int, default=5008)
    ap.add_argument("--sites", type=int, default=1_000_000)
    ap.add_argument("--block-len", type=int, default=8192)
    ap.add_argument("--maf", type=float, default=0.001)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--cpu-sample-sites", type=int, default=98304, help="sites of the CPU-oracle baseline sample")
    ap.add_argument("--cpu-threads", type=int, default=16, help="threads of the block-parallel CPU baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-dist", action="store_true",
                    help="take the multi-rank code path (process group + RCCL gather) even with one rank (testing)")
    args = ap.parse_args()

    import torch
    from xsqueezeit_amd import binding, synth, dist as xdist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    distributed = world > 1 or args.force_dist
    if distributed:
        import torch.distributed as tdist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        tdist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    else:
        tdist = None
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    L = binding.lib()
    # ONE explicit stream shared by torch and the codec (torch's default stream has handle 0, which
    # the C ABI reads as "create your own stream": two unordered streams would race)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx = binding.Context(local_rank, stream.cuda_stream)

    N, S = args.haps, args.sites
    n_samples = N // 2
    thr = int(float(N) * args.maf)
    stride = synth.row_stride_bytes(N)
    p = binding.EncodeParams(n_samples, args.block_len, thr, 1, 0, 0)
    n_blocks = (S + args.block_len - 1) // args.block_len
    first_site = rank * S  # weak scaling: rank r owns sites [r*S, (r+1)*S)

    d_bits = torch.empty(S * stride, dtype=torch.uint8, device=dev)
    binding.check(L.xsi_hip_synth_packed(ctx.handle, args.seed, first_site, S, N, d_bits.data_ptr(), stride))
    cap = int(L.xsi_hip_encode_bound(ctypes.byref(p), S, S))
    # the file image (header + blocks region + index) is assembled in place: the encoder writes the
    # blocks region straight behind the 256 header bytes
    d_file = torch.empty(256 + cap + 8 * n_blocks + 64, dtype=torch.uint8, device=dev)
    d_out = d_file[256:256 + cap]
    d_off = torch.zeros(n_blocks, dtype=torch.int64, device=dev)
    d_dec = torch.empty(S * stride, dtype=torch.uint8, device=dev)
    res = binding.EncodeResult()
    rows = ctypes.c_uint64(0)
    state = {}

    def make_file_image():
        """Header + blocks + index as one device image (what the decoder consumes)."""
        nb = res.blocks_bytes
        pad = (-(256 + nb)) % 8
        io = 256 + nb + pad
        so = io + 8 * n_blocks
        hf = binding.HeaderFields(n_samples, 2, args.block_len, thr, 1, 0, S, S, io, so)
        hdr = (ctypes.c_uint8 * 256)()
        binding.check(L.xsi_hip_make_header(ctypes.byref(hf), hdr))
        d_file[:256] = torch.frombuffer(bytearray(hdr), dtype=torch.uint8).to(dev, non_blocking=True)
        if pad:
            d_file[256 + nb:io] = 0
        d_file[io:so] = d_off.view(torch.uint8)
        state["file_len"] = so
        return so

    def step():
        binding.check(L.xsi_hip_encode_packed(ctx.handle, ctypes.byref(p), d_bits.data_ptr(), S, stride,
                                              d_out.data_ptr(), cap, d_off.data_ptr(), ctypes.byref(res)))
        handle = None
        if distributed:
            # the path's one exchange step: compressed block streams -> writer rank over RCCL/xGMI,
            # started as soon as the encode is done and overlapped with the decode (both only read d_out)
            handle = xdist.gather_block_streams_async(d_out, res.blocks_bytes, d_off - 256, tdist, dev)
        flen = make_file_image()
        binding.check(L.xsi_hip_decode_packed(ctx.handle, d_file.data_ptr(), flen, 0, n_blocks, d_dec.data_ptr(),
                                              stride, S, ctypes.byref(rows), None))
        if handle is not None:
            state["gathered"] = handle.wait()

    def fence():
        torch.cuda.synchronize()
        if distributed:
            tdist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    ctx.set_timing(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    timing = ctx.timing()
    ctx.set_timing(False)
    if distributed:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- correctness of what was timed (outside the timed region) ----
    assert rows.value == S
    roundtrip_ok = bool(torch.equal(d_dec, d_bits))
    xsi_bytes = int(res.blocks_bytes)
    cells = float(N) * float(S)
    c = xsi_bytes / cells

    out = None
    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = cells * world / (dt / args.steps)
        # dominant kernel: the PBWT chain.  Algorithmic bytes per launch:
        # encode launch = packed input read + .xsi written = cells/8 + xsi_bytes; decode launch mirrors it.
        enc_ms, enc_n = timing.get("chain_encode", (0.0, 0))
        dec_ms, dec_n = timing.get("chain_decode", (0.0, 0))
        alg_bytes = cells / 8.0 + xsi_bytes
        kern_ms = (enc_ms / max(enc_n, 1))
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
        stages = {k: round(v[0] / max(v[1], 1), 4) for k, v in timing.items() if v[1]}
        # HBM bytes of that kernel from the PMC passes kept under profiles/ (separate --pmc FETCH_SIZE /
        # WRITE_SIZE runs of this same command; FETCH_SIZE doubled per the gfx950 note in
        # MI355X_MICROARCH.md).  Only valid for the default workload; null otherwise.
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath) and (N, S, args.block_len) == (5008, 1_000_000, 8192):
            try:
                tj = json.load(open(tpath))
                key = [k for k in tj if k.startswith("xsi::k_chain_lds") and "false" in k]
                if key:
                    traffic = tj[key[0]]["hbm_bytes_per_launch"]
            except Exception:
                traffic = None
        pipeline_gbs = (cells / 4.0 + 2.0 * xsi_bytes) / (dt / args.steps) / 1e9
        out = {
            "metric": "GT cells/sec (hap x site) encode+decode round-trip",
            "value": value, "unit": "GT cells/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": "synthetic %d hap x %d biallelic sites per GPU, MAC threshold %d, %d-line blocks "
                                   "(%s), encode to .xsi + decode to packed bits, inputs in HBM"
                                   % (N, S, thr, args.block_len,
                                      "BASELINE.json configs[1]" if (N, S) == (5008, 1000000) else
                                      "BASELINE.json configs[2] shape" if (N, S) == (64976, 2000000) else
                                      "not a BASELINE.json config: parity / scaling case"),
                       "haps": N, "sites_per_gpu": S, "block_len": args.block_len, "mac_threshold": thr,
                       "seed": args.seed, "xsi_bytes_per_gpu": xsi_bytes, "bytes_per_cell": c,
                       "parallelism": "blocks sharded over %d GPU(s); RCCL gather of block streams" % world
                       if distributed else "1 GPU"},
            "roofline": {"bound": "hbm", "kernel": ("k_chain_lds" if N < 49152 else "k_chain_stream") + " (PBWT chain, encode)", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": "profiles/hbm_traffic.json (rocprofv3 --pmc passes)" if traffic else None,
                         "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": kern_ms,
                         "chain_decode_ms": dec_ms / max(dec_n, 1),
                         "pipeline_achieved_GBps": pipeline_gbs, "pipeline_frac": pipeline_gbs / HBM_PEAK_GBS,
                         "stage_ms": stages},
            "roundtrip_equal": roundtrip_ok,
        }

    # ---- CPU baseline: the oracle (parity-pinned restatement of the reference), 1 thread ----
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle
        bl = args.block_len
        cs = min(S, max(bl, args.cpu_sample_sites // bl * bl))
        packed = d_bits[:cs * stride].cpu().numpy().reshape(cs, stride)
        w = oracle.Writer(n_samples, bl, thr, 1)
        t_enc = 0.0
        chunk = 8192
        for r0 in range(0, cs, chunk):
            gt = synth.bits_to_gt(synth.unpack_rows(packed[r0:r0 + chunk], N), 1)
            t = time.perf_counter()
            w.append_rows(gt, 2)
            t_enc += time.perf_counter() - t
        t = time.perf_counter()
        ref = w.finalize(2)
        t_enc += time.perf_counter() - t
        rd = oracle.Reader(ref)
        t_dec = 0.0
        buf = np.empty((chunk, N), dtype=np.int32)
        dec_ok = True
        for r0 in range(0, cs, chunk):
            n = min(chunk, cs - r0)
            t = time.perf_counter()
            rd.fill_rows(r0, n, bl, buf)
            t_dec += time.perf_counter() - t
            if r0 == 0:
                dec_ok = bool(np.array_equal(buf[:n], synth.bits_to_gt(synth.unpack_rows(packed[:n], N), 1)))
        cpu_cells = float(N) * cs
        # bit-exactness of the GPU output on the sample: blocks are independent, so the first
        # cs/bl blocks of the GPU run must equal the blocks of the oracle's file
        nb = cs // bl
        offs = d_off[:nb + 1].cpu().numpy() if nb < n_blocks else np.append(d_off.cpu().numpy(), 256 + xsi_bytes)
        gpu_blocks = d_out[:int(offs[nb]) - 256].cpu().numpy().tobytes()
        import struct
        io = struct.unpack_from("<Q", ref, 72)[0]
        ref_region = ref[256:io]
        bit_exact = ref_region[:len(gpu_blocks)] == gpu_blocks and len(ref_region) - len(gpu_blocks) < 8
        out["cpu_baseline"] = {"value": cpu_cells / (t_enc + t_dec), "unit": "GT cells/s", "cores": 1, "kind": "port",
                               "sample": "first %d sites x %d hap of the same matrix (int32 rows in host memory), "
                                         "oracle encode %.2f s + decode %.2f s" % (cs, N, t_enc, t_dec),
                               "encode_cells_per_s": cpu_cells / t_enc, "decode_cells_per_s": cpu_cells / t_dec,
                               "decode_matches_input": dec_ok}
        out["bit_exact_vs_oracle"] = bool(bit_exact)
        out["bit_exact_blocks_checked"] = nb
        # block-parallel leg: the same sample, one thread per block range, every
        # thread with its own writer/reader (blocks are independent); ctypes drops the GIL in the calls
        n_thr = max(1, min(os.cpu_count() or 1, nb, args.cpu_threads))
        if n_thr > 1:
            from concurrent.futures import ThreadPoolExecutor
            gts = [synth.bits_to_gt(synth.unpack_rows(packed[b * bl:(b + 1) * bl], N), 1) for b in range(nb)]

            def one_range(k):
                buf_k = np.empty((bl, N), dtype=np.int32)
                for b in range(k, nb, n_thr):
                    wk = oracle.Writer(n_samples, bl, thr, 1)
                    wk.append_rows(gts[b], 2)
                    rk = oracle.Reader(wk.finalize(2))
                    rk.fill_rows(0, bl, bl, buf_k)
                return True

            t = time.perf_counter()
            with ThreadPoolExecutor(n_thr) as ex:
                list(ex.map(one_range, range(n_thr)))
            t_par = time.perf_counter() - t
            out["cpu_baseline"]["all_cores"] = {"value": float(N) * nb * bl / t_par, "unit": "GT cells/s",
                                                "cores": n_thr, "wall_s": t_par,
                                                "sample": "%d blocks of the same sample, one thread per block range" % nb}
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    ctx.close()
    if distributed:
        tdist.destroy_process_group()
    if rank == 0 and not roundtrip_ok:
        raise SystemExit("round trip mismatch")


if __name__ == "__main__":
    main()
