#!/usr/bin/env python3
"""bench.py — encode+decode round-trip throughput of the genotype-block path on MI355X.

One step = one pass of the hot path over one batch resident in HBM: a synthetic haplotype matrix is
encoded to the .xsi blocks region and decoded back to packed bits.  Workloads (BASELINE.json configs):

  --config 2 (default)  64 976 hap x 2 000 000 sites, MAC threshold 64, seed 43: the largest single-GPU
                        encode+decode configuration and the one BASELINE.json quotes the roofline on.
                        N > 1: every rank runs the same shape on its own site range (weak scaling).
  --config 1            5008 hap x 1 000 000 sites, MAC threshold 5, seed 42 (weak scaling).
  --config 4            decode-only random access through Accessor::get_genotypes on a mixed-ploidy, multi-allelic
                        200 000-haplotype file (run_config4).
  --config 3            500 000 hap x 10 000 000 sites IN TOTAL (1221 blocks of 8192 lines), seed 44: the
                        blocks are sharded over the ranks with dist.shard_blocks (strong scaling), every
                        rank generates its own shard on the device; --sites-fraction scales the total
                        (0.125 on one GPU = the 153-block shard one of 8 GPUs gets).

--gpus N without a launcher (no WORLD_SIZE in the environment) starts the N ranks itself, as children of
torch.distributed.run, before this process touches the GPU; under a launcher it is rank RANK of WORLD_SIZE.

The default command (no flags: configs[2] on one GPU) counts the rows' ALT alleles on the device INSIDE the timed
step (xsi_hip_encode_packed; --producer-counts is the opt-in for callers whose producer has the counts) and then
runs configs[1], the configs[3] shard of one of 8 GPUs and configs[4] in the same process, attached to the same JSON
line as `other_configs` (each with its own ms_per_step, roofline and parity flags; --no-other-configs skips them).

No data-path collective: blocks are independent.  The path's one exchange step, the gather of the
compressed block streams to the writer rank over RCCL, runs inside the timed region (overlapped with the
decode) and is also timed on its own after it (`gather_ms`).  Prints ONE JSON line on rank 0.
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
CONFIGS = {
    1: dict(haps=5008, sites=1_000_000, seed=42, scaling="weak", name="BASELINE.json configs[1]"),
    2: dict(haps=64976, sites=2_000_000, seed=43, scaling="weak", name="BASELINE.json configs[2]"),
    3: dict(haps=500_000, sites=10_000_000, seed=44, scaling="strong", name="BASELINE.json configs[3]"),
    4: dict(haps=200_000, sites=5_000_000, seed=45, scaling="weak", name="BASELINE.json configs[4]"),
}
# Issue model of the chain kernels.  Two pipes bound a chain line, and the slower one is its floor:
#  * vector issue: a SIMD sustains one wave64 VALU instruction per 2.8 cycles with four waves resident
#    (tools/microbench2.hip, profiles/r02_microbench2.txt); VALU instructions per 64-haplotype chunk per WAH line as
#    counted by SQ_INSTS_VALU (profiles/r02_pmc_sq_chains.txt);
#  * the LDS pipe of the CU: every chunk-line is one random ds_read_b64 gather (7.1 cycles per wave-instruction,
#    tools/microbench3.hip: its two groups of 32 lanes each wait for the fullest of 32 bank pairs) plus, on encode,
#    the atomic deposit of the next line's ones (7.4 with all lanes, less with a few) and the table build; LDS
#    array cycles per chunk-line as counted by SQ_LDS_IDX_ACTIVE / chunk-lines in the same PMC runs
#    (k_chain_rank_enc 11.84 G / 1.22 G, 61 % of them bank conflicts; k_chain_decode_rank_wg 8.99 G / 1.22 G).
SIMDS, MODEL_CLOCK_HZ, CYCLES_PER_VALU = 256 * 4, 2.4e9, 2.8
VALU_PER_CHUNK_LINE = {"k_chain_rank_enc": 9.8, "k_chain_decode_rank_wg": 12.9, "k_chain_lds": 27.0,
                       "k_chain_decode_rank": 14.0, "k_chain_decode_rank_big": 14.0, "k_chain_stream": 40.0,
                       "k_chain_rank_enc_multi": 12.0}  # _multi: 8 for gathers + updates, ~4 for appends / deposits; the per-line exchange is not in the model
LDS_CYCLES_PER_CHUNK_LINE = {"k_chain_rank_enc": 9.7, "k_chain_decode_rank_wg": 7.4, "k_chain_decode_rank": 7.4,
                             "k_chain_decode_rank_big": 7.4, "k_chain_rank_enc_multi": 9.7}


HBM_BYTES = 288 * 10**9  # MI355X, per GPU


def job_shape(args, world, rank):
    """What rank `rank` of `world` encodes and decodes: config 3 is ONE job whose blocks are sharded over the ranks
    (dist.shard_blocks: contiguous block ranges, so the gathered streams concatenate in file order; strong scaling);
    the other configs run one shape per rank on its own site range (weak scaling)."""
    from xsqueezeit_amd import dist as xdist, synth
    cfg = CONFIGS[args.config]
    N = args.haps if args.haps is not None else cfg["haps"]
    blen = args.block_len
    one_job = cfg["scaling"] == "strong" and args.sites is None
    if one_job:
        total_sites = max(blen, int(cfg["sites"] * args.sites_fraction))
        total_blocks = (total_sites + blen - 1) // blen
        b_lo, b_hi = xdist.shard_blocks(total_blocks, world, rank)
        first_site = b_lo * blen
        S = max(0, min(b_hi * blen, total_sites) - first_site)
        cells_job = float(N) * total_sites
    else:
        S = args.sites if args.sites is not None else cfg["sites"]
        first_site = rank * S  # weak scaling: rank r owns sites [r*S, (r+1)*S)
        total_sites = S * world
        total_blocks = ((S + blen - 1) // blen) * world
        cells_job = float(N) * S * world
    return dict(N=N, S=S, first_site=first_site, total_sites=total_sites, total_blocks=total_blocks, cells_job=cells_job,
                strong=one_job, block_len=blen, n_blocks=(S + blen - 1) // blen, stride=synth.row_stride_bytes(N))


def out_capacity(N, S, bound):
    """Bytes provided for a rank's blocks region: the worst-case bound (every line incompressible) is ~N/7.5 bytes per
    line; the synthetic generator needs < 0.03 B per cell at 5008 haplotypes and 0.0076 at 500 000, so large jobs get 8 GiB
    or 0.02 B per cell instead (the encoder reports XSI_ERR_CAPACITY instead of overrunning)."""
    cap = bound if bound <= (8 << 30) else max(8 << 30, int(0.02 * float(N) * S))
    return min(cap, bound)


GATHER_BYTES_PER_CELL = 0.009  # planning figure for the writer rank's receive buffer (measured at configs[3]: 0.0076)


def memory_plan(shape, world, rank, bound=None):
    """HBM a rank holds beside the library's workspace: packed input, decoded output, the file image (blocks region at
    its capacity + header + index) and, on the writer rank of a multi-rank job, the gathered block streams.  The
    workspace (permuted rows / rank-select rows: the bulk) takes what is left - the library cuts a job into batches of
    whole blocks that fit its budget - so the plan only has to leave room for ONE block's rows."""
    N, S, stride, nb = shape["N"], shape["S"], shape["stride"], shape["n_blocks"]
    if bound is None:
        bound = int((N / 7.5 + 64.0) * S) + 4096 * nb  # (no library without a build: the same order as xsi_hip_encode_bound)
    plan = {"input": S * stride, "decoded": S * stride, "file_image": 256 + out_capacity(N, S, bound) + 8 * nb + 64,
            "gathered": int(GATHER_BYTES_PER_CELL * shape["cells_job"]) + 8 * (shape["total_blocks"] + world) if (world > 1 and rank == 0) else 0,
            "row_counts": 4 * S}
    held = sum(plan.values())
    one_block_rows = 2 * shape["block_len"] * ((N + 63) // 64) * 8  # a block's permuted rows, and its rank-select rows at twice that
    plan["held"] = held
    plan["left_for_workspace"] = HBM_BYTES - held
    plan["workspace_floor"] = 2 * one_block_rows + (1 << 30)
    plan["fits"] = bool(plan["left_for_workspace"] >= plan["workspace_floor"])
    return plan


def launch_ranks(n):
    """python -m torch.distributed.run --nnodes=1 --nproc-per-node n ... bench.py <the same arguments>, one rank
    per GPU; returns its exit code."""
    import socket
    import subprocess
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def dry_launch_rank(args, real_stdout):
    """--dry-launch: what a rank does before and after the GPU work (rendezvous, per-rank report to rank 0, the
    one JSON line), over gloo and without a device."""
    import torch
    import torch.distributed as tdist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    tdist.init_process_group(backend="gloo", rank=rank, world_size=world)
    shape = job_shape(args, world, rank)
    plan = memory_plan(shape, world, rank)
    mine = torch.tensor([float(rank), float(os.getpid()), float(shape["n_blocks"]), float(shape["S"]), float(plan["held"]),
                         1.0 if plan["fits"] else 0.0], dtype=torch.float64)
    got = [torch.zeros(6, dtype=torch.float64) for _ in range(world)]
    tdist.all_gather(got, mine)
    tdist.barrier()
    if rank == 0:
        out = {"dry_launch": True, "n_gpus": world, "ranks_reported": [int(t[0]) for t in got],
               "pids": [int(t[1]) for t in got], "config": args.config,
               "blocks_per_rank": [int(t[2]) for t in got], "sites_per_rank": [int(t[3]) for t in got],
               "held_bytes_per_rank": [int(t[4]) for t in got], "fits_per_rank": [bool(t[5]) for t in got],
               "hbm_bytes": HBM_BYTES, "memory_plan_rank0": plan,
               "also_runs": ("--config 3 strong-scaled over the same ranks, attached as other_configs"
                             if (world > 1 and args.config == 2 and not args.no_other_configs) else None)}
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    tdist.destroy_process_group()
    return 0


def run_config4(args, real_stdout, emit=True):
    """BASELINE.json configs[4]: decode-only random access through Accessor::get_genotypes (accessor.hpp:58-67; the
    replay it replaces: accessor_internals_new.hpp:154-196).  Mixed-ploidy + multi-allelic content at 200 000
    haplotypes (10 % tri-allelic sites, 5 % "male" samples whose second value is end-of-vector, seed 45), built and
    encoded on the device through xsi_hip_encode_gt; the file is then served by xsi_accessor_get_genotypes.
    One step = --queries uniformly random BM positions + --windows windows of --window-len consecutive lines.
    The file of a rank is --blocks blocks of 8192 lines (the 5 M sites of the config are 611 blocks; decoded
    blocks stay resident in HBM, so the steady state does not depend on how many there are); N > 1: every rank
    serves its own block range, queries are routed by block (no exchange), weak scaling."""
    import tempfile
    import torch
    from xsqueezeit_amd import binding, synth
    cfg = CONFIGS[4]
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    distributed = (world > 1 or args.force_dist) and emit
    tdist = None
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        import torch.distributed as tdist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        tdist.init_process_group(backend="nccl", device_id=dev)
    L = binding.lib()
    stream = torch.cuda.current_stream(dev) if not emit else torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx = binding.Context(local_rank, stream.cuda_stream)
    N = args.haps if args.haps is not None else cfg["haps"]
    seed = args.seed if args.seed is not None else cfg["seed"]
    bl = args.block_len
    n_blocks = args.blocks
    S = n_blocks * bl
    first_line = rank * S
    n_samples = N // 2
    thr = int(float(N) * args.maf)
    steps = args.steps if args.steps is not None else 1
    warmup = args.warmup if args.warmup is not None else 1

    # ---- build the file on the device ----
    t0 = time.perf_counter()
    rows, nal = synth.config4_rows_device(L, ctx, torch, dev, seed, first_line, S, N, 1)
    torch.cuda.synchronize()
    t_synth = time.perf_counter() - t0
    n_bin = int((nal.astype(np.int64) - 1).sum())
    p = binding.EncodeParams(n_samples, bl, thr, 1, 0, 0)
    cap = int(L.xsi_hip_encode_gt_bound(ctypes.byref(p), S, n_bin))
    cap = min(cap, max(2 << 30, int(0.08 * float(N) * S)))
    d_out = torch.empty(cap, dtype=torch.uint8, device=dev)
    d_off = torch.zeros(n_blocks, dtype=torch.int64, device=dev)
    res = binding.EncodeResult()
    ngt = np.full(S, N, dtype=np.uint32)
    t0 = time.perf_counter()
    binding.check(L.xsi_hip_encode_gt(ctx.handle, ctypes.byref(p), rows.data_ptr(), N, S, ngt.ctypes.data, nal.ctypes.data,
                                      d_out.data_ptr(), cap, d_off.data_ptr(), ctypes.byref(res)))
    torch.cuda.synchronize()
    t_encode = time.perf_counter() - t0
    nb = int(res.blocks_bytes)
    pad = (-(256 + nb)) % 8
    io = 256 + nb + pad
    so = io + 8 * n_blocks
    hf = binding.HeaderFields(n_samples, 2, bl, thr, 1, 0, n_bin, S, io, so)
    hdr = (ctypes.c_uint8 * 256)()
    binding.check(L.xsi_hip_make_header(ctypes.byref(hf), hdr))
    names = b"".join(b"S%d\0" % i for i in range(n_samples))
    image = bytes(hdr) + d_out[:nb].cpu().numpy().tobytes() + b"\0" * pad + d_off.cpu().numpy().astype("<u8").tobytes() + names
    del d_out
    tmp = tempfile.NamedTemporaryFile(suffix=".xsi", delete=False)
    tmp.write(image)
    tmp.close()

    # ---- the queries ----
    bm = synth.bm_positions(nal, bl)
    rng = np.random.default_rng(seed + rank)
    q_lines = rng.integers(0, S, args.queries)
    starts = rng.integers(0, max(S - args.window_len, 1), args.windows)
    w_lines = (starts[:, None] + np.arange(args.window_len)[None, :]).reshape(-1) if args.windows else np.zeros(0, dtype=np.int64)
    all_lines = np.concatenate([q_lines, w_lines]).astype(np.int64)
    all_bm = bm[all_lines]
    all_na = nal[all_lines]
    a = ctypes.c_void_p()
    binding.check(L.xsi_accessor_open(ctypes.byref(a), ctx.handle, tmp.name.encode()))
    # the caller reuses one destination array (an htslib caller's gt_arr): page-locked memory from the accessor
    # (xsi_accessor_alloc_array) opts in to the direct path - the accessor does not page-lock pageable caller memory
    pbuf = ctypes.c_void_p()
    binding.check(L.xsi_accessor_alloc_array(a, N, ctypes.byref(pbuf)))
    buf = np.ctypeslib.as_array(ctypes.cast(pbuf, ctypes.POINTER(ctypes.c_int32)), shape=(N,))
    buf[:] = 0
    nout = ctypes.c_int(0)
    binding.check(L.xsi_accessor_register_array(a, buf.ctypes.data, buf.size))
    get = L.xsi_accessor_get_genotypes
    u64 = ctypes.c_uint64

    def one_pass(check_every):
        """All queries of a step through xsi_accessor_get_genotypes; every check_every-th row is compared with the
        source row (the comparison's time is taken out)."""
        ok = True
        t_chk = 0.0
        t = time.perf_counter()
        for k in range(len(all_lines)):
            r = get(a, int(all_na[k]), int(all_bm[k]), ctypes.byref(pbuf), ctypes.byref(nout))
            if r != N:
                raise SystemExit("get_genotypes failed: %s" % L.xsi_hip_last_error())
            if check_every and k % check_every == 0:
                tc = time.perf_counter()
                ok = ok and bool(np.array_equal(buf, rows[int(all_lines[k])].cpu().numpy()))
                t_chk += time.perf_counter() - tc
        return time.perf_counter() - t - t_chk, ok

    def fence():
        torch.cuda.synchronize()
        if distributed:
            tdist.barrier()
            torch.cuda.synchronize()

    # cold pass: the first touch of a block decodes all its lines on the device (what a seek replays in the reference)
    t_cold, ok_cold = one_pass(997)
    cb, cby, hits, misses = u64(0), u64(0), u64(0), u64(0)
    binding.check(L.xsi_accessor_cache_stats(a, ctypes.byref(cb), ctypes.byref(cby), ctypes.byref(hits), ctypes.byref(misses)))
    cold_misses = int(misses.value)
    for _ in range(max(warmup - 1, 0)):
        one_pass(0)
    fence()
    t0 = time.perf_counter()
    ok_warm = True
    for _ in range(steps):
        _, okp = one_pass(0)
        ok_warm = ok_warm and okp
    fence()
    dt = time.perf_counter() - t0
    _, ok_chk = one_pass(499)  # the same queries again, checked against the source rows (not timed)
    # ---- the random queries again through ONE call per batch (xsi_accessor_get_genotypes_batch): the kernels store the
    # rows of a batch into a page-locked array of args.batch rows; one launch per block touched, one completion ----
    binding.check(L.xsi_accessor_unregister_array(a))
    nb_rows = max(1, min(args.batch, len(q_lines)))
    bat = torch.empty((nb_rows, N), dtype=torch.int32, pin_memory=True)
    bat_np = bat.numpy()
    binding.check(L.xsi_accessor_register_array(a, bat_np.ctypes.data, bat_np.size))
    q_bm = np.ascontiguousarray(all_bm[:len(q_lines)], dtype=np.uint64)
    q_na = np.ascontiguousarray(all_na[:len(q_lines)], dtype=np.uint32)

    def batched_pass(check):
        ok = True
        t_chk = 0.0
        t = time.perf_counter()
        for b0 in range(0, len(q_lines), nb_rows):
            m = min(nb_rows, len(q_lines) - b0)
            r = L.xsi_accessor_get_genotypes_batch(a, m, q_na[b0:].ctypes.data, q_bm[b0:].ctypes.data, bat_np.ctypes.data, N, None)
            if r != m * N:
                raise SystemExit("get_genotypes_batch failed: %s" % L.xsi_hip_last_error())
            if check:
                tc = time.perf_counter()
                for k in range(0, m, 37):
                    ok = ok and bool(np.array_equal(bat_np[k], rows[int(q_lines[b0 + k])].cpu().numpy()))
                t_chk += time.perf_counter() - tc
        return time.perf_counter() - t - t_chk, ok

    # ---- cold isolated queries: the block is not in HBM, ONE line at binary-line offset o is asked for.  The first
    # touch runs the chain over the WAH lines in front of o only (what the reference's seek replays on the host,
    # accessor_internals_new.hpp:154-196), so its cost scales with o; full_block_ms is the same query with
    # XSI_ACCESSOR_FULL_DECODE=1 (the whole block decoded on first touch, as before round 4) ----
    cold_iso = []
    binding.check(L.xsi_accessor_unregister_array(a))
    binding.check(L.xsi_accessor_register_array(a, buf.ctypes.data, buf.size))
    budget_before = int(cby.value) * 4 + (8 << 30)
    line_of_block0 = np.arange(min(bl, S))
    for frac in (0.3, 1.0 / 128, 1.0 / 16, 0.25, 0.5, 0.97):  # the first one warms the path up and is dropped
        li = int(line_of_block0[int(frac * (len(line_of_block0) - 1))])
        rec = {"line": li, "bm_offset": int(bm[li] & 0x7FFF)}
        for mode in ("prefix", "full"):
            had_switch = os.environ.get("XSI_ENABLE_TUNING_ENV")
            if mode == "full":  # (the library ignores its tuning variables unless the process opts in)
                os.environ["XSI_ENABLE_TUNING_ENV"] = "1"
                os.environ["XSI_ACCESSOR_FULL_DECODE"] = "1"
            binding.check(L.xsi_accessor_set_cache_bytes(a, 0))       # evict everything
            binding.check(L.xsi_accessor_set_cache_bytes(a, budget_before))
            torch.cuda.synchronize()
            tq = time.perf_counter()
            r = get(a, int(nal[li]), int(bm[li]), ctypes.byref(pbuf), ctypes.byref(nout))
            rec[mode + "_ms"] = (time.perf_counter() - tq) * 1e3
            os.environ.pop("XSI_ACCESSOR_FULL_DECODE", None)
            if had_switch is None:
                os.environ.pop("XSI_ENABLE_TUNING_ENV", None)
            if r != N or not np.array_equal(buf, rows[li].cpu().numpy()):
                raise SystemExit("cold query at line %d (%s) failed: %s" % (li, mode, L.xsi_hip_last_error()))
        cold_iso.append(rec)
    cold_iso = cold_iso[1:]

    # ---- cold sequential scan (the reference's own benchmark loads every line in order,
    # loading_time/gt_loader_new.hpp:112-172): nothing in HBM, then every line of the first blocks in file order, one
    # get_genotypes call per line.  "stalls" = calls that took more than a millisecond (first touches and continuations of a
    # prefix decode).  Measured twice: with the accessor's sequential policy (one continuation per prefix-decoded block,
    # whole-block first touches, block b + 1 decoded by a second thread while b is served) and with
    # XSI_ACCESSOR_NO_READAHEAD=1, which is the accessor of rounds 4 - 5.
    def seq_scan(readahead):
        had_switch = os.environ.get("XSI_ENABLE_TUNING_ENV")
        if not readahead:
            os.environ["XSI_ENABLE_TUNING_ENV"] = "1"
            os.environ["XSI_ACCESSOR_NO_READAHEAD"] = "1"
        binding.check(L.xsi_accessor_set_cache_bytes(a, 0))       # evict everything
        binding.check(L.xsi_accessor_set_cache_bytes(a, budget_before))
        torch.cuda.synchronize()
        n_scan = min(4, n_blocks) * bl
        lat = np.empty(n_scan)
        ok = True
        ra0, rh0 = u64(0), u64(0)
        binding.check(L.xsi_accessor_readahead_stats(a, ctypes.byref(ra0), ctypes.byref(rh0)))
        t_scan = 0.0
        for li in range(n_scan):
            tq = time.perf_counter()
            r = get(a, int(nal[li]), int(bm[li]), ctypes.byref(pbuf), ctypes.byref(nout))
            lat[li] = time.perf_counter() - tq
            t_scan += lat[li]
            if r != N:
                raise SystemExit("sequential scan failed at line %d: %s" % (li, L.xsi_hip_last_error()))
            if li % 509 == 0:
                ok = ok and bool(np.array_equal(buf, rows[li].cpu().numpy()))
        os.environ.pop("XSI_ACCESSOR_NO_READAHEAD", None)
        if had_switch is None:
            os.environ.pop("XSI_ENABLE_TUNING_ENV", None)
        ra1, rh1 = u64(0), u64(0)
        binding.check(L.xsi_accessor_readahead_stats(a, ctypes.byref(ra1), ctypes.byref(rh1)))
        return {"lines": n_scan, "seconds": t_scan, "lines_per_s": n_scan / t_scan, "cells_per_s": float(N) * n_scan / t_scan,
                "stalls_over_1ms": int((lat > 1e-3).sum()), "stall_ms_total": float(lat[lat > 1e-3].sum() * 1e3),
                "max_call_ms": float(lat.max() * 1e3), "median_call_us": float(np.median(lat) * 1e6),
                "stall_calls": [[int(i), round(float(lat[i]) * 1e3, 2)] for i in np.nonzero(lat > 1e-3)[0][:24]],
                "readaheads_started": int(ra1.value - ra0.value), "first_touches_served_by_readahead": int(rh1.value - rh0.value),
                "rows_match_source": ok}

    seq_scan(True)  # (warms the second context up: its workspace is allocated on first use)
    seq_new = seq_scan(True)
    seq_old = seq_scan(False)
    binding.check(L.xsi_accessor_unregister_array(a))
    binding.check(L.xsi_accessor_register_array(a, bat_np.ctypes.data, bat_np.size))
    batched_pass(False)
    fence()
    t_b0 = time.perf_counter()
    for _ in range(steps):
        batched_pass(False)
    fence()
    dt_batched = (time.perf_counter() - t_b0) / steps
    _, ok_batched = batched_pass(True)
    binding.check(L.xsi_accessor_cache_stats(a, ctypes.byref(cb), ctypes.byref(cby), ctypes.byref(hits), ctypes.byref(misses)))
    timed_misses = int(misses.value) - cold_misses
    if distributed:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
        dt = float(t.item())
    L.xsi_accessor_close(a)
    cells_step = float(N) * len(all_lines)
    out = None
    if rank == 0:
        xsi_touched = timed_misses * (nb / n_blocks)
        alg = (xsi_touched / steps + 4.0 * cells_step)
        achieved = alg / (dt / steps) / 1e9
        out = {
            "metric": "GT cells/sec (hap x site) decode-only random access through Accessor::get_genotypes",
            "value": cells_step * world / (dt / steps), "unit": "GT cells/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int32", "data": "synthetic",
            "config": {"workload": "%s: mixed-ploidy + multi-allelic .xsi (10 %% tri-allelic sites, 5 %% male samples with "
                                   "end-of-vector second value, seed %d), %d hap, %d blocks of %d lines per GPU resident in HBM; "
                                   "one step = %d random BM queries + %d windows of %d lines through xsi_accessor_get_genotypes "
                                   "(int32 rows to host memory)" % (cfg["name"], seed, N, n_blocks, bl, args.queries, args.windows, args.window_len),
                       "haps": N, "blocks_this_gpu": n_blocks, "block_len": bl, "mac_threshold": thr, "binary_lines": n_bin,
                       "xsi_bytes_this_gpu": nb, "bytes_per_cell": nb / (float(N) * S),
                       "queries_per_step": int(len(all_lines)), "us_per_query": dt / steps / len(all_lines) * 1e6,
                       "parallelism": "queries routed by block to the rank that serves it; no exchange"},
            "roofline": {"bound": "hbm", "kernel": "k_compose_gt + D2H copy of the row (the get_genotypes boundary returns host memory)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": None,
                         "algorithmic_bytes_per_step": alg,
                         "what": "SURVEY 8d: xsi_bytes_touched + 4 x cells_returned; blocks first touched inside the timed "
                                 "region: %d (decoded blocks stay resident, LRU in HBM)" % timed_misses,
                         "pcie_note": "every row crosses PCIe (800 KB at 200 000 hap): the boundary, not HBM, bounds this path"},
            "cold_pass": {"seconds": t_cold, "block_decodes": cold_misses, "decoded_bytes_in_hbm": int(cby.value),
                          "ms_per_block_decode_upper_bound": 1e3 * t_cold / max(cold_misses, 1),
                          # the cold pass ran the same queries as a timed (warm) step: the difference is what the first
                          # touches cost (decode of all the block's lines to planes in HBM, cache allocation)
                          "ms_per_block_decode": 1e3 * max(t_cold - dt / steps, 0.0) / max(cold_misses, 1)},
            "build": {"synth_s": t_synth, "encode_gt_s": t_encode, "encode_cells_per_s": float(N) * S / t_encode},
            "batched": {"what": "the %d random queries of the step through xsi_accessor_get_genotypes_batch, %d per call, rows stored by "
                                "the compose kernels into a page-locked array" % (len(q_lines), nb_rows),
                        "us_per_query": dt_batched / max(len(q_lines), 1) * 1e6,
                        "value": float(N) * len(q_lines) / dt_batched, "unit": "GT cells/s",
                        "host_GBps": 4.0 * float(N) * len(q_lines) / dt_batched / 1e9,
                        "rows_match_source": bool(ok_batched)},
            "cold_isolated_queries": {"what": "one line asked of a block that is not in HBM, at rising offsets into the block: "
                                              "prefix_ms = first touch decodes the lines in front of it only, full_ms = the whole block",
                                      "block_lines": int(min(bl, S)), "queries": cold_iso},
            "cold_sequential_scan": dict(seq_new, what="every line of %d cold blocks in file order, one xsi_accessor_get_genotypes "
                                                       "call per line (int32 rows into a registered page-locked array)" % min(4, n_blocks),
                                         without_sequential_policy=seq_old,
                                         speedup=seq_new["lines_per_s"] / seq_old["lines_per_s"]),
            "rows_match_source": bool(ok_cold and ok_warm and ok_chk and ok_batched and seq_new["rows_match_source"] and seq_old["rows_match_source"]),
        }
        if not args.no_cpu_baseline:
            from oracle import oracle
            rd = oracle.Reader(image)
            ks = list(range(min(args.cpu_queries, len(q_lines))))
            t = time.perf_counter()
            cpu_ok = True
            for k in ks:
                gt, _ = rd.fill_genotype_array(int(all_na[k]), int(all_bm[k]))
                cpu_ok = cpu_ok and bool(np.array_equal(gt, rows[int(all_lines[k])].cpu().numpy()))
            t_cpu = time.perf_counter() - t
            out["cpu_baseline"] = {"value": float(N) * len(ks) / t_cpu, "unit": "GT cells/s", "cores": 1, "kind": "port",
                                   "cpu_model": cpu_model(),
                                   "sample": "the first %d random queries of the same file through the oracle's reader (every "
                                             "seek replays the block prefix, accessor_internals_new.hpp:154-196): %.2f s"
                                             % (len(ks), t_cpu),
                                   "ms_per_query": 1e3 * t_cpu / max(len(ks), 1), "rows_match_source": cpu_ok}
        else:
            out["cpu_baseline"] = None
        if emit:
            os.write(real_stdout, (json.dumps(out) + "\n").encode())
    os.unlink(tmp.name)
    ctx.close()
    if distributed:
        tdist.destroy_process_group()
    if not emit:
        return out
    return 0 if (rank != 0 or out["rows_match_source"]) else 1


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def main():
    # Libraries (RCCL prints a version banner) write to stdout; the contract is ONE JSON line there.
    # Park the real stdout and point fd 1 at stderr until the line is ready.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS))
    ap.add_argument("--haps", type=int, default=None, help="override the config's haplotype count")
    ap.add_argument("--sites", type=int, default=None, help="override the config's site count")
    ap.add_argument("--sites-fraction", type=float, default=1.0, help="config 3: fraction of the 10 M sites")
    ap.add_argument("--block-len", type=int, default=8192)
    ap.add_argument("--maf", type=float, default=0.001)
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--cpu-sample-cells", type=float, default=3.2e9, help="cells of the CPU-oracle baseline sample")
    ap.add_argument("--cpu-threads", type=int, default=64, help="threads of the block-parallel CPU baseline leg (capped by the host's cores and the job's blocks)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--producer-counts", action="store_true",
                    help="opt-in: the rows' ALT counts are handed over by their producer (xsi_hip_encode_packed_counted) "
                         "instead of being counted on the device inside the timed step (the default)")
    ap.add_argument("--count-on-device", action="store_true", help="(the default since round 4; kept for old command lines)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="default command only: do not run configs[1], the configs[3] shard and configs[4] after configs[2]")
    ap.add_argument("--other-configs-budget-s", type=float, default=300.0,
                    help="wall-time budget after which the remaining other configs are skipped")
    ap.add_argument("--force-dist", action="store_true",
                    help="take the multi-rank code path (process group + RCCL gather) even with one rank (testing)")
    ap.add_argument("--blocks", type=int, default=8, help="config 4: 8192-line blocks of the file each rank serves")
    ap.add_argument("--queries", type=int, default=100000, help="config 4: random BM positions per step")
    ap.add_argument("--windows", type=int, default=1000, help="config 4: contiguous windows per step")
    ap.add_argument("--window-len", type=int, default=1000)
    ap.add_argument("--cpu-queries", type=int, default=12, help="config 4: random queries of the CPU-oracle leg")
    ap.add_argument("--batch", type=int, default=256, help="config 4: lines per xsi_accessor_get_genotypes_batch call (64: 44 GB/s of rows into host memory, 256: 50, 1024: 52)")
    ap.add_argument("--dry-launch", action="store_true",
                    help="launcher check without a GPU: the ranks only form a gloo group and report in (CPU test)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # No launcher around us: start the N ranks ourselves, as CHILD processes of torch.distributed.run, before
        # anything in this process has touched HIP (a process that has initialised the GPU must never exec).
        # Rank 0 of the children prints the one JSON line to the stdout they inherit.
        os.dup2(real_stdout, 1)
        raise SystemExit(launch_ranks(args.gpus))
    if args.dry_launch:
        raise SystemExit(dry_launch_rank(args, real_stdout))
    if args.config == 4:
        raise SystemExit(run_config4(args, real_stdout))
    t_start = time.perf_counter()
    out, ok = run_roundtrip(args, emit=True)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    custom = args.haps is not None or args.sites is not None
    if rank == 0 and world == 1 and ok and args.config == 2 and not custom and not args.no_other_configs and not args.force_dist:
        # The other single-GPU configurations of BASELINE.json in the same process, attached to the same line, so
        # that they are timed by whoever runs this command and not only by the builder (VERDICT r3 #1b).
        out["other_configs"] = run_other_configs(args, t_start)
    if (world > 1 or args.force_dist) and args.config == 2 and not custom and not args.no_other_configs:
        # N > 1: the north_star job itself - BASELINE configs[3], 500 000 hap x 10 M sites, its 1221 blocks sharded over
        # these same ranks (strong scaling), with gather_ms, the ranks' times and its own roofline - attached to the
        # weak-scaling line, so that a driver that only runs the default command at N = 1, 2, 4, 8 records it
        sub = north_star_args(args)
        try:
            import gc
            import torch
            gc.collect()
            torch.cuda.empty_cache()  # (the library sizes its workspace by what the device reports free)
            o3, ok3 = run_roundtrip(sub, emit=False, dist_sub=True)
        except (Exception, SystemExit) as e:  # (every rank raises or none: the failure modes are collective)
            o3, ok3 = {"error": "%s: %s" % (type(e).__name__, e)}, False
        if rank == 0:
            o3["wall_s"] = time.perf_counter() - t_start
            out["other_configs"] = {"configs[3] strong-scaled over %d GPUs (the north_star job)" % world: o3}
            # The one N > 1 measurement that matters, a second time as FLAT scalars at the very END of the line
            # (VERDICT r5 #3): a record that keeps only the tail of stdout, or only the top level of the line, still has it.
            ns = north_star_summary(o3, world)
            out["north_star"] = ns
            for k, v in ns.items():
                out["north_star_" + k] = v
    close_process_group()
    if rank == 0:
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if rank == 0 and not ok:
        raise SystemExit("round trip mismatch")


def north_star_summary(o3, world):
    """Flat scalars of the north_star job (BASELINE configs[3] sharded over the ranks) for the top level of the line."""
    pr = o3.get("ms_per_step_per_rank") or []
    rf = o3.get("roofline") or {}
    return {
        "metric": o3.get("metric"), "workload": (o3.get("config") or {}).get("workload"),
        "value": o3.get("value"), "unit": o3.get("unit"), "ms_per_step": o3.get("ms_per_step"),
        "frac": rf.get("frac"), "gather_ms": o3.get("gather_ms"), "n_gpus": world, "scaling": "strong",
        "bit_exact_vs_oracle": o3.get("bit_exact_vs_oracle"), "roundtrip_equal": o3.get("roundtrip_equal"),
        "max_rank_ms": max(pr) if pr else None, "min_rank_ms": min(pr) if pr else None,
        "chain_encode_ms": rf.get("chain_encode_ms"), "chain_decode_ms": rf.get("chain_decode_ms"),
        "error": o3.get("error"),
    }


def north_star_args(args):
    import copy
    a = copy.copy(args)
    a.config, a.haps, a.sites, a.seed = 3, None, None, None
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        a.sites_fraction = 1.0  # the whole job; (--force-dist on one GPU: the fraction given, a rehearsal of this path)
    a.steps, a.warmup, a.cpu_sample_cells = 2, 1, 4.2e9
    return a


def close_process_group():
    try:
        import torch.distributed as tdist
        if tdist.is_available() and tdist.is_initialized():
            tdist.destroy_process_group()
    except Exception:
        pass


def run_general_path(n_haps, sites, steps=4):
    """The int32 entry points an HTSLIB caller lands on (xsi_hip_encode_gt / xsi_hip_decode_gt: htslib-encoded int32 rows
    resident in HBM -> .xsi blocks -> int32 rows; gt_block.hpp:207-406, accessor_internals_new.hpp:198-384) at a BASELINE
    shape, bi-allelic phased rows made from the bench generator's bit matrix on the device.  Algorithmic bytes per SURVEY 8d:
    4 x cells in + xsi written, xsi read + 4 x cells out."""
    import torch
    from xsqueezeit_amd import binding, synth
    L = binding.lib()
    dev = torch.device("cuda", torch.cuda.current_device())
    stream = torch.cuda.current_stream(dev)
    ctx = binding.Context(dev.index, stream.cuda_stream)
    N, S = n_haps, sites
    n, bl = N // 2, 8192
    thr = int(N * 0.001)
    stride = synth.row_stride_bytes(N)
    p = binding.EncodeParams(n, bl, thr, 1, 0, 0)
    d_gt = torch.empty((S, N), dtype=torch.int32, device=dev)
    shifts = torch.arange(8, device=dev, dtype=torch.uint8)
    odd = (torch.arange(N, device=dev) & 1).to(torch.int32)
    chunk = max(64, min(16384, int(2e9 / (4 * N))))
    d_bits = torch.empty(chunk * stride, dtype=torch.uint8, device=dev)
    for r0 in range(0, S, chunk):
        m = min(chunk, S - r0)
        binding.check(L.xsi_hip_synth_packed(ctx.handle, 42, r0, m, N, d_bits.data_ptr(), stride))
        rows = d_bits[:m * stride].view(m, stride)
        b = ((rows.unsqueeze(-1) >> shifts) & 1).reshape(m, -1)[:, :N].to(torch.int32)
        d_gt[r0:r0 + m] = ((b + 1) << 1) | odd
        del b, rows
    del d_bits
    ngt = np.full(S, N, dtype=np.uint32)
    nal = np.full(S, 2, dtype=np.uint32)
    bound = int(L.xsi_hip_encode_gt_bound(ctypes.byref(p), S, S))
    cap = out_capacity(N, S, bound)
    n_blocks = (S + bl - 1) // bl
    d_file = torch.empty(cap + 256 + 8 * n_blocks + 64, dtype=torch.uint8, device=dev)
    d_out = d_file[256:256 + cap]
    d_off = torch.zeros(n_blocks, dtype=torch.int64, device=dev)
    res = binding.EncodeResult()
    d_dec = torch.empty((S, N), dtype=torch.int32, device=dev)
    d_cnt = torch.zeros((S, 2), dtype=torch.int64, device=dev)
    ngt_out = np.zeros(S, dtype=np.uint32)
    wall = {"encode_gt": 0.0, "decode_gt": 0.0}

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
        hf = binding.HeaderFields(n, 2, bl, thr, 1, 0, S, S, io, so)
        hdr = (ctypes.c_uint8 * 256)()
        binding.check(L.xsi_hip_make_header(ctypes.byref(hf), hdr))
        d_file[:256] = torch.frombuffer(bytearray(hdr), dtype=torch.uint8).to(dev, non_blocking=True)
        d_file[256 + nb:io] = 0
        d_file[io:so] = d_off.view(torch.uint8)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        binding.check(L.xsi_hip_decode_gt(ctx.handle, d_file.data_ptr(), so, 0, n_blocks, nal.ctypes.data, S,
                                          d_dec.data_ptr(), N, ngt_out.ctypes.data, d_cnt.data_ptr(), 2))
        torch.cuda.synchronize()
        wall["encode_gt"] += t1 - t0
        wall["decode_gt"] += time.perf_counter() - t2

    step()
    wall = {k: 0.0 for k in wall}
    ctx.set_timing(True)
    for _ in range(steps):
        step()
    stages = {k: round(v[0] / steps, 4) for k, v in ctx.timing().items() if v[1]}
    ok = bool(torch.equal(d_dec, d_gt))
    cells = float(N) * S
    xsi = int(res.blocks_bytes)
    t_enc, t_dec = wall["encode_gt"] / steps, wall["decode_gt"] / steps
    alg = 8.0 * cells + 2.0 * xsi
    gbs = alg / (t_enc + t_dec) / 1e9
    ctx.close()
    return {"metric": "GT cells/sec (hap x site) encode+decode round-trip through the int32 entry points", "value": cells / (t_enc + t_dec),
            "unit": "GT cells/s", "ms_per_step": 1e3 * (t_enc + t_dec), "steps": steps, "dtype": "u16" if N <= 65535 else "u32",
            "config": {"workload": "%d hap x %d bi-allelic sites, int32 genotype rows in HBM -> xsi_hip_encode_gt -> xsi_hip_decode_gt -> "
                                   "int32 rows (the path an HTSLIB caller's rows take; the file image is assembled in place)" % (N, S),
                       "haps": N, "sites": S, "xsi_bytes": xsi, "bytes_per_cell": xsi / cells},
            "encode_ms": 1e3 * t_enc, "decode_ms": 1e3 * t_dec,
            "encode_cells_per_s": cells / t_enc, "decode_cells_per_s": cells / t_dec,
            "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None,
                         "what": "SURVEY 8d: 4 x cells read + xsi written (encode), xsi read + 4 x cells written (decode), over the wall time of the two calls",
                         "algorithmic_bytes_per_step": alg,
                         "encode_frac": (4.0 * cells + xsi) / t_enc / 1e9 / HBM_PEAK_GBS, "decode_frac": (4.0 * cells + xsi) / t_dec / 1e9 / HBM_PEAK_GBS,
                         "stage_ms_per_step": stages},
            "rows_equal": ok}


def run_file_boundary(n_samples, n_lines, zstd_level):
    """The file-level boundary from a C program (tests/c/boundary_roundtrip.c: xsi_writer_append / xsi_accessor_get_genotypes,
    host int32 rows in and out, no interpreter in the loop), PCIe-inclusive and never `value`; with zstd_level > 0 the
    outer block layer of --zstd (interfaces.hpp:288-315) on write and read."""
    import shutil
    import subprocess
    import tempfile
    gcc = shutil.which("gcc")
    if not gcc:
        return {"skipped": "no gcc on this box"}
    exe = os.path.join(tempfile.gettempdir(), "xsi_boundary_roundtrip_%d" % os.getpid())
    lib_dir = os.path.join(ROOT, "xsqueezeit_amd")
    r = subprocess.run([gcc, "-std=c99", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "boundary_roundtrip.c"),
                        "-o", exe, "-L", lib_dir, "-lxsi_hip", "-Wl,-rpath," + lib_dir], capture_output=True, text=True)
    if r.returncode:
        return {"error": "gcc: " + r.stderr[-500:]}
    path = os.path.join(tempfile.gettempdir(), "xsi_bench_%d.xsi" % os.getpid())
    try:
        r = subprocess.run([exe, path, str(n_samples), str(n_lines), "8192", str(zstd_level)], capture_output=True, text=True, timeout=600)
    finally:
        for f in (exe, path):
            if os.path.exists(f):
                os.unlink(f)
    line = (r.stdout.strip().splitlines() or [""])[-1]
    if r.returncode or not line.startswith("ok"):
        return {"error": (line + " " + r.stderr[-500:]).strip()}
    kv = dict(x.split("=") for x in line.split()[1:])
    return {"workload": "%d hap x %d lines through xsi_writer_append / xsi_accessor_get_genotypes[_batch] from C, host int32 rows, zstd level %d"
                        % (2 * n_samples, n_lines, zstd_level),
            "pcie_inclusive": True, "unit": "GT cells/s", "write": float(kv["write_cells_per_s"]), "read": float(kv["read_cells_per_s"]),
            "batch_read": float(kv["batch_read_cells_per_s"]), "every_value_equal": kv["bad_lines"] == "0"}


def run_other_configs(args, t_start):
    """configs[1], the configs[3] shard one of 8 GPUs gets, and configs[4] with 4 blocks, each with its own
    ms_per_step, roofline and parity flags; bounded by --other-configs-budget-s of wall time."""
    import copy
    import gc
    import torch
    res = {}
    plan = [
        ("configs[1]", dict(config=1, steps=5, warmup=2, cpu_sample_cells=0.6e9)),
        ("configs[3] shard (1 of 8 GPUs)", dict(config=3, sites_fraction=0.125, steps=2, warmup=1, cpu_sample_cells=4.2e9)),
        ("configs[4]", dict(config=4, blocks=4, queries=20000, windows=100, steps=1, warmup=1, cpu_queries=4)),
        # what an HTSLIB caller actually lands on: the int32 entry points on the device, and the file-level boundary with
        # and without the zstd layer (PCIe-inclusive), driver-timed since round 5
        ("general path, configs[1] shape", dict(general=(5008, 1_000_000))),
        ("general path, configs[2] haplotypes", dict(general=(64976, 262144))),
        ("file boundary, 5008 hap", dict(boundary=(2504, 100000, 0))),
        ("file boundary, 5008 hap, zstd 7", dict(boundary=(2504, 100000, 7))),
    ]
    for name, over in plan:
        gc.collect()
        torch.cuda.empty_cache()
        spent = time.perf_counter() - t_start
        if spent > args.other_configs_budget_s:
            res[name] = {"skipped": "wall-time budget of --other-configs-budget-s spent (%.0f s)" % spent}
            continue
        a = copy.copy(args)
        a.haps = a.sites = a.seed = None
        a.sites_fraction = 1.0
        for k, v in over.items():
            setattr(a, k, v)
        t = time.perf_counter()
        try:
            if "general" in over:
                o = run_general_path(*over["general"])
            elif "boundary" in over:
                o = run_file_boundary(*over["boundary"])
            elif a.config == 4:
                o = run_config4(a, None, emit=False)
            else:
                o, _ = run_roundtrip(a, emit=False)
        except (Exception, SystemExit) as e:  # a sub-run must not take the main line with it
            o = {"error": "%s: %s" % (type(e).__name__, e)}
        o["wall_s"] = time.perf_counter() - t
        res[name] = o
    return res


def run_roundtrip(args, emit=True, dist_sub=False):
    """One encode+decode configuration; returns (the JSON object of rank 0 or None, round trip equal).  emit=False:
    a sub-run of the default command (a smaller CPU-oracle sample: encode only, for the byte comparison of the first
    blocks); dist_sub: a sub-run that every rank of a multi-rank job makes together (the process group of the main
    line is reused)."""
    cfg = CONFIGS[args.config]
    custom = args.haps is not None or args.sites is not None
    N = args.haps if args.haps is not None else cfg["haps"]
    seed = args.seed if args.seed is not None else cfg["seed"]
    steps = args.steps if args.steps is not None else (5 if args.config == 1 else 3 if args.config == 2 else 2)
    warmup = args.warmup if args.warmup is not None else (2 if args.config == 1 else 1)

    import torch
    from xsqueezeit_amd import binding, synth, dist as xdist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    distributed = (world > 1 or args.force_dist) and (emit or dist_sub)
    if distributed:
        import torch.distributed as tdist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:  # --force-dist without a launcher
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local_rank)
        if not tdist.is_initialized():  # (one group per process: main() closes it)
            tdist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    else:
        tdist = None
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    L = binding.lib()
    # ONE explicit stream shared by torch and the codec (torch's default stream has handle 0, which
    # the C ABI reads as "create your own stream": two unordered streams would race)
    stream = torch.cuda.Stream(device=dev) if emit else torch.cuda.current_stream(dev)
    torch.cuda.set_stream(stream)
    ctx = binding.Context(local_rank, stream.cuda_stream)

    bl = args.block_len
    shape = job_shape(args, world, rank)
    strong, S, first_site, total_sites, cells_job = shape["strong"], shape["S"], shape["first_site"], shape["total_sites"], shape["cells_job"]
    n_samples = N // 2
    thr = int(float(N) * args.maf)
    stride = synth.row_stride_bytes(N)
    p = binding.EncodeParams(n_samples, bl, thr, 1, 0, 0)
    n_blocks = (S + bl - 1) // bl

    d_bits = torch.empty(S * stride, dtype=torch.uint8, device=dev)
    binding.check(L.xsi_hip_synth_packed(ctx.handle, seed, first_site, S, N, d_bits.data_ptr(), stride))
    # The step counts the ALT alleles of every row on the device (xsi_hip_encode_packed: the histogram half of
    # GtBlock::scan_genotypes, gt_block.hpp:207-269): the synthetic producer makes no counts.  --producer-counts is the
    # opt-in for a caller whose producer has them (the file writer's packer counts while it packs) and hands them to
    # xsi_hip_encode_packed_counted; the counting pass alone is timed either way (count_rows_ms).
    d_cnt = torch.empty(S, dtype=torch.int32, device=dev)
    binding.check(L.xsi_hip_count_packed_rows(ctx.handle, d_bits.data_ptr(), S, stride, N, d_cnt.data_ptr()))
    torch.cuda.synchronize()
    _t = time.perf_counter()
    binding.check(L.xsi_hip_count_packed_rows(ctx.handle, d_bits.data_ptr(), S, stride, N, d_cnt.data_ptr()))
    ctx.synchronize()
    count_rows_ms = (time.perf_counter() - _t) * 1e3
    count_in_step = not args.producer_counts
    cnt_ptr = None if count_in_step else d_cnt.data_ptr()
    bound = int(L.xsi_hip_encode_bound(ctypes.byref(p), S, S))
    cap = out_capacity(N, S, bound)
    plan = memory_plan(shape, world if distributed else 1, rank, bound)
    if distributed and rank == 0 and world > 1 and plan["left_for_workspace"] < (140 << 30):
        # the writer rank also receives every rank's block streams (allocated at the first exchange, behind the first
        # encode): the library's workspace - by default half of what is free when it is sized - must leave that room
        binding.check(L.xsi_hip_ctx_set_workspace_budget(ctx.handle, max(plan["workspace_floor"], int(0.9 * plan["left_for_workspace"]) - (4 << 30))))
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
        hf = binding.HeaderFields(n_samples, 2, bl, thr, 1, 0, S, S, io, so)
        hdr = (ctypes.c_uint8 * 256)()
        binding.check(L.xsi_hip_make_header(ctypes.byref(hf), hdr))
        d_file[:256] = torch.frombuffer(bytearray(hdr), dtype=torch.uint8).to(dev, non_blocking=True)
        if pad:
            d_file[256 + nb:io] = 0
        d_file[io:so] = d_off.view(torch.uint8)
        state["file_len"] = so
        return so

    # the writer rank's buffers for the gathered streams: sized like the per-rank output (every rank has the same
    # share of the job, +1 block of slack for the rounding of the shards)
    gat = None
    gat_error = None
    if distributed:
        try:
            if os.environ.get("XSI_BENCH_TORCH_GATHER"):
                raise RuntimeError("XSI_BENCH_TORCH_GATHER is set")
            gat = xdist.RcclGather(ctx, tdist, dev)
        except Exception as e:  # the library's own communicator could not be made: gather through torch.distributed
            gat_error = "%s: %s" % (type(e).__name__, e)
        # every rank must take the same path
        flag = torch.tensor([1 if gat is None else 0], dtype=torch.int32, device=dev)
        tdist.all_reduce(flag)
        if int(flag.item()) and gat is not None:
            gat.close()
            gat = None
            gat_error = gat_error or "another rank could not create its communicator"
    gat_blocks = (n_blocks + 1) * world

    def exchange():
        """The path's one exchange step: compressed block streams -> writer rank over RCCL / xGMI, through the
        library's own communicator (xsi_hip_gather_block_streams).  It runs on the communicator's stream behind the
        encode and overlaps with the decode that follows (both only read d_out)."""
        if gat is None:
            return xdist.gather_block_streams_async(d_out, res.blocks_bytes, d_off - 256, tdist, dev)
        if "gat_cap" not in state:
            # the writer rank's receive buffer: the sum of the ranks' regions (the same in every step), learnt once
            t = torch.tensor([int(res.blocks_bytes)], dtype=torch.int64, device=dev)
            tdist.all_reduce(t)
            state["gat_cap"] = int(t.item())
        return gat.gather(d_out, res.blocks_bytes, d_off - 256, state["gat_cap"], gat_blocks)

    def step():
        binding.check(L.xsi_hip_encode_packed_counted(ctx.handle, ctypes.byref(p), d_bits.data_ptr(), S, stride, cnt_ptr,
                                                      d_out.data_ptr(), cap, d_off.data_ptr(), ctypes.byref(res)))
        got = exchange() if distributed else None
        flen = make_file_image()
        binding.check(L.xsi_hip_decode_packed(ctx.handle, d_file.data_ptr(), flen, 0, n_blocks, d_dec.data_ptr(),
                                              stride, S, ctypes.byref(rows), None))
        if got is not None:
            if gat is None:
                parts = got.wait()
                if parts is not None:
                    state["gathered"] = (torch.cat(parts[0]), torch.cat(parts[1]), [int(x.numel()) for x in parts[0]],
                                         [int(x.numel()) for x in parts[1]])
            else:
                gat.wait()
                state["gathered"] = got

    def fence():
        torch.cuda.synchronize()
        if distributed:
            tdist.barrier()
            torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    fence()
    ctx.set_timing(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    timing = ctx.timing()
    ctx.set_timing(False)
    per_rank_ms = [dt / steps * 1e3]
    if distributed:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        every = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(world)]
        tdist.all_gather(every, t)
        per_rank_ms = [float(e.item()) / steps * 1e3 for e in every]
        tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
        dt = float(t.item())
    gather_ms = None
    if distributed:
        # the exchange step alone, not overlapped with anything
        fence()
        tg = time.perf_counter()
        h = exchange()
        if gat is None:
            h.wait()
        else:
            gat.wait()
        fence()
        gather_ms = (time.perf_counter() - tg) * 1e3

    # ---- correctness of what was timed (outside the timed region) ----
    assert rows.value == S
    roundtrip_ok = True
    for o in range(0, d_bits.numel(), 1 << 30):  # in pieces: torch.equal makes a temporary of the full size
        roundtrip_ok = roundtrip_ok and bool(torch.equal(d_dec[o:o + (1 << 30)], d_bits[o:o + (1 << 30)]))
    xsi_bytes = int(res.blocks_bytes)
    cells = float(N) * float(S)  # this rank
    c = xsi_bytes / cells

    out = None
    if rank == 0:
        ms_per_step = dt / steps * 1e3
        value = cells_job / (dt / steps)
        # Dominant kernel: the slower of the two PBWT chains.  Algorithmic bytes of its launch
        # (DESIGN.md §7): encode = packed input read + .xsi written = cells/8 + xsi_bytes;
        # the decode launch mirrors it.  One launch processes the whole batch of this rank.
        enc_ms, enc_n = timing.get("chain_encode", (0.0, 0))
        dec_ms, dec_n = timing.get("chain_decode", (0.0, 0))
        enc_ms, dec_ms = enc_ms / max(enc_n, 1), dec_ms / max(dec_n, 1)
        launches_per_step = max(enc_n, 1) / steps  # > 1 when the job ran as several batches of blocks
        dom_decode = dec_ms > enc_ms
        kname = L.xsi_hip_chain_kernel(N, int(n_blocks / max(launches_per_step, 1)), 1 if dom_decode else 0).decode()
        kern_ms = dec_ms if dom_decode else enc_ms
        # The dominant kernel against ITS OWN compulsory bytes: the encode chain reads the input rows of the
        # WAH lines and writes as many permuted rows; the decode chain reads a rank-select row per WAH line (10 bytes per 64
        # positions in the one-workgroup kernel's compact form, 16 otherwise) and writes the line's packed row.  The
        # headline `frac` is the whole round trip's: (cells/4 + 2 xsi_bytes) / step time, SURVEY 8d - every kernel's time in it.
        row_b = ((N + 63) // 64) * 8.0
        wah_l = float(res.n_wah_lines) / launches_per_step
        if dom_decode:
            kern_bytes = wah_l * (row_b * (1.25 if kname == "k_chain_decode_rank_wg" else 2.0) + row_b)
            kern_bytes_what = "rank-select row read + packed row written per WAH line"
        else:
            kern_bytes = wah_l * 2.0 * row_b
            kern_bytes_what = "input row read + permuted row written per WAH line"
        kern_gbs = kern_bytes / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
        # (what rounds 1-4 printed as `achieved`: the whole ENCODE's algorithmic bytes over the chain kernel's time alone)
        alg_bytes = (cells / 8.0 + xsi_bytes) / launches_per_step
        achieved_r4 = alg_bytes / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
        stages = {k: round(v[0] / max(v[1], 1), 4) for k, v in timing.items() if v[1]}  # per launch
        # per step: a job that runs as several batches of blocks launches a stage several times a step
        stages_step = {k: round(v[0] / steps, 4) for k, v in timing.items() if v[1]}
        launches = {"encode": enc_n / steps, "decode": dec_n / steps}
        # issue model of that launch: the slower of the vector-issue pipe (all SIMDs) and the LDS pipe (the CUs that
        # hold a workgroup: one per block up to the chip's 256)
        chunk_lines = float(res.n_wah_lines) * ((N + 63) // 64) / launches_per_step
        vpc = VALU_PER_CHUNK_LINE.get(kname)
        lpc = LDS_CYCLES_PER_CHUNK_LINE.get(kname)
        valu_ms = chunk_lines * vpc * CYCLES_PER_VALU / (SIMDS * MODEL_CLOCK_HZ) * 1e3 if vpc else None
        wgs_per_block = max(1, (N + 65535) // 65536) if kname == "k_chain_rank_enc_multi" else 1
        cus_busy = min(256.0, float(int(n_blocks / max(launches_per_step, 1))) * wgs_per_block)
        lds_ms = chunk_lines * lpc / (cus_busy * MODEL_CLOCK_HZ) * 1e3 if lpc else None
        model_ms = max(x for x in (valu_ms, lds_ms) if x is not None) if (valu_ms or lds_ms) else None
        # HBM bytes of that kernel from the PMC passes kept under profiles/ (separate --pmc FETCH_SIZE /
        # WRITE_SIZE runs of this same command, FETCH_SIZE doubled per the gfx950 note in
        # MI355X_MICROARCH.md); labelled with the commit they were taken at, null for other workloads
        traffic = traffic_src = kernel_traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        # (weak scaling: every rank runs the profiled shape; configs[3]: the shard of one of 8 GPUs was profiled)
        if os.path.exists(tpath) and not custom and (world == 1 or not strong) and (args.config != 3 or abs(args.sites_fraction - 0.125) < 1e-9):
            try:
                tj = json.load(open(tpath)).get("config%d" % args.config, {})
                ks = tj.get("kernels", {})
                ent = ks.get(kname)
                if ent:
                    kernel_traffic = ent["hbm_bytes_per_launch"]
                    # per STEP, like `achieved`: every kernel's bytes per launch x its launches per step of the profiled run
                    steps_prof = float(tj.get("steps_profiled") or ks.get("k_classify", {}).get("launches_in_the_profiled_run") or 1)
                    # (outside the steps the command launches the generator and two counting passes: not a step's traffic)
                    setup = {"k_synth_packed": None, "k_count_rows_wide": 2, "k_count_rows": 2, "k_count_rows_v4": 2}
                    traffic = sum(v["hbm_bytes_per_launch"] * (v["launches_in_the_profiled_run"] - (setup.get(k) or 0))
                                  for k, v in ks.items() if not (k in setup and setup[k] is None)) / steps_prof
                    traffic_src = ("profiles/hbm_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command on one GPU at %s; "
                                   "all kernels of a step" % tj.get("measured_at", "?"))
            except Exception:
                traffic = kernel_traffic = None
        pipeline_gbs = (cells / 4.0 + 2.0 * xsi_bytes) / (dt / steps) / 1e9
        out = {
            "metric": "GT cells/sec (hap x site) encode+decode round-trip",
            "value": value, "unit": "GT cells/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "u16" if n_samples * 2 <= 65535 else "u32", "data": "synthetic",
            "config": {"workload": "synthetic %d hap x %d biallelic sites %s, MAC threshold %d, %d-line blocks "
                                   "(%s), encode to .xsi + decode to packed bits, inputs in HBM"
                                   % (N, total_sites if strong else S, "in total, blocks sharded over the ranks" if strong
                                      else "per GPU", thr, bl,
                                      "not a BASELINE.json config: parity / scaling case" if custom else
                                      cfg["name"] + (" x %.4g of the sites" % args.sites_fraction if strong and args.sites_fraction != 1.0 else "")),
                       "haps": N, "sites_this_gpu": S, "blocks_this_gpu": n_blocks, "block_len": bl, "mac_threshold": thr,
                       "seed": seed, "xsi_bytes_this_gpu": xsi_bytes, "bytes_per_cell": c,
                       "wah_lines_this_gpu": int(res.n_wah_lines), "row_stride_bytes": stride,
                       "launches_per_step": launches_per_step,
                       "row_counts": ("counted on the device inside the timed step (xsi_hip_encode_packed)" if count_in_step else
                                      "--producer-counts: handed over with the rows (xsi_hip_encode_packed_counted); the counting "
                                      "pass alone takes count_rows_ms and is NOT in ms_per_step"),
                       "count_rows_ms": count_rows_ms,
                       "memory_plan_bytes": plan,
                       "parallelism": "blocks sharded over %d GPU(s); RCCL gather of block streams" % world
                       if distributed else "1 GPU"},
            "roofline": {"bound": "hbm", "achieved": pipeline_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": pipeline_gbs / HBM_PEAK_GBS,
                         "what": "whole encode + decode round trip: (cells/4 + 2 x xsi_bytes) algorithmic bytes of this GPU / ms_per_step",
                         "algorithmic_bytes_per_step": cells / 4.0 + 2.0 * xsi_bytes,
                         "traffic": traffic, "traffic_source": traffic_src, "kernel_traffic": kernel_traffic,
                         "kernel": "%s (PBWT chain, %s)" % (kname, "decode" if dom_decode else "encode"),
                         "kernel_ms": kern_ms, "kernel_bytes_per_launch": kern_bytes, "kernel_bytes_are": kern_bytes_what,
                         "kernel_achieved": kern_gbs, "kernel_frac": kern_gbs / HBM_PEAK_GBS,
                         "kernel_achieved_by_round4_accounting": achieved_r4,
                         "issue_model": {"what": "floor of this launch: max(vector issue: chunk-lines x VALU per 64-haplotype chunk "
                                                 "per line x 2.8 cycles / (1024 SIMDs x 2.4 GHz), LDS pipe: chunk-lines x LDS array "
                                                 "cycles per chunk-line (random ds_read_b64 gather 7.1 + deposits) / (busy CUs x 2.4 GHz))",
                                         "valu_per_chunk_line": vpc, "lds_cycles_per_chunk_line": lpc, "busy_cus": cus_busy,
                                         "valu_ms": valu_ms, "lds_ms": lds_ms, "model_ms": model_ms,
                                         "bound": ("lds" if (lds_ms or 0) >= (valu_ms or 0) else "valu") if model_ms else None,
                                         "achieved_over_model": (model_ms / kern_ms) if model_ms and kern_ms else None},
                         "chain_encode_ms": enc_ms, "chain_decode_ms": dec_ms,
                         "pipeline_achieved_GBps": pipeline_gbs, "pipeline_frac": pipeline_gbs / HBM_PEAK_GBS,  # (= achieved / frac; kept for readers of rounds 1-4)
                         "stage_ms": stages, "stage_ms_per_step": stages_step, "chain_launches_per_step": launches},
            "roundtrip_equal": roundtrip_ok,
        }
        if not count_in_step:
            # the same step with the counting pass made on the device as well (measured by itself, outside the timed
            # region, added here): what xsi_hip_encode_packed without producer counts costs
            out["with_device_row_count"] = {"ms_per_step": ms_per_step + count_rows_ms,
                                            "value": cells_job / ((dt / steps) + count_rows_ms * 1e-3)}
        if gather_ms is not None:
            out["gather_ms"] = gather_ms
            out["ms_per_step_per_rank"] = per_rank_ms
            region_all, offs_all, per_b, per_n = state["gathered"]  # the last timed step's exchange, on the writer rank
            offs_np = offs_all.cpu().numpy()
            out["gathered"] = {"ranks": len(per_b), "bytes": int(region_all.numel()), "blocks": int(offs_np.size),
                               "bytes_per_rank": [int(x) for x in per_b],
                               "offsets_ascending": bool(np.all(np.diff(offs_np) > 0)) if offs_np.size > 1 else True,
                               "own_part_equals_encode_output": bool(torch.equal(region_all[:xsi_bytes], d_out[:xsi_bytes])),
                               "via": "xsi_hip_gather_block_streams (RCCL from libxsi_hip.so)" if gat is not None
                               else "torch.distributed point-to-point (the library's communicator failed: %s)" % gat_error}

    # ---- CPU baseline: the oracle (parity-pinned restatement of the reference), 1 thread ----
    if rank == 0 and not args.no_cpu_baseline:  # (N > 1: rank 0's host cores, while the other ranks wait at the group's next collective)
        from oracle import oracle
        full_leg = emit  # a sub-run only encodes its sample with the oracle, for the byte comparison
        cs = int(args.cpu_sample_cells / N)
        cs = cs // bl * bl if cs >= bl else max(1024, cs // 1024 * 1024)
        cs = min(S, cs, 12 * bl)
        packed = d_bits[:cs * stride].cpu().numpy().reshape(cs, stride)
        w = oracle.Writer(n_samples, bl, thr, 1)
        t_enc = 0.0
        chunk = max(1, min(8192, int(2e9 / (4 * N))))  # int32 rows of a chunk stay under ~2 GB
        for r0 in range(0, cs, chunk):
            gt = synth.bits_to_gt(synth.unpack_rows(packed[r0:r0 + chunk], N), 1)
            t = time.perf_counter()
            w.append_rows(gt, 2)
            t_enc += time.perf_counter() - t
        t = time.perf_counter()
        ref = w.finalize(2)
        t_enc += time.perf_counter() - t
        rd = oracle.Reader(ref) if full_leg else None
        t_dec = 0.0
        buf = np.empty((chunk, N), dtype=np.int32) if full_leg else None
        dec_ok = True if full_leg else None
        for r0 in (range(0, cs, chunk) if full_leg else ()):
            n = min(chunk, cs - r0)
            t = time.perf_counter()
            rd.fill_rows(r0, n, bl, buf)
            t_dec += time.perf_counter() - t
            if r0 == 0:
                dec_ok = bool(np.array_equal(buf[:n], synth.bits_to_gt(synth.unpack_rows(packed[:n], N), 1)))
        del buf
        cpu_cells = float(N) * cs
        # bit-exactness of the GPU output on the sample: blocks are independent, so the first
        # cs/bl blocks of the GPU run must equal the blocks of the oracle's file
        nb = cs // bl
        bit_exact = None
        if nb:
            offs = d_off[:nb + 1].cpu().numpy() if nb < n_blocks else np.append(d_off.cpu().numpy(), 256 + xsi_bytes)
            gpu_blocks = d_out[:int(offs[nb]) - 256].cpu().numpy().tobytes()
            import struct
            io = struct.unpack_from("<Q", ref, 72)[0]
            ref_region = ref[256:io]
            bit_exact = bool(ref_region[:len(gpu_blocks)] == gpu_blocks and len(ref_region) - len(gpu_blocks) < 8)
        if full_leg:
            out["cpu_baseline"] = {"value": cpu_cells / (t_enc + t_dec), "unit": "GT cells/s", "cores": 1, "kind": "port",
                                   "cpu_model": cpu_model(), "host_cores": os.cpu_count(),
                                   "sample": "first %d sites x %d hap of the same matrix (int32 rows in host memory), "
                                             "oracle encode %.2f s + decode %.2f s" % (cs, N, t_enc, t_dec),
                                   "encode_cells_per_s": cpu_cells / t_enc, "decode_cells_per_s": cpu_cells / t_dec,
                                   "decode_matches_input": dec_ok}
        else:
            out["cpu_baseline"] = {"value": cpu_cells / t_enc, "unit": "GT cells/s (encode only)", "cores": 1, "kind": "port",
                                   "sample": "first %d sites x %d hap of the same matrix, oracle encode %.2f s" % (cs, N, t_enc)}
        out["bit_exact_vs_oracle"] = bit_exact
        out["bit_exact_blocks_checked"] = nb
        # block-parallel leg: one thread per block, every thread with its own
        # writer/reader (blocks are independent); ctypes drops the GIL in the calls.  The int32 rows of
        # a block are made inside its thread and dropped again, so host memory stays at threads x block.
        # Since round 6 a thread feeds its codec in slices of `sl` lines (the whole block as int32 rows,
        # 2.1 GB per thread at 64 976 haplotypes, had capped the leg at 11 threads): host memory is threads x slice, and the
        # leg runs on as many threads as the host has cores, up to --cpu-threads (64) and the blocks of the job.
        sl = max(64, min(bl, int(256e6 / (4.0 * N))))
        try:
            usable = len(os.sched_getaffinity(0))
        except (AttributeError, OSError):
            usable = os.cpu_count() or 1
        n_thr = max(1, min(os.cpu_count() or 1, args.cpu_threads, int(48e9 / (4.0 * N * sl))))
        par_blocks = (min(n_blocks, n_thr) if cs >= bl else 0) if full_leg else 0
        if par_blocks > 1:
            from concurrent.futures import ThreadPoolExecutor
            pk = d_bits[:par_blocks * bl * stride].cpu().numpy().reshape(par_blocks * bl, stride)
            busy = [0.0] * par_blocks
            t_wall0 = time.perf_counter()

            def one_block(b):
                wk = oracle.Writer(n_samples, bl, thr, 1)
                tb = 0.0
                gt = None
                for r0 in range(0, bl, sl):
                    gt = synth.bits_to_gt(synth.unpack_rows(pk[b * bl + r0:b * bl + min(bl, r0 + sl)], N), 1)
                    t0 = time.perf_counter()
                    wk.append_rows(gt, 2)
                    tb += time.perf_counter() - t0
                del gt
                buf = np.empty((min(sl, bl), N), dtype=np.int32)  # (the last encode slice may be shorter than a decode slice)
                t0 = time.perf_counter()
                rk = oracle.Reader(wk.finalize(2))
                for r0 in range(0, bl, sl):
                    rk.fill_rows(r0, min(sl, bl - r0), bl, buf)
                busy[b] = tb + time.perf_counter() - t0
                return True

            with ThreadPoolExecutor(par_blocks) as ex:
                list(ex.map(one_block, range(par_blocks)))
            t_par = max(busy)  # the threads run side by side: the slowest thread's codec calls (its row making left out)
            out["cpu_baseline"]["all_cores"] = {"value": float(N) * par_blocks * bl / t_par, "unit": "GT cells/s",
                                                "cores": par_blocks, "usable_cpus": usable, "host_cores": os.cpu_count(),
                                                "wall_s": t_par, "leg_wall_s": time.perf_counter() - t_wall0,
                                                "sample": "%d blocks of the same matrix, one thread per block, %d lines per "
                                                          "codec call (slowest thread's encode+decode time)" % (par_blocks, sl)}
    elif rank == 0:
        out["cpu_baseline"] = None
    if distributed:
        tdist.barrier()  # (rank 0 comes out of its CPU leg)
    if gat is not None:
        gat.close()
    ctx.close()
    if distributed:
        ok_t = torch.tensor([1 if roundtrip_ok else 0], dtype=torch.int32, device=dev)
        tdist.all_reduce(ok_t, op=tdist.ReduceOp.MIN)  # the verdict of the job: every rank's round trip
        roundtrip_ok = bool(int(ok_t.item()))
        if out is not None:
            out["roundtrip_equal"] = roundtrip_ok
    return out, roundtrip_ok


if __name__ == "__main__":
    main()
