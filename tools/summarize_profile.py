#!/usr/bin/env python3
"""Fold the output of tools/collect_profiles.sh (bench lines, rocprofv3 --stats, --pmc FETCH_SIZE, --pmc
WRITE_SIZE and two SQ counter passes, all CSV) into the files kept under profiles/.

usage: python tools/summarize_profile.py gpurun_out r02 profiles "commit abc1234, 2026-10-04"

HBM traffic follows MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are in KiB (x1024); on gfx950
FETCH_SIZE reports half of the bytes of a wide coalesced streaming read, so the read side is doubled - and,
because the guide calls other access widths uncalibrated, the doubled figure of every kernel that moves
gigabytes is set against a byte count known from the workload (KNOWN_READS below: the input matrix, the WAH
lines' share of it, the permuted rows the chain wrote, the .xsi bytes); the ratio is printed per kernel.
WRITE_SIZE is taken as is.  The two counters come from separate passes.  Values are per launch.
"""
import collections
import csv
import json
import os
import re
import shutil
import sys


def short_name(k):
    k = k.replace("void ", "").split("(")[0]
    k = re.sub(r"<.*", "", k)
    return k.replace("xsi::", "")


def per_kernel(counter_csv):
    """{(short kernel name, counter): (mean value per launch, launches)}"""
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(counter_csv)):
        a = agg[(short_name(r["Kernel_Name"]), r["Counter_Name"])]
        a[0] += float(r["Counter_Value"])
        a[1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in agg.items()}


def known_reads(bench):
    """Bytes each heavy kernel must read per LAUNCH at the bench workload, from quantities the bench line states."""
    c = bench["config"]
    lines, stride = c["sites_this_gpu"], c.get("row_stride_bytes", ((c["haps"] + 1023) // 1024) * 128)
    wah = c.get("wah_lines_this_gpu", 0)
    inp = float(lines) * stride
    f = wah / float(lines) if lines else 0.0
    y_row = ((c["haps"] + 63) // 64) * 8.0
    return {
        "k_count_rows_wide": (inp, "the packed input matrix, streamed once with 16-byte loads"),
        "k_count_rows": (inp, "the packed input matrix, streamed once"),
        "k_chain_rank_enc": (f * inp, "the input rows of the WAH lines (row prefetch, 8 bytes per lane, + scalar loads)"),
        "k_sparse_write": ((1.0 - f) * inp, "the input rows of the sparse lines"),
        "k_wah_units": (wah * y_row, "the permuted rows the chain wrote, read once by the sizing pass (which leaves the words in the rows)"),
        # per STEP: the caller divides by the launches a step makes (the phased decode runs the chain range by range)
        "k_chain_decode_rank_wg": (wah * (y_row * 1.25), "the compact rank-select rows (10 bytes per 64 positions), all ranges of a step"),
        "k_wah_tile_sums": (c["xsi_bytes_this_gpu"] * 0.97, "the WAH matrices of the file image"),
        "k_wah_boundaries": (c["xsi_bytes_this_gpu"] * 0.97, "the WAH matrices of the file image"),
    }


def main():
    src, tag, dst, when = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4]
    os.makedirs(dst, exist_ok=True)
    p = lambda *a: os.path.join(src, *a)
    for name in ("bench", "bench_config1", "config3_shard"):
        f = p("%s_%s.json" % (tag, name))
        if os.path.exists(f):
            shutil.copy(f, os.path.join(dst, "%s_%s.json" % (tag, name)))
    stats = p("%s_stats" % tag, "p_kernel_stats.csv")
    if os.path.exists(stats):
        shutil.copy(stats, os.path.join(dst, "%s_kernel_stats.csv" % tag))
    c3 = p("%s_c3stats" % tag, "p_kernel_stats.csv")
    if os.path.exists(c3):
        shutil.copy(c3, os.path.join(dst, "%s_config3_kernel_stats.csv" % tag))
    clocks = p("%s_config3_phase_clocks.txt" % tag)
    if os.path.exists(clocks) and os.path.getsize(clocks):
        with open(os.path.join(dst, "%s_config3_phase_clocks.txt" % tag), "w") as out:
            out.write("XSI_MULTI_PROF=983041 (wave 15 of workgroup 0) of k_chain_rank_enc_multi over the configs[3] shard, %s:\n"
                      "time between consecutive phase records of that wave, summed per phase (the records cost the chain ~10 %%).\n\n" % when)
            out.write(open(clocks).read())
    sq3 = {}
    for part in ("c3sq1", "c3sq2"):
        f = p("%s_%s" % (tag, part), "p_counter_collection.csv")
        if os.path.exists(f):
            sq3.update(per_kernel(f))
    if sq3:
        with open(os.path.join(dst, "%s_pmc_sq_config3.txt" % tag), "w") as out:
            out.write("rocprofv3 --pmc (two passes of SQ counters) of `python3 bench.py --config 3 --sites-fraction 0.125 --steps 1 --warmup 1\n"
                      "--no-cpu-baseline` (500 000 hap x 153 blocks), %s; counters summed over all waves, mean per launch.\n"
                      "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md).\n\n" % when)
            for kern in sorted({k[0] for k in sq3}):
                if "chain" not in kern and "wah" not in kern:
                    continue
                out.write("%s\n" % kern)
                for (kk, c), (v, n) in sorted(sq3.items()):
                    if kk == kern:
                        out.write("  %-22s %18.0f   (%d launches)\n" % (c, v, n))
                wc = sq3.get((kern, "SQ_WAVE_CYCLES"), (0, 0))[0]
                if wc:
                    for c, label in (("SQ_ACTIVE_INST_ANY", "issuing"), ("SQ_WAIT_ANY", "waiting"), ("SQ_WAIT_INST_ANY", "waiting to issue")):
                        v = sq3.get((kern, c))
                        if v:
                            out.write("  %-22s %17.1f %% of wave cycles\n" % (label, 100.0 * v[0] / wc))
                out.write("\n")
    fetch, write = p("%s_fetch" % tag, "p_counter_collection.csv"), p("%s_write" % tag, "p_counter_collection.csv")
    if os.path.exists(fetch) and os.path.exists(write):
        f, w = per_kernel(fetch), per_kernel(write)
        kernels = {}
        rows = []
        bj = p("%s_bench.json" % tag)
        known = known_reads(json.loads(open(bj).read().strip().splitlines()[-1])) if os.path.exists(bj) else {}
        for name in sorted({k[0] for k in f} | {k[0] for k in w}):
            if not name.startswith("k_"):
                continue
            fk, launches = f.get((name, "FETCH_SIZE"), (0.0, 0))
            fk *= 1024.0
            wk = w.get((name, "WRITE_SIZE"), (0.0, 0))[0] * 1024.0
            kb, why = known.get(name, (None, ""))
            if kb and name == "k_chain_decode_rank_wg":
                kb /= max(1.0, launches / 2.0)  # the profiled command runs two steps (--steps 1 --warmup 1)
            kernels[name] = {"fetch_size_bytes_raw": fk, "read_bytes_corrected": 2.0 * fk, "write_bytes": wk,
                             "hbm_bytes_per_launch": 2.0 * fk + wk, "launches_in_the_profiled_run": launches,
                             "known_read_bytes": kb, "x2_over_known": (2.0 * fk / kb) if kb else None, "known_read_is": why}
            rows.append((name, launches, fk, 2.0 * fk, wk, 2.0 * fk + wk, kb or 0.0, (2.0 * fk / kb) if kb else 0.0, why))
        with open(os.path.join(dst, "%s_hbm_traffic.csv" % tag), "w") as out:
            out.write("kernel,launches,FETCH_SIZE_bytes_raw,read_bytes_x2_gfx950,WRITE_SIZE_bytes,hbm_bytes_per_launch,"
                      "known_read_bytes_per_launch,x2_over_known,known_read_is\n")
            for r in sorted(rows, key=lambda x: -x[5] * max(x[1], 1)):
                out.write("%s,%d,%.0f,%.0f,%.0f,%.0f,%.0f,%.3f,%s\n" % r)
        tpath = os.path.join(dst, "hbm_traffic.json")
        tj = json.load(open(tpath)) if os.path.exists(tpath) else {}
        if "config2" not in tj and "config1" not in tj:
            tj = {}  # round-1 layout
        tj["config2"] = {"measured_at": when, "command": "python bench.py (BASELINE configs[2])", "kernels": kernels}
        json.dump(tj, open(tpath, "w"), indent=1, sort_keys=True)
    # the same two passes over the configs[3] shard (the long-row kernels)
    f3, w3 = p("%s_c3fetch" % tag, "p_counter_collection.csv"), p("%s_c3write" % tag, "p_counter_collection.csv")
    if os.path.exists(f3) and os.path.exists(w3):
        f, w = per_kernel(f3), per_kernel(w3)
        kernels, rows = {}, []
        for name in sorted({k[0] for k in f} | {k[0] for k in w}):
            if not name.startswith("k_"):
                continue
            fk, launches = f.get((name, "FETCH_SIZE"), (0.0, 0))
            fk *= 1024.0
            wk = w.get((name, "WRITE_SIZE"), (0.0, 0))[0] * 1024.0
            kernels[name] = {"fetch_size_bytes_raw": fk, "read_bytes_corrected": 2.0 * fk, "write_bytes": wk,
                             "hbm_bytes_per_launch": 2.0 * fk + wk, "launches_in_the_profiled_run": launches}
            rows.append((name, launches, fk, 2.0 * fk, wk, 2.0 * fk + wk))
        with open(os.path.join(dst, "%s_config3_hbm_traffic.csv" % tag), "w") as out:
            out.write("kernel,launches,FETCH_SIZE_bytes_raw,read_bytes_x2_gfx950,WRITE_SIZE_bytes,hbm_bytes_per_launch\n")
            for r in sorted(rows, key=lambda x: -x[5] * max(x[1], 1)):
                out.write("%s,%d,%.0f,%.0f,%.0f,%.0f\n" % r)
        tpath = os.path.join(dst, "hbm_traffic.json")
        tj = json.load(open(tpath)) if os.path.exists(tpath) else {}
        tj["config3"] = {"measured_at": when, "command": "python bench.py --config 3 --sites-fraction 0.125 (the shard of one of 8 GPUs)",
                         "kernels": kernels}
        json.dump(tj, open(tpath, "w"), indent=1, sort_keys=True)
    sq = {}
    for part in ("sq1", "sq2"):
        f = p("%s_%s" % (tag, part), "p_counter_collection.csv")
        if os.path.exists(f):
            sq.update(per_kernel(f))
    if sq:
        with open(os.path.join(dst, "%s_pmc_sq_chains.txt" % tag), "w") as out:
            out.write("rocprofv3 --pmc (two passes of SQ counters) of `python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline`\n"
                      "(64 976 hap x 2 M sites, 244 blocks), %s; counters summed over all waves of ONE launch.\n"
                      "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md).\n\n" % when)
            for kern in sorted({k[0] for k in sq}):
                if "chain" not in kern and "wah" not in kern:
                    continue
                out.write("%s\n" % kern)
                for (kk, c), (v, n) in sorted(sq.items()):
                    if kk == kern:
                        out.write("  %-22s %18.0f\n" % (c, v))
                out.write("\n")
    print("wrote", sorted(os.listdir(dst)))


if __name__ == "__main__":
    main()
