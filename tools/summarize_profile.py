#!/usr/bin/env python3
"""Fold the output of tools/collect_profiles.sh (bench lines, rocprofv3 --stats, --pmc FETCH_SIZE, --pmc
WRITE_SIZE and two SQ counter passes, all CSV) into the files kept under profiles/.

usage: python tools/summarize_profile.py gpurun_out r02 profiles "commit abc1234, 2026-10-04"

HBM traffic follows MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are in KiB (x1024); on gfx950
FETCH_SIZE reports half of the bytes of a coalesced streaming read, so the read side is doubled (check:
k_count_rows, which streams the 16.25 GB input matrix once, reads 16.38 GB after the correction);
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
    fetch, write = p("%s_fetch" % tag, "p_counter_collection.csv"), p("%s_write" % tag, "p_counter_collection.csv")
    if os.path.exists(fetch) and os.path.exists(write):
        f, w = per_kernel(fetch), per_kernel(write)
        kernels = {}
        rows = []
        for name in sorted({k[0] for k in f} | {k[0] for k in w}):
            if not name.startswith("k_"):
                continue
            fk = f.get((name, "FETCH_SIZE"), (0.0, 0))[0] * 1024.0
            wk = w.get((name, "WRITE_SIZE"), (0.0, 0))[0] * 1024.0
            kernels[name] = {"fetch_size_bytes_raw": fk, "read_bytes_corrected": 2.0 * fk, "write_bytes": wk,
                             "hbm_bytes_per_launch": 2.0 * fk + wk}
            rows.append((name, fk, 2.0 * fk, wk, 2.0 * fk + wk))
        with open(os.path.join(dst, "%s_hbm_traffic.csv" % tag), "w") as out:
            out.write("kernel,FETCH_SIZE_bytes_raw,read_bytes_x2_gfx950,WRITE_SIZE_bytes,hbm_bytes_per_launch\n")
            for r in sorted(rows, key=lambda x: -x[4]):
                out.write("%s,%.0f,%.0f,%.0f,%.0f\n" % r)
        tpath = os.path.join(dst, "hbm_traffic.json")
        tj = json.load(open(tpath)) if os.path.exists(tpath) else {}
        if "config2" not in tj and "config1" not in tj:
            tj = {}  # round-1 layout
        tj["config2"] = {"measured_at": when, "command": "python bench.py (BASELINE configs[2])", "kernels": kernels}
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
