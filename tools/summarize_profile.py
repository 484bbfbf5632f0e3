#!/usr/bin/env python3
"""Fold a gpurun profile directory (bench.json, rocprofv3 --stats, --pmc FETCH_SIZE, --pmc WRITE_SIZE
passes) into the files kept under profiles/.

usage: python tools/summarize_profile.py gpurun_out/r01 profiles r01

HBM traffic follows MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are in KiB (x1024); on
gfx950 FETCH_SIZE reports half of the bytes of a coalesced streaming read, so the read side is
doubled; WRITE_SIZE is taken as is.  The two counters come from separate passes.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def per_kernel(counter_csv):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(counter_csv)):
        a = agg[r["Kernel_Name"]]
        a[0] += float(r["Counter_Value"])
        a[1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in agg.items()}


def newest(pattern):
    fs = sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)
    return fs[-1:]


def main():
    src, dst, tag = sys.argv[1], sys.argv[2], sys.argv[3]
    os.makedirs(dst, exist_ok=True)
    if os.path.exists(os.path.join(src, "bench.json")):
        shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, "%s_bench.json" % tag))
    stats = newest(os.path.join(src, "stats", "**", "*_kernel_stats.csv"))
    if stats:
        shutil.copy(stats[0], os.path.join(dst, "%s_kernel_stats.csv" % tag))
    fetch = newest(os.path.join(src, "fetch", "**", "*_counter_collection.csv"))
    write = newest(os.path.join(src, "write", "**", "*_counter_collection.csv"))
    traffic = {}
    if fetch and write:
        f = per_kernel(fetch[0])
        w = per_kernel(write[0])
        rows = []
        for k in sorted(set(f) | set(w)):
            if not k.startswith(("xsi::", "void xsi::")):
                continue
            fk = f.get(k, (0.0, 0))[0] * 1024.0
            wk = w.get(k, (0.0, 0))[0] * 1024.0
            short = k.split("(")[0].replace("void ", "")
            traffic[short] = {"fetch_size_bytes_raw": fk, "read_bytes_corrected": 2.0 * fk, "write_bytes": wk,
                              "hbm_bytes_per_launch": 2.0 * fk + wk, "launches": f.get(k, (0, 0))[1]}
            rows.append((short, fk, 2.0 * fk, wk, 2.0 * fk + wk))
        with open(os.path.join(dst, "%s_hbm_traffic.csv" % tag), "w") as out:
            out.write("kernel,FETCH_SIZE_bytes_raw,read_bytes_x2_gfx950,WRITE_SIZE_bytes,hbm_bytes_per_launch\n")
            for r in sorted(rows, key=lambda x: -x[4]):
                out.write("%s,%.0f,%.0f,%.0f,%.0f\n" % r)
        json.dump(traffic, open(os.path.join(dst, "hbm_traffic.json"), "w"), indent=1, sort_keys=True)
    print("wrote", sorted(os.listdir(dst)))


if __name__ == "__main__":
    main()
