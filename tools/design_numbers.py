#!/usr/bin/env python3
"""Regenerates the "Numbers of record" table of DESIGN.md (between the numbers:begin / numbers:end markers) from the
bench lines kept under profiles/: usage  python tools/design_numbers.py profiles/r04_bench.json [profiles/<config 4 line>.json]
The default line carries configs[1], the configs[3] shard and configs[4] as other_configs; a separate config-4 line
(more blocks / queries) overrides that entry when given."""
import json
import re
import sys

ROOT = __file__.rsplit("/", 2)[0]


def main():
    o = json.load(open(sys.argv[1]))
    oc = o.get("other_configs", {})
    c1 = oc.get("configs[1]", {})
    c3 = oc.get("configs[3] shard (1 of 8 GPUs)", {})
    c4 = json.load(open(sys.argv[2])) if len(sys.argv) > 2 else oc.get("configs[4]", {})

    def row(x, kernel_fmt):
        r = x.get("roofline", {})
        return "| %s | %s | **%.2f** | %.2f T | %s | %.1f | **%.3f** (whole round trip %.3f) |"

    def rt(name, x, kname):
        r = x["roofline"]
        return "| %s | **%.2f** | %.2f T | `%s` (encode chain %.1f, decode chain %.1f) | %.1f | **%.3f** (whole round trip %.3f) |" % (
            name, x["ms_per_step"], x["value"] / 1e12, kname, r["chain_encode_ms"], r["chain_decode_ms"] * r["chain_launches_per_step"]["decode"] /
            max(r["chain_launches_per_step"]["encode"], 1), r["kernel_ms"], r["frac"], r["pipeline_frac"])

    lines = ["| workload | ms per step | GT cells/s | dominant kernel (chains, ms per step) | its ms per launch | algorithmic GB/s ÷ 8000 |",
             "|---|---|---|---|---|---|"]
    lines.append(rt("configs[2]: 64 976 hap × 2 M sites, encode + decode, **row count inside the step**", o, o["roofline"]["kernel"].split(" ")[0]))
    if c1.get("roofline"):
        lines.append(rt("configs[1]: 5008 hap × 1 M sites", c1, c1["roofline"]["kernel"].split(" ")[0]))
    if c3.get("roofline"):
        lines.append(rt("configs[3] shard of one of 8 GPUs: 500 000 hap × 153 blocks (78 GB in, 78 GB back)", c3, c3["roofline"]["kernel"].split(" ")[0]))
    if c4.get("config"):
        lines.append("| configs[4]: 200 000 hap, mixed ploidy + tri-allelic, random access, one line per `get_genotypes` call | %.1f µs per query | %.1f G | "
                     "`k_compose_gt` + PCIe | — | PCIe-bound: %.1f GB/s of int32 rows into host memory |" % (
                         c4["config"]["us_per_query"], c4["value"] / 1e9, c4["roofline"]["achieved"]))
        b = c4.get("batched")
        if b:
            lines.append("| configs[4], the random queries through `xsi_accessor_get_genotypes_batch` | %.1f µs per query | %.1f G | same | — | "
                         "PCIe-bound: %.1f GB/s |" % (b["us_per_query"], b["value"] / 1e9, b["host_GBps"]))
        ci = c4.get("cold_isolated_queries")
        if ci:
            qs = ci["queries"]
            lines.append("| configs[4], cold isolated query at offset %s of a %d-line block | %s ms | | prefix decode (§9) | whole block: %.1f | |" % (
                " / ".join(str(q["bm_offset"]) for q in qs), ci["block_lines"], " / ".join("%.2f" % q["prefix_ms"] for q in qs),
                max(q["full_ms"] for q in qs)))
    cb = o.get("cpu_baseline") or {}
    tail = ""
    if cb:
        ac = cb.get("all_cores", {})
        tail = ("\nCPU oracle beside configs[2] (same box, %s): %.2f G cells/s on one core%s — a stated baseline, not the target."
                % (cb.get("cpu_model", "?"), cb["value"] / 1e9, (", %.2f G on %d threads" % (ac["value"] / 1e9, ac["cores"])) if ac else ""))
    block = "<!-- numbers:begin (tools/design_numbers.py) -->\n" + "\n".join(lines) + "\n" + tail + "\n<!-- numbers:end -->"
    p = ROOT + "/DESIGN.md"
    s = open(p).read()
    s2 = re.sub(r"<!-- numbers:begin.*?<!-- numbers:end -->", lambda m: block, s, flags=re.S)
    assert s2 != s or block in s, "markers not found"
    open(p, "w").write(s2)
    print(block)


if __name__ == "__main__":
    main()
