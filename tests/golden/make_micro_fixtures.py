#!/usr/bin/env python3
"""Re-express the reference's 7 micro VCF test fixtures as small GT-only VCFs.

Run in the build container only (reads /root/reference/test/test_files/micro_*.vcf, which are
DATA files the reference's own integration tests use: test/cukinia_v4.conf:4-10).  Keeps the
#CHROM line and, per record, CHROM POS ID REF ALT and the GT sample columns verbatim; drops
the ~250 '##' meta lines and the INFO column, which the genotype path never reads.
"""
import glob
import os

SRC = "/root/reference/test/test_files"
DST = os.path.dirname(os.path.abspath(__file__))

for path in sorted(glob.glob(os.path.join(SRC, "micro_*.vcf"))):
    out = ["##fileformat=VCFv4.1", '##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">']
    with open(path) as f:
        for line in f:
            line = line.rstrip("\n")
            if line.startswith("##") or not line:
                continue
            t = line.split("\t")
            if line.startswith("#CHROM"):
                out.append(line)
                continue
            assert t[8] == "GT", path
            out.append("\t".join(t[:5] + [".", ".", ".", "GT"] + t[9:]))
    with open(os.path.join(DST, os.path.basename(path)), "w") as f:
        f.write("\n".join(out) + "\n")
    print(os.path.basename(path), len(out) - 3, "records")
