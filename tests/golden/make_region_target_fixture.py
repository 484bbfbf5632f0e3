#!/usr/bin/env python3
"""Re-express the reference's binary BCF test fixture test/test_files/test_region_target.bcf as a GT-only VCF.

Run in the build container only (reads /root/reference/test/test_files/test_region_target.bcf, a DATA file of the
reference's own integration tests: test/cukinia_v4.conf:19 runs verify_v4.sh on it with -t chr17:117980-117999).
The file is 6 records x 3202 samples of phased 1000 Genomes high-coverage data (6404 haplotypes: 320 x the micro
VCFs).  htslib is not in this image, so the container is walked by hand:

  BGZF   = concatenated gzip members (python's gzip reads them as one stream);
  BCF2.2 = "BCF\\2\\2", l_text, header text, then per record
           l_shared, l_indiv, CHROM, POS (0-based), rlen, QUAL, n_info | n_allele << 16, n_sample | n_fmt << 24,
           typed ID, n_allele typed allele strings, typed FILTER vector, n_info (typed key, typed value) pairs,
           then per FORMAT field: typed key, type byte (count << 4 | type), n_sample x count values.
  GT     = one int8/int16/int32 vector per record, values (allele + 1) << 1 | phased, exactly what bcf_get_genotypes
           widens to int32 for the reference's encoder (bcf_traversal.cpp:3-16).

Output (tests/golden/region_target.vcf): "#CHROM" line with the sample names, per record CHROM POS ID REF ALT and the
GT columns as text; INFO and the ~100 meta lines, which the genotype path never reads, are dropped - the same
reduction make_micro_fixtures.py applies to the micro VCFs.
"""
import gzip
import os
import struct

SRC = "/root/reference/test/test_files/test_region_target.bcf"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "region_target.vcf")

TYPE_SIZE = {1: 1, 2: 2, 3: 4, 5: 4, 7: 1}
TYPE_FMT = {1: "b", 2: "h", 3: "i", 5: "f", 7: "c"}


class Cursor:
    def __init__(self, buf, pos=0):
        self.b, self.p = buf, pos

    def take(self, fmt):
        v = struct.unpack_from("<" + fmt, self.b, self.p)
        self.p += struct.calcsize("<" + fmt)
        return v if len(v) > 1 else v[0]

    def typed_desc(self):
        """(count, type) of a typed value; count 15 means "a typed int follows with the real count"."""
        d = self.take("B")
        n, t = d >> 4, d & 15
        if n == 15:
            n = self.typed_int()
        return n, t

    def typed_int(self):
        n, t = self.typed_desc()
        assert n == 1 and t in (1, 2, 3)
        return self.take(TYPE_FMT[t])

    def typed_vector(self):
        n, t = self.typed_desc()
        if t == 0 or n == 0:
            return t, []
        if t == 7:
            s = self.b[self.p:self.p + n]
            self.p += n
            return t, s.decode()
        v = struct.unpack_from("<%d%s" % (n, TYPE_FMT[t]), self.b, self.p)
        self.p += n * TYPE_SIZE[t]
        return t, list(v)


def header_dictionary(text):
    """IDX of every FILTER / INFO / FORMAT id and the contig list, as htslib builds them (IDX= when present)."""
    ids, contigs = {}, []
    nxt = 0
    for ln in text.split("\n"):
        if ln.startswith("##contig=<"):
            contigs.append(ln.split("ID=", 1)[1].split(",")[0].rstrip(">"))
        for kind in ("FILTER", "INFO", "FORMAT"):
            if ln.startswith("##%s=<" % kind):
                name = ln.split("ID=", 1)[1].split(",")[0].rstrip(">")
                if "IDX=" in ln:
                    idx = int(ln.rsplit("IDX=", 1)[1].rstrip(">"))
                else:
                    idx = ids.get(name, nxt)
                ids[name] = idx
                nxt = max(nxt, idx + 1)
    return ids, contigs


def main():
    data = gzip.open(SRC, "rb").read()
    assert data[:5] == b"BCF\x02\x02"
    l_text = struct.unpack_from("<I", data, 5)[0]
    text = data[9:9 + l_text].rstrip(b"\0").decode()
    ids, contigs = header_dictionary(text)
    gt_key = ids["GT"]
    chrom_line = [ln for ln in text.split("\n") if ln.startswith("#CHROM")][0]
    samples = chrom_line.split("\t")[9:]
    out = ["##fileformat=VCFv4.2", '##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">', chrom_line]
    c = Cursor(data, 9 + l_text)
    n_rec = 0
    while c.p < len(data):
        l_shared, l_indiv = c.take("II")
        end_shared = c.p + l_shared
        chrom, pos, _rlen, _qual, n_allele_info, n_fmt_sample = c.take("iiifII")
        n_allele, n_info = n_allele_info >> 16, n_allele_info & 0xFFFF
        n_fmt, n_sample = n_fmt_sample >> 24, n_fmt_sample & 0xFFFFFF
        assert n_sample == len(samples)
        _, rec_id = c.typed_vector()
        alleles = [c.typed_vector()[1] for _ in range(n_allele)]
        c.p = end_shared  # FILTER and INFO are not needed
        end_indiv = c.p + l_indiv
        cols = None
        for _ in range(n_fmt):
            key = c.typed_int()
            cnt, t = c.typed_desc()
            size = TYPE_SIZE[t]
            if key == gt_key:
                assert t in (1, 2, 3)
                vals = struct.unpack_from("<%d%s" % (cnt * n_sample, TYPE_FMT[t]), data, c.p)
                eov = {1: -127, 2: -32767, 3: -(2 ** 31) + 1}[t]  # bcf_int*_vector_end
                cols = []
                for s in range(n_sample):
                    v = vals[s * cnt:(s + 1) * cnt]
                    txt = ""
                    for j, x in enumerate(v):
                        if x == eov:
                            break
                        if j:
                            txt += "|" if x & 1 else "/"
                        txt += "." if (x >> 1) == 0 else str((x >> 1) - 1)
                    cols.append(txt)
            c.p += cnt * n_sample * size
        assert c.p == end_indiv and cols is not None
        out.append("\t".join([contigs[chrom], str(pos + 1), rec_id if rec_id else ".", alleles[0],
                              ",".join(alleles[1:]) if n_allele > 1 else ".", ".", ".", ".", "GT"] + cols))
        n_rec += 1
    with open(DST, "w") as f:
        f.write("\n".join(out) + "\n")
    print("%s: %d records x %d samples" % (os.path.basename(DST), n_rec, len(samples)))


if __name__ == "__main__":
    main()
