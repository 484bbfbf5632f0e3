"""GT-only VCF text reader that reproduces htslib's int32 genotype encoding.

htslib is not available in this image, so real BCF I/O is out of reach; this
small reader exists so the reference's own micro VCF fixtures can drive the genotype-block
path.  It yields exactly what ``bcf_get_genotypes`` hands the reference's encoder
(bcf_traversal.cpp:3-16): per record an int32 row of ``n_samples * max_ploidy`` values with
``(allele+1)<<1 | phased``, ``.`` -> 0 | phased, short samples padded with
``bcf_int32_vector_end``; the first allele of a sample never carries the phase bit.
"""
import numpy as np

INT32_MISSING = -(2 ** 31)
INT32_VECTOR_END = -(2 ** 31) + 1


def parse_gt_field(s):
    """One sample's GT string -> list of int32 values (no padding)."""
    vals = []
    phased = 0
    tok = ""
    for ch in s + "\0":
        if ch in "/|\0":
            if tok == "." or tok == "":
                vals.append(0 | phased)
            else:
                vals.append(((int(tok) + 1) << 1) | phased)
            tok = ""
            phased = 1 if ch == "|" else 0
        else:
            tok += ch
    return vals


def read_vcf(path):
    """Returns (sample_names, records); records = list of dict(chrom,pos,id,ref,alt,n_allele,gt)."""
    samples = []
    records = []
    with open(path) as f:
        for line in f:
            line = line.rstrip("\n")
            if not line or line.startswith("##"):
                continue
            t = line.split("\t")
            if line.startswith("#CHROM"):
                samples = t[9:]
                continue
            fmt = t[8].split(":")
            gi = fmt.index("GT")
            per = [parse_gt_field(x.split(":")[gi]) for x in t[9:]]
            ploidy = max(len(p) for p in per)
            gt = np.full(len(per) * ploidy, INT32_VECTOR_END, dtype=np.int32)
            for i, p in enumerate(per):
                gt[i * ploidy:i * ploidy + len(p)] = p
            alts = [] if t[4] == "." else t[4].split(",")
            records.append(dict(chrom=t[0], pos=int(t[1]), id=t[2], ref=t[3], alt=alts,
                                n_allele=1 + len(alts), gt=gt))
    return samples, records
