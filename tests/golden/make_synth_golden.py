#!/usr/bin/env python3
"""Golden manifest for seeded synthetic cases: SHA-256 + size of the .xsi the CPU oracle writes.

The oracle itself is pinned to the reference's recorded outputs (tests/test_oracle.py); this
manifest freezes its output on the synthetic generator so that (a) a change of the generator or
of the oracle is noticed on CPU, and (b) the GPU tests have size-independent anchors that do not
need the oracle at run time.  Run from the repo root: python tests/golden/make_synth_golden.py
"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import oracle  # noqa: E402
from xsqueezeit_amd import synth  # noqa: E402

CASES = [
    # name, seed, first_line, n_lines, n_haps, block_len, mac_thr
    ("h20_l64", 42, 0, 64, 20, 16, 0),
    ("h200_l700", 42, 0, 700, 200, 128, 0),
    ("h5008_l3000", 42, 0, 3000, 5008, 1024, 5),
    ("h5008_l9000_b8192", 42, 100000, 9000, 5008, 8192, 5),
    ("h16390_l200", 43, 0, 200, 16390, 64, 16),
    ("h64976_l48", 43, 0, 48, 64976, 16, 64),
    ("h131074_l16", 44, 0, 16, 131074, 8, 131),
]


def main():
    out = {}
    for name, seed, first, n_lines, n_haps, bl, thr in CASES:
        bits = synth.synth_bits(seed, first, n_lines, n_haps)
        w = oracle.Writer(n_haps // 2, bl, thr, 1)
        w.append_rows(synth.bits_to_gt(bits, 1), 2)
        data = w.finalize(2)
        out[name] = dict(seed=seed, first_line=first, n_lines=n_lines, n_haps=n_haps, block_len=bl, mac_thr=thr,
                         size=len(data), sha256=hashlib.sha256(data).hexdigest(),
                         bits_sha256=hashlib.sha256(bits.tobytes()).hexdigest())
        print(name, len(data))
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "synth_manifest.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
