#!/usr/bin/env python3
"""Looks at the gfx950 ISA of every kernel of the library for three things that cost real time in round 4 and that no
counter points at:

  serial loads   runs of  <memory load> ... s_waitcnt vmcnt(0) ... <memory load> ...: loads the source issues together but
                 the machine scheduler (saving registers) interleaved with their uses, one round trip at a time
                 (k_chain_rank_enc_multi's table copy: eight L2 round trips per line instead of one);
  serial gathers the same with ds_read / s_waitcnt lgkmcnt(0) (k_chain_decode_rank_big<64, 16>: one LDS gather in flight
                 instead of eight - 179 -> 157 ms of decode at the configs[3] shard);
  FLAT ops       flat_load / flat_store: a pointer the compiler cannot place in an address space (rebuilt from two
                 v_readlane halves, selected between two pointers, ...).  A FLAT operation counts in vmcnt AND lgkmcnt
                 and may return out of order, so while one is outstanding every wait the compiler inserts is a wait for
                 ALL loads (the slice flag store of k_chain_rank_enc_multi made every wait of its table copy vmcnt(0)).

Runs of dependent metadata loads (line id -> block -> offset) show up as serial loads too: read the hit before acting.
The cure for the first two is __builtin_amdgcn_sched_barrier(0) between the loads and their uses, for the third a cast to
an address_space(1) pointer.

usage: python tools/isa_scan.py [file.hip ...]        (default: every .hip under xsqueezeit_amd/csrc; hipcc needed,
                                                       no GPU; about half a minute per file)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "xsqueezeit_amd", "csrc")


def isa_of(src):
    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-I", CSRC, src, "-o", out]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    text = open(out).read()
    os.unlink(out)
    return text


def kernels(text):
    idx = [(m.start(), m.group(1)) for m in re.finditer(r"\n(_Z\w+):", text)]
    for i, (st, name) in enumerate(idx):
        en = idx[i + 1][0] if i + 1 < len(idx) else len(text)
        yield name, text[st:en].split("\n")


def longest_alternation(lines, is_load, is_wait, gap):
    ev = []
    for n, l in enumerate(lines):
        t = l.strip()
        if is_load(t):
            ev.append((n, "L"))
        elif is_wait(t):
            ev.append((n, "W"))
    best = run = 0
    last, lastn = None, 0
    for n, k in ev:
        run = run + 1 if (last is None or (k != last and n - lastn < gap)) else 1
        best = max(best, run)
        last, lastn = k, n
    return best


def demangle(name):
    r = subprocess.run(["c++filt", name], stdout=subprocess.PIPE, text=True).stdout.strip()
    return re.sub(r"\(.*", "", r).replace("void xsi::", "").replace("xsi::", "")


def main():
    files = sys.argv[1:] or sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    for f in files:
        text = isa_of(f)
        for name, lines in kernels(text):
            body = "\n".join(lines)
            if "s_endpgm" not in body:
                continue
            flat = sorted(set(re.findall(r"\n\s+(flat_\w+)", body)))
            vm = longest_alternation(lines, lambda t: re.match(r"(global_load|buffer_load|flat_load)", t) is not None,
                                     lambda t: t.startswith("s_waitcnt") and "vmcnt(0)" in t, 15)
            ds = longest_alternation(lines, lambda t: t.startswith("ds_read"),
                                     lambda t: t.startswith("s_waitcnt") and "lgkmcnt(0)" in t, 20)
            notes = []
            if flat:
                notes.append("FLAT: " + " ".join(flat))
            if vm >= 6:
                notes.append("serial loads: run of %d" % vm)
            if ds >= 8:
                notes.append("serial gathers: run of %d" % ds)
            if notes:
                print("%-22s %-64s %s" % (os.path.basename(f), demangle(name)[:64], "; ".join(notes)))


if __name__ == "__main__":
    main()
