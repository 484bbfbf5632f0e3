#!/usr/bin/env python3
"""One line per gfx950 kernel of a .hip source: VGPRs, SGPRs, spills, scratch bytes (hipcc
-Rpass-analysis=kernel-resource-usage).  Usage: tools/kernel_resources.py xsi_pair.hip [filter]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "xsqueezeit_amd", "csrc")


def main():
    src = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    if not os.path.exists(src):
        src = os.path.join(CSRC, src)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
           "-Rpass-analysis=kernel-resource-usage", "-I", CSRC, src, "-o", "/dev/null"]
    out = subprocess.run(cmd, stderr=subprocess.PIPE, stdout=subprocess.PIPE, text=True).stderr
    cur = None
    rows = []
    for line in out.splitlines():
        m = re.search(r"remark: [^:]*:\d+:\d+: (.*) \[-Rpass", line) or re.search(r"remark: (.*) \[-Rpass", line)
        if not m:
            m = re.search(r":\d+:\d+: +(.*) \[-Rpass", line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:") or t.startswith("Name:"):
            cur = {"name": t.split(":", 1)[1].strip()}
            rows.append(cur)
        elif cur is not None and ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip()
    for r in rows:
        name = subprocess.run(["c++filt", r["name"]], stdout=subprocess.PIPE, text=True).stdout.strip()
        name = re.sub(r"\(.*", "", name).replace("void xsi::", "")
        if flt and flt not in name:
            continue
        print("%-48s vgpr %3s sgpr %3s  spill v%s s%s  scratch %s  occ %s" % (
            name, r.get("VGPRs"), r.get("TotalSGPRs"), r.get("VGPRs Spill"), r.get("SGPRs Spill"),
            r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]")))


if __name__ == "__main__":
    main()
