#!/bin/bash
# File-level boundary rates from the C program (tests/c/boundary_roundtrip.c) at 5008 and 64 976 haplotypes.
# Run on a GPU box: gpurun -- bash tools/boundary_rates.sh <out.txt>
set -e
cd "${GRAFT_REPO_ROOT:?not on a gpurun box}"
OUT=${1:-gpurun_out/boundary_rates.txt}
mkdir -p "$(dirname "$OUT")"
gcc -std=c99 -O2 -I include tests/c/boundary_roundtrip.c -o /tmp/boundary_roundtrip -L xsqueezeit_amd -lxsi_hip -Wl,-rpath,"$PWD/xsqueezeit_amd"
{
  echo "# long streams, default writer batches (several batches: the encode of one runs under the appends of the next)"
  /tmp/boundary_roundtrip /tmp/b1.xsi 2504 600000 8192
  /tmp/boundary_roundtrip /tmp/b2.xsi 32488 49152 8192
  echo "# short files (one or two batches: the last batch's encode is not hidden)"
  /tmp/boundary_roundtrip /tmp/b1.xsi 2504 100000 8192
  /tmp/boundary_roundtrip /tmp/b2.xsi 32488 12000 8192
  echo "# XSI_WRITER_NO_PACK=1: every line the int32 way (what round 2 did)"
  XSI_ENABLE_TUNING_ENV=1 XSI_WRITER_NO_PACK=1 /tmp/boundary_roundtrip /tmp/b3.xsi 2504 600000 8192
  echo "# the pack loop alone (tools/pack_rate.c): xsi_debug_pack_bit_row over rows in DRAM, one thread"
  gcc -O2 -I include tools/pack_rate.c -o /tmp/pack_rate -L xsqueezeit_amd -lxsi_hip -Wl,-rpath,"$PWD/xsqueezeit_amd"
  /tmp/pack_rate 5008 300000 | tail -1
  /tmp/pack_rate 64976 24000 | tail -1
} > "$OUT" 2>&1
cat "$OUT"
