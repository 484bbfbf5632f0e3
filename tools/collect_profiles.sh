#!/bin/bash
# Collects the rocprofv3 evidence kept under profiles/ for the default bench line (BASELINE configs[2]):
# kernel-trace stats, HBM traffic (separate --pmc FETCH_SIZE / WRITE_SIZE passes, as the MI355X guide
# prescribes) and SQ instruction counters of the chain kernels.  Run on a GPU box:
#   gpurun -- bash tools/collect_profiles.sh r03
# Output lands in gpurun_out/<tag>_*; tools/summarize_profile.py folds it into profiles/.
set -e
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?not on a gpurun box}"
O=gpurun_out
mkdir -p "$O"
# the default line carries configs[1], the configs[3] shard and configs[4] as other_configs since round 4
python3 bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
# the 153-block x 500 000-haplotype shard one of 8 GPUs gets of BASELINE configs[3], under the profiler
rocprofv3 --kernel-trace --stats -d $O/${TAG}_c3stats -o p --output-format csv -- python3 bench.py --config 3 --sites-fraction 0.125 --steps 2 --warmup 1 --no-cpu-baseline > $O/${TAG}_config3_shard.json 2> $O/${TAG}_config3_shard.err
# ... its SQ counters (two passes) and the phase records of one wave of the long-row encode chain (wave 15 of workgroup 0)
C3="bench.py --config 3 --sites-fraction 0.125 --steps 1 --warmup 1 --no-cpu-baseline"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $O/${TAG}_c3sq1 -o p --output-format csv -- python3 $C3 > /dev/null 2> $O/${TAG}_c3sq1.err
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SMEM --kernel-trace -d $O/${TAG}_c3sq2 -o p --output-format csv -- python3 $C3 > /dev/null 2> $O/${TAG}_c3sq2.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/${TAG}_c3fetch -o p --output-format csv -- python3 $C3 > /dev/null 2> $O/${TAG}_c3fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/${TAG}_c3write -o p --output-format csv -- python3 $C3 > /dev/null 2> $O/${TAG}_c3write.err
XSI_ENABLE_TUNING_ENV=1 XSI_MULTI_PROF=983041 python3 $C3 > /dev/null 2> $O/${TAG}_config3_phase_clocks.err
grep "xsi multi prof" $O/${TAG}_config3_phase_clocks.err > $O/${TAG}_config3_phase_clocks.txt || true
rocprofv3 --kernel-trace --stats -d $O/${TAG}_stats -o p --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs > $O/${TAG}_stats.json 2> $O/${TAG}_stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/${TAG}_fetch -o p --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs > /dev/null 2> $O/${TAG}_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/${TAG}_write -o p --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs > /dev/null 2> $O/${TAG}_write.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $O/${TAG}_sq1 -o p --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs > /dev/null 2> $O/${TAG}_sq1.err
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SMEM --kernel-trace -d $O/${TAG}_sq2 -o p --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs > /dev/null 2> $O/${TAG}_sq2.err
ls $O/${TAG}_stats $O/${TAG}_fetch $O/${TAG}_sq1
