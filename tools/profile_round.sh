#!/bin/bash
# Collects the rocprofv3 evidence for one bench workload on the GPU box:
#   tools/profile_round.sh c4|c3 <tag> [extra bench.py arguments]
# One --kernel-trace --stats run, then separate --pmc passes (instruction mix, FETCH_SIZE,
# WRITE_SIZE, matrix pipe), each with its own time limit; results land in gpurun_out/prof/<tag>*.
set -e
WL=$1; TAG=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
STEPS=${STEPS:-3}; WARMUP=${WARMUP:-1}     # (short launches need more of them to reach the steady clock)
ARGS="$ROOT/bench.py --workload $WL --steps $STEPS --warmup $WARMUP --no-cpu-baseline $*"
timeout -s KILL 240 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}stats -o r -- python3 $ARGS > $OUT/${TAG}stats.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $OUT/${TAG}pmc1 -o r -- python3 $ARGS > $OUT/${TAG}pmc1.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE -d $OUT/${TAG}pmc2 -o r -- python3 $ARGS > $OUT/${TAG}pmc2.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE -d $OUT/${TAG}pmc3 -o r -- python3 $ARGS > $OUT/${TAG}pmc3.log 2>&1
# matrix pipe and memory instruction counts (the GGX kernels' matrix form)
timeout -s KILL 240 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM -d $OUT/${TAG}pmc4 -o r -- python3 $ARGS > $OUT/${TAG}pmc4.log 2>&1 || echo "pmc4 pass failed"
grep -h '"metric"' $OUT/${TAG}stats.log | tail -1
