#!/bin/bash
# A/B of two builds of the library on ONE box: the GGX chain of a workload with each, alternating.
#   tools/ab_chain.sh c5 tools/variants/libibl_r02.so [steps]
WL=$1; OTHER=$2; STEPS=${3:-2}
for round in 1 2; do
  for lib in "" "$OTHER"; do
    IBL_HIP_LIB=$lib timeout -k 10 300 python bench.py --workload $WL --steps $STEPS --warmup 1 --no-cpu-baseline 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${lib:-current}', d['ms_per_step'], d['roofline']['kernel_ms'])" || exit 1
  done
done
