#!/bin/bash
# Dev: same-box A/B of the Performer training step between the product library and a variant (SA_BUILD_VARIANT=<name> python -m synthanatomy_amd.build), alternating.
# usage: tools/ab_perf.sh <variant> [rounds]
V=${1:-base}; R=${2:-2}
for i in $(seq $R); do
  for lib in "" "$PWD/synthanatomy_amd/libsynthanatomy_hip_$V.so"; do
    SA_HIP_LIB=$lib python bench.py --only-performer --no-sampling --no-kernel-timer --steps 30 --warmup 8 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('${lib:-product}'.split('/')[-1], d['value'], d['step_ms']['median'])"
  done
done
