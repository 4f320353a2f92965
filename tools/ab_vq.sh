#!/bin/bash
# Dev: same-box A/B of the VQ-VAE step between the product library and a variant (SA_BUILD_VARIANT=<name> python -m synthanatomy_amd.build), alternating runs.
# usage: tools/ab_vq.sh <variant> [rounds]
V=${1:-base}; R=${2:-2}
for i in $(seq $R); do
  for lib in "" "$PWD/synthanatomy_amd/libsynthanatomy_hip_$V.so"; do
    SA_HIP_LIB=$lib python bench.py --no-performer --no-extras --no-cpu-baseline --no-kernel-timer --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('${lib:-product}'.split('/')[-1], d['value'], d['step_ms']['median'])"
  done
done
