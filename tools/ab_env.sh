#!/bin/bash
# Dev: same-box A/B of the VQ-VAE step with / without one environment switch, alternating.   usage: tools/ab_env.sh VAR=VALUE [rounds]
KV=$1; R=${2:-3}
for i in $(seq $R); do
  for v in 0 1; do
    if [ $v = 1 ]; then export "$KV"; else unset "${KV%%=*}"; fi
    python bench.py --no-performer --no-extras --no-cpu-baseline --no-kernel-timer --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('$KV on=$v', d['value'], d['step_ms']['median'])"
  done
done
