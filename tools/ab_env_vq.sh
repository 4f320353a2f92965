#!/bin/bash
# Dev: same-box comparison of the VQ-VAE step under several environment settings, alternating.   usage: tools/ab_env_vq.sh "A=1" "B=2 C=3" ...
for i in 1 2; do
  for kv in "$@"; do
    env $kv python bench.py --no-performer --no-extras --no-cpu-baseline --no-kernel-timer --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('[$kv]', d['value'], d['step_ms']['median'])"
  done
done
