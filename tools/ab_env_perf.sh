#!/bin/bash
# Dev: same-box comparison of the Performer step under several environment settings, alternating.   usage: tools/ab_env_perf.sh "A=1" "B=2 C=3" ... (first = "" for the default)
for i in 1 2; do
  for kv in "$@"; do
    env $kv python bench.py --only-performer --no-sampling --no-kernel-timer --steps 30 --warmup 8 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('[$kv]', d['value'], d['step_ms']['median'])"
  done
done
