#!/bin/bash
# Round 6: side-stream weight gradients with the bounded operand window (SA_SIDE_WGRAD_LAG) -- adversarial leg, VQ-VAE step, Performer step.
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/lag; mkdir -p $OUT
VQ="python bench.py --no-performer --no-extras --no-cpu-baseline --no-kernel-timer --steps 20 --warmup 5"
PF="python bench.py --only-performer --no-sampling --no-kernel-timer --steps 30 --warmup 10"
for lag in 3 8 2 3; do
  export SA_SIDE_WGRAD_LAG=$lag
  python bench.py --only-adversarial 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('adv lag=$lag', d['value'], d['iteration_ms'], d['allocator'])"
  $VQ 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('vq lag=$lag', d['value'], d['step_ms']['median'], d['peak_mem_gb'], d['reserved_mem_gb'])"
  $PF 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('perf lag=$lag', d['value'], d['step_ms']['median'])"
done 2>&1 | tee $OUT/ab_lag.txt
