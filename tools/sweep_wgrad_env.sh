#!/bin/bash
# Dev (round 6): sweep of the weight-gradient split tunables on the VQ-VAE step (the defaults date from before the weight gradients moved to the second stream).
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/sweep; mkdir -p $OUT
VQ="python bench.py --no-performer --no-extras --no-cpu-baseline --no-kernel-timer --steps 20 --warmup 6"
run() { env $1 $VQ 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('[$1]', d['value'], d['step_ms']['median'])"; }
for round in 1 2; do
  run "SA_X=0"
  for kv in SA_WGRAD_HALO_SPLITS=64 SA_WGRAD_HALO_SPLITS=96 SA_WGRAD_HALO_SPLITS=192 SA_WGRAD_HALO_SPLITS=256; do run $kv; done
  run "SA_X=0"
  for kv in SA_WGRAD_ROWS=5120 SA_WGRAD_ROWS=20480 SA_WGRAD_MIN_BLOCKS=512 SA_WGRAD_MIN_BLOCKS=2048; do run $kv; done
done 2>&1 | tee $OUT/wgrad_env_sweep.txt
