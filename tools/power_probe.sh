#!/bin/bash
# Dev: socket power and clocks (rocm-smi) while one kernel family runs back to back.   usage (on the GPU box): tools/power_probe.sh [fwd|dgrad3|wgrad3|bwd1x1]
cd "${GRAFT_REPO_ROOT:-.}"
python -c "import torch; torch.zeros(1).cuda()" > /dev/null 2>&1     # (the first import on a fresh box takes a minute or two)
for k in ${1:-idle dgrad3 fwd wgrad3 bwd1x1}; do
  if [ "$k" != idle ]; then
    python tools/bench_resblock.py --only $k --iters 20000 > /dev/null 2>&1 &
    pid=$!
    sleep 12        # import + set-up + first launches
  fi
  echo "== $k"
  for i in 1 2 3; do
    rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk|fclk" | sed 's/^ *//' | tr '\n' ';'
    echo
    sleep 1
  done
  if [ "$k" != idle ]; then kill $pid 2>/dev/null; wait $pid 2>/dev/null; fi
done
rocm-smi --showmaxpower 2>/dev/null | grep -i power
