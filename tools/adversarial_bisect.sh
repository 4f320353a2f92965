#!/bin/bash
# Round 6: locate the adversarial-iteration regression (168 -> 1 431 ms in round 5).  Run through gpurun from the repo root.
# Kernel trace of the leg + the leg under each developer switch; output under gpurun_out/adv/.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/adv; mkdir -p $OUT
ADV="python bench.py --only-adversarial"
db() { find "$1" -name "*_results.db" | head -1; }
$ADV > $OUT/base.json 2> $OUT/base.err
timeout 600 rocprofv3 --kernel-trace -d $OUT/kt -o kt -- $ADV > $OUT/kt.log 2>&1
python tools/rocpd_tools.py stats "$(db $OUT/kt)" --by-grid > $OUT/adversarial_kernel_stats_by_grid.txt 2>&1
rm -rf $OUT/kt
for kv in SA_NO_SIDE_WGRAD_VQVAE=1 SA_NO_CELLS256=1 SA_NO_CLASS_LAUNCH=1; do
  env $kv $ADV > $OUT/${kv%%=*}.json 2> $OUT/${kv%%=*}.err
done
tail -n 3 $OUT/*.json
head -40 $OUT/adversarial_kernel_stats_by_grid.txt
