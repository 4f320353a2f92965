#!/bin/bash
# Dev: duration of the quantizer launches inside the VQ-VAE step (rocprofv3 kernel trace of two steps).
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/vq; mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace -d $OUT/kt -o kt -- python bench.py --no-performer --no-cpu-baseline --no-extras --no-kernel-timer --steps 2 --warmup 1 > $OUT/kt.log 2>&1
python tools/rocpd_tools.py stats "$(find $OUT/kt -name '*_results.db' | head -1)" --by-grid | grep -i "vq_\|kernel  " | tee $OUT/vq_kernels.txt
rm -rf $OUT/kt
