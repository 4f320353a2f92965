#!/bin/bash
# Dev: ablation of the im2col-order mainloop on the stride-2 layers of config 2 at batch 8 (debug-variant library, see tools/dense_ablate.sh): SA_PP_DBG bits
# 32 no MFMA / LDS reads, 64 no activation DMA, 128 no weight DMA.
export SA_HIP_LIB=$PWD/synthanatomy_amd/libsynthanatomy_hip_dbg.so
for dbg in 0 32 64 128 192 224; do
  echo "== SA_PP_DBG=$dbg"
  SA_PP_DBG=$dbg python tools/microbench.py --batch 8 --only "k4s2 128" 2>&1 | grep -v amdgpu.ids
  SA_PP_DBG=$dbg python tools/microbench.py --batch 8 --only "up convT" 2>&1 | grep -v amdgpu.ids
done
