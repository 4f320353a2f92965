#!/bin/bash
# Dev: ablation of the dense-layer mainloops (needs the debug-variant library: SA_BUILD_VARIANT=dbg SA_EXTRA_HIPCC_FLAGS=-DSA_PP_DEBUG_VARIANTS python -m synthanatomy_amd.build).
# SA_PP_DBG bits: 8 no main loop, 16 no epilogue, 32 no MFMA / LDS reads, 64 no activation DMA, 128 no weight DMA.  Kernel durations from rocprofv3 (the Python
# launch path has a ~20 us floor).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export SA_HIP_LIB=$PWD/synthanatomy_amd/libsynthanatomy_hip_dbg.so
OUT=gpurun_out/abl; mkdir -p $OUT
for shape in ${SHAPES:-"w2 fwd"}; do
  for dbg in ${DBGS:-0 16 224 240 232 248}; do
    echo "== $shape SA_PP_DBG=$dbg"
    rm -rf $OUT/t
    DENSE_ONLY="${shape/_/ }" SA_PP_DBG=$dbg timeout 300 rocprofv3 --kernel-trace -d $OUT/t -o t -- python tools/bench_dense_tiles.py > $OUT/log.txt 2>&1
    python tools/rocpd_tools.py stats "$(find $OUT/t -name '*_results.db' | head -1)" --by-grid | grep -E "conv_fprop_dma" | cut -c1-150
  done
done
rm -rf $OUT
