#!/bin/bash
# Round evidence on the GPU box (run through gpurun from the repo root): rocprofv3 kernel stats (per kernel and per launch size), HBM-side traffic (two PMC
# passes), SQ passes; summaries land in gpurun_out/prof/ -- copy them to profiles/rNN_* afterwards.   usage: tools/profile_round.sh vqvae|performer|sampling
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof; mkdir -p $OUT
export SA_NO_SIDE_WGRAD_VQVAE=1   # one stream: overlapped weight gradients inflate (and reorder) every per-kernel duration; the timed bench runs with both streams
VQ="python bench.py --no-performer --no-cpu-baseline --no-extras --no-kernel-timer --steps 2 --warmup 1"
db() { find "$1" -name "*_results.db" | head -1; }
case "${1:-vqvae}" in
vqvae)
  timeout 600 rocprofv3 --kernel-trace -d $OUT/kt -o kt -- $VQ > $OUT/kt.log 2>&1
  python tools/rocpd_tools.py stats "$(db $OUT/kt)" > $OUT/vqvae_train_b8_kernel_stats.txt 2>&1
  python tools/rocpd_tools.py stats "$(db $OUT/kt)" --by-grid > $OUT/vqvae_train_b8_kernel_stats_by_grid.txt 2>&1
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pf -o pf -- $VQ > $OUT/pf.log 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pw -o pw -- $VQ > $OUT/pw.log 2>&1
  python tools/rocpd_tools.py traffic "$(db $OUT/pf)" "$(db $OUT/pw)" $OUT/pmc_traffic.json "round 6: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate passes (--kernel-trace only) of: $VQ; hbm_bytes_per_launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB * 1024 (gfx950 correction of MI355X_MICROARCH.md)" --by-grid > $OUT/vqvae_pmc_hbm.txt 2>&1
  timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace -d $OUT/sq -o sq -- $VQ > $OUT/sq.log 2>&1
  python tools/rocpd_tools.py pmc "$(db $OUT/sq)" --by-grid > $OUT/vqvae_pmc_sq.txt 2>&1
  timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace -d $OUT/lds -o lds -- $VQ > $OUT/lds.log 2>&1
  python tools/rocpd_tools.py pmc "$(db $OUT/lds)" --by-grid > $OUT/vqvae_pmc_lds.txt 2>&1
  ;;
performer)
  PF="python bench.py --only-performer --no-sampling --no-kernel-timer --steps 3 --warmup 1"
  timeout 600 rocprofv3 --kernel-trace -d $OUT/pkt -o pkt -- $PF > $OUT/pkt.log 2>&1
  python tools/rocpd_tools.py stats "$(db $OUT/pkt)" > $OUT/performer_train_kernel_stats.txt 2>&1
  SA_NO_SIDE_WGRAD=1 timeout 600 rocprofv3 --kernel-trace -d $OUT/pkt1 -o pkt1 -- $PF > $OUT/pkt1.log 2>&1
  python tools/rocpd_tools.py stats "$(db $OUT/pkt1)" > $OUT/performer_train_one_stream_kernel_stats.txt 2>&1
  SA_NO_SIDE_WGRAD=1 timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d $OUT/plds -o plds -- $PF > $OUT/plds.log 2>&1
  python tools/rocpd_tools.py pmc "$(db $OUT/plds)" > $OUT/performer_pmc_lds.txt 2>&1
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/ppf -o ppf -- $PF > $OUT/ppf.log 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/ppw -o ppw -- $PF > $OUT/ppw.log 2>&1
  python tools/rocpd_tools.py traffic "$(db $OUT/ppf)" "$(db $OUT/ppw)" $OUT/pmc_traffic_performer.json "round 6: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate passes (--kernel-trace only) of: $PF; hbm_bytes_per_launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB * 1024 (gfx950 correction of MI355X_MICROARCH.md)" --sources=performer > $OUT/performer_pmc_hbm.txt 2>&1
  timeout 300 rocprofv3 --kernel-trace -d $OUT/dense -o dense -- python tools/bench_dense_tiles.py > $OUT/dense.log 2>&1
  python tools/rocpd_tools.py stats "$(db $OUT/dense)" --by-grid > $OUT/dense_layers_by_grid.txt 2>&1
  ;;
sampling)
  timeout 600 rocprofv3 --kernel-trace -d $OUT/skt -o skt -- python bench.py --only-performer --steps 1 --warmup 1 --no-kernel-timer > $OUT/skt.log 2>&1
  python tools/rocpd_tools.py stats "$(db $OUT/skt)" > $OUT/performer_sampling_kernel_stats.txt 2>&1
  ;;
esac
rm -rf $OUT/kt $OUT/pf $OUT/pw $OUT/sq $OUT/lds $OUT/pkt $OUT/pkt1 $OUT/plds $OUT/ppf $OUT/ppw $OUT/dense $OUT/skt
ls -la $OUT
