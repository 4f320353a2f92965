#!/bin/bash
# Copies the summaries tools/profile_round.sh left in gpurun_out/prof/ into profiles/ (tracked), named per round, each with a header naming the commit and command.
# usage: tools/collect_profiles.sh <round tag, e.g. r05>
R=${1:?round tag}; P=gpurun_out/prof; C=$(git rev-parse --short HEAD)
H="# round ${R#r} (commit $C): tools/profile_round.sh"
for f in vqvae_train_b8_kernel_stats.txt vqvae_train_b8_kernel_stats_by_grid.txt vqvae_pmc_hbm.txt vqvae_pmc_sq.txt vqvae_pmc_lds.txt; do
  [ -f $P/$f ] && (echo "$H vqvae -- rocprofv3 passes of: SA_NO_SIDE_WGRAD_VQVAE=1 python bench.py --no-performer --no-cpu-baseline --no-extras --no-kernel-timer --steps 2 --warmup 1 (MI355X, batch 8, weight gradients on the launch stream; 3 training steps + the eval passes of the inference leg)"; cat $P/$f) > profiles/${R}_$f
done
for f in performer_train_kernel_stats.txt performer_train_one_stream_kernel_stats.txt performer_pmc_lds.txt performer_pmc_hbm.txt dense_layers_by_grid.txt; do
  [ -f $P/$f ] && (echo "$H performer -- (MI355X; Performer N=1400, batch 6, 4 steps; *_one_stream / pmc: SA_NO_SIDE_WGRAD=1, weight gradients not overlapped; dense_layers: tools/bench_dense_tiles.py under rocprofv3)"; cat $P/$f) > profiles/${R}_$f
done
[ -f $P/performer_sampling_kernel_stats.txt ] && (echo "$H sampling -- python bench.py --only-performer --steps 1 --warmup 1 --no-kernel-timer (training steps + one sample() of 6 x 1400 tokens)"; cat $P/performer_sampling_kernel_stats.txt) > profiles/${R}_performer_sampling_kernel_stats.txt
[ -f $P/pmc_traffic.json ] && cp $P/pmc_traffic.json profiles/${R}_pmc_traffic.json
[ -f $P/pmc_traffic_performer.json ] && cp $P/pmc_traffic_performer.json profiles/${R}_pmc_traffic_performer.json
ls profiles | grep "^${R}_"
