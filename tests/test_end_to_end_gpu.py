"""GPU: BASELINE config 5 at toy scale through the CLIs -- VQ-VAE training -> extracting (uint16 codes on disk) ->
Performer training -> inference (autoregressive sampling) -> VQ-VAE decoding."""
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_pipeline(tmp_path):
    import run_transformer
    import run_vqvae
    proj = str(tmp_path) + "/"
    vq = ["--project_directory=" + proj, "--experiment_name=e2e", "--no_levels=2", "--downsample_parameters=((4,2,1,1),(4,2,1,1))",
          "--upsample_parameters=((4,2,1,0,1),(4,2,1,0,1))", "--no_channels=32", "--num_embeddings=(64,)", "--embedding_dim=(16,)", "--decay=(0.5,)",
          "--roi=((0,16),(0,24),(0,16))", "--batch_size=2", "--eval_batch_size=2", "--loss=mse", "--learning_rate=1e-3"]
    run_vqvae.run(vq + ["--training_subjects=synthetic:4", "--validation_subjects=synthetic:2", "--mode=training", "--epochs=2"])
    ck = glob.glob(proj + "e2e/baseline_vqvae/checkpoints/checkpoint_epoch=*.pt")
    assert len(ck) == 1
    run_vqvae.run(vq + ["--training_subjects=synthetic:4", "--validation_subjects=synthetic:3", "--mode=extracting"])
    codes = sorted(glob.glob(proj + "e2e/baseline_vqvae/outputs/*/*_quantization_0.npy"))
    assert len(codes) == 3
    c0 = np.load(codes[0])
    assert c0.dtype == np.uint16 and c0.shape == (4, 6, 4) and c0.max() < 64
    rec = np.load(codes[0].replace("quantization_0", "reconstruction"))
    assert rec.shape == (16, 24, 16) and np.isfinite(rec).all()
    tr = ["--project_directory=" + proj, "--experiment_name=e2e", "--vocab_size=64", "--n_embd=64", "--n_layers=2", "--n_head=2", "--local_attn_heads=1",
          "--local_window_size=24", "--use_rezero=True", "--spatial_position_emb=absolute", "--feature_redraw_interval=1", "--batch_size=3",
          "--eval_batch_size=2", "--log_every=1", "--learning_rate=1e-3", "--ordering_type=hilbert_curve"]
    code_dir = proj + "e2e/baseline_vqvae/outputs/*/*_quantization_0.npy"
    run_transformer.run(tr + ["--training_subjects=" + code_dir, "--validation_subjects=" + code_dir, "--mode=training", "--epochs=3", "--checkpoint_every=1"])
    assert glob.glob(proj + "e2e/performer/checkpoints/checkpoint_epoch=3.pt")
    run_transformer.run(tr + ["--training_subjects=" + code_dir, "--validation_subjects=synthetic:2", "--mode=inference", "--spatial_shape=(4,6,4)", "--top_k=8"])
    samples = sorted(glob.glob(proj + "e2e/performer/outputs/*/*_sample.npy"))
    assert len(samples) == 2
    s0 = np.load(samples[0])
    assert s0.dtype == np.uint16 and s0.shape == (4, 6, 4) and s0.max() <= 64
    np.save(samples[0], np.minimum(s0, 63).astype(np.uint16))  # a sampled BOS id (64) is not a codebook entry
    np.save(samples[1], np.minimum(np.load(samples[1]), 63).astype(np.uint16))
    run_vqvae.run(vq + ["--training_subjects=" + proj + "e2e/performer/outputs/*/*_sample.npy", "--validation_subjects=synthetic:1", "--mode=decoding"])
    dec = glob.glob(proj + "e2e/baseline_vqvae/outputs/*/*_sample_sample.npy")
    assert len(dec) == 2 and np.load(dec[0]).shape == (16, 24, 16)


def test_adversarial_training_step(tmp_path):
    """generator step with the least-square GAN term + discriminator step (run_vqvae.py --adversarial_component)."""
    import torch

    import run_vqvae
    proj = str(tmp_path) + "/"
    vq = ["--project_directory=" + proj, "--experiment_name=adv", "--no_levels=2", "--downsample_parameters=((4,2,1,1),(4,2,1,1))",
          "--upsample_parameters=((4,2,1,0,1),(4,2,1,0,1))", "--no_channels=32", "--num_embeddings=(64,)", "--embedding_dim=(16,)", "--decay=(0.5,)",
          "--roi=((0,32),(0,32),(0,32))", "--batch_size=2", "--eval_batch_size=2", "--loss=mse", "--learning_rate=1e-3", "--adversarial_component=True"]
    run_vqvae.run(vq + ["--training_subjects=synthetic:4", "--validation_subjects=synthetic:2", "--mode=training", "--epochs=1"])
    ck = glob.glob(proj + "adv/baseline_vqvae/checkpoints/checkpoint_epoch=*.pt")
    assert len(ck) == 1
    sd = torch.load(ck[0], map_location="cpu", weights_only=False)
    assert "network" in sd and all(torch.isfinite(v).all() for v in sd["network"].values() if v.is_floating_point())
