#!/usr/bin/env python3
"""BASELINE.json configs[4] at PRODUCTION size through the CLIs (reference README.md:100-141, run_vqvae.py:395-535, run_transformer.py:296-402):

    run_vqvae.py --mode=training     config-2 network (no_levels=4, no_channels=256, D=32, K=2048) on synthetic 160x224x160 volumes, 1 epoch
    run_vqvae.py --mode=extracting   -> <name>_quantization_0.npy  uint16 [10,14,10]   (+ fp32 reconstructions)
    run_transformer.py --mode=training    README widths (n_embd 512, 24 layers, 16 heads, 8 local heads, window 420, ReZero, absolute spatial
                                          embeddings, raster scan with the README transforms) on the extracted codes
    run_transformer.py --mode=inference   stateful sample(): 1 400 tokens per sample -> <name>_sample.npy  uint16 [10,14,10]
    run_vqvae.py --mode=decoding     -> <name>_sample_sample.npy  fp32 [160,224,160]

``world == 1`` runs the stages in this process; ``world > 1`` starts every stage under ``torch.distributed.run`` (one process per rank;
``share_device=True`` puts all ranks on cuda:0 over gloo -- SA_SHARE_DEVICE, the N > 1 code path on a one-GPU box).  Returns the per-stage
wall seconds and the output listings; ``tests/test_end_to_end_fullsize_gpu.py`` asserts on them, ``bench.py`` reports them as ``end_to_end``.

    python tools/end_to_end.py /tmp/proj/ [--world 2 --share-device]
"""
import glob
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ROI = ((0, 160), (0, 224), (0, 160))
LATENT = (10, 14, 10)


def vqvae_flags(proj, exp, batch=2):
    return ["--project_directory=" + proj, "--experiment_name=" + exp, "--no_levels=4", "--downsample_parameters=((4,2,1,1),(4,2,1,1),(4,2,1,1),(4,2,1,1))",
            "--upsample_parameters=((4,2,1,0,1),(4,2,1,0,1),(4,2,1,0,1),(4,2,1,0,1))", "--no_channels=256", "--no_res_layers=3", "--num_embeddings=(2048,)",
            "--embedding_dim=(32,)", "--decay=(0.5,)", "--commitment_cost=(0.25,)", f"--roi={ROI}".replace(" ", ""), f"--batch_size={batch}",
            f"--eval_batch_size={batch}", "--loss=mse", "--learning_rate=1.65e-4", "--amp=True", "--checkpoint_every=1", "--eval_every=1000"]


def transformer_flags(proj, exp, batch=2):
    return ["--project_directory=" + proj, "--experiment_name=" + exp, "--vocab_size=2048", "--n_embd=512", "--n_layers=24", "--n_head=16",
            "--local_attn_heads=8", "--local_window_size=420", "--feature_redraw_interval=1", "--use_rezero=True", "--spatial_position_emb=absolute",
            "--ordering_type=raster_scan", "--transpositions_axes=((2,0,1),)", "--rot90_axes=((0,1),)", "--transformation_order=('rotate_90','transpose')",
            f"--batch_size={batch}", f"--eval_batch_size={batch}", "--learning_rate=1e-3", "--log_every=1", "--compute_dtype=bf16", "--eval_every=1000"]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _stage(script, argv, world, share_device):
    """One CLI invocation; returns wall seconds."""
    t0 = time.perf_counter()
    if world == 1:
        mod = __import__(script)
        mod.run(list(argv))
        import torch
        torch.cuda.synchronize()
    else:
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", "8")
        if share_device:
            env["SA_SHARE_DEVICE"] = "1"
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1", "--master-port",
               str(_free_port()), os.path.join(ROOT, script + ".py"), "run", *argv]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1800)
        if r.returncode != 0:
            raise RuntimeError(f"{script} {argv[-3:]} failed under torchrun (world {world}):\n{r.stdout[-3000:]}\n{r.stderr[-3000:]}")
    return round(time.perf_counter() - t0, 3)


def run_chain(proj, exp="e2e_full", volumes=4, extract=4, samples=2, vq_batch=2, tr_batch=2, tr_epochs=2, world=1, share_device=False):
    """Runs the five stages; returns {"seconds": {stage: s}, "codes": [...], "samples": [...], "decoded": [...], ...}."""
    proj = proj if proj.endswith("/") else proj + "/"
    vq = vqvae_flags(proj, exp, vq_batch)
    tr = transformer_flags(proj, exp, tr_batch)
    sec = {}
    sec["vqvae_training"] = _stage("run_vqvae", vq + [f"--training_subjects=synthetic:{volumes}", "--validation_subjects=synthetic:1", "--mode=training",
                                                      "--epochs=1"], world, share_device)
    sec["vqvae_extracting"] = _stage("run_vqvae", vq + [f"--training_subjects=synthetic:{volumes}", f"--validation_subjects=synthetic:{extract}",
                                                        "--mode=extracting"], world, share_device)
    out_vq = proj + exp + "/baseline_vqvae/outputs/"
    codes = sorted(glob.glob(out_vq + "*/*_quantization_0.npy"))
    sec["performer_training"] = _stage("run_transformer", tr + ["--training_subjects=" + out_vq, "--validation_subjects=" + out_vq, "--mode=training",
                                                                f"--epochs={tr_epochs}", "--checkpoint_every=1"], world, share_device)
    sec["performer_inference"] = _stage("run_transformer", tr + ["--training_subjects=" + out_vq, f"--validation_subjects=synthetic:{samples}",
                                                                 "--mode=inference", f"--spatial_shape={LATENT}".replace(" ", "")], world, share_device)
    out_tr = proj + exp + "/performer/outputs/"
    sampled = sorted(glob.glob(out_tr + "*/*_sample.npy"))
    n_bos = 0
    for f in sampled:   # a sampled BOS id (== vocab_size) is not a codebook entry: torch's embedding lookup raises upstream too (a trained model never emits it)
        s = np.load(f)
        n_bos += int((s >= 2048).sum())
        if (s >= 2048).any():
            np.save(f, np.minimum(s, 2047).astype(np.uint16))
    sec["vqvae_decoding"] = _stage("run_vqvae", vq + ["--training_subjects=" + out_tr, "--validation_subjects=synthetic:1", "--mode=decoding"],
                                   world, share_device)
    decoded = sorted(glob.glob(out_vq + "*/*_sample_sample.npy"))
    return {"seconds": sec, "total_s": round(sum(sec.values()), 3), "world": world, "share_device": bool(share_device), "codes": codes,
            "samples": sampled, "decoded": decoded, "bos_tokens_clamped": n_bos, "project": proj, "experiment": exp,
            "workload": f"config-2 VQ-VAE (160x224x160, batch {vq_batch}) train 1 epoch of {volumes} -> extract {extract} -> Performer README widths "
                        f"(N=1400, batch {tr_batch}, bf16) {tr_epochs} epochs -> stateful sample() x{samples} -> decode"}


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("project")
    ap.add_argument("--world", type=int, default=1)
    ap.add_argument("--share-device", action="store_true")
    a = ap.parse_args()
    res = run_chain(a.project, world=a.world, share_device=a.share_device)
    print(json.dumps({k: v for k, v in res.items() if k not in ("codes", "samples", "decoded")} | {"n_codes": len(res["codes"]),
                     "n_samples": len(res["samples"]), "n_decoded": len(res["decoded"])}))
