"""GPU: BASELINE.json configs[4] at PRODUCTION size through the CLIs -- config-2 VQ-VAE on 160x224x160 volumes (train -> extract uint16 [10,14,10]
codes) -> Performer at the README widths on the 1 400-token raster sequences (train -> stateful sample()) -> VQ-VAE decoding back to 160x224x160
(reference run_vqvae.py:395-535, run_transformer.py:296-402, src/handlers/general.py:491-590; chain in tools/end_to_end.py).  Once in this process
on one GPU, once with TWO ranks per stage under torch.distributed.run (both on cuda:0 over gloo, SA_SHARE_DEVICE): DistributedSampler-style file
sharding, bucketed gradient reduction and the summed EMA statistics at full size."""
import glob
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _check_outputs(res, extract, samples):
    assert len(res["codes"]) == extract and len(res["samples"]) == samples and len(res["decoded"]) == samples, {k: len(res[k]) for k in ("codes", "samples", "decoded")}
    for f in res["codes"]:
        c = np.load(f)
        assert c.dtype == np.uint16 and c.shape == (10, 14, 10) and c.max() < 2048
        r = np.load(f.replace("quantization_0", "reconstruction"))
        assert r.dtype == np.float32 and r.shape == (160, 224, 160) and np.isfinite(r).all()
    for f in res["samples"]:
        s = np.load(f)
        assert s.dtype == np.uint16 and s.shape == (10, 14, 10) and s.max() < 2048
    for f in res["decoded"]:
        d = np.load(f)
        assert d.dtype == np.float32 and d.shape == (160, 224, 160) and np.isfinite(d).all() and d.std() > 0
    assert all(v > 0 for v in res["seconds"].values()) and set(res["seconds"]) == {"vqvae_training", "vqvae_extracting", "performer_training",
                                                                                   "performer_inference", "vqvae_decoding"}


def test_production_size_chain_on_one_gpu(tmp_path):
    import run_vqvae
    from synthanatomy_amd.utils.general import check_for_checkpoints, create_folder_structure, load_network_state, parse_flags
    from tools import end_to_end
    res = end_to_end.run_chain(str(tmp_path), volumes=4, extract=3, samples=2)
    print("[end_to_end world 1]", res["seconds"], "total", res["total_s"], "BOS clamped", res["bos_tokens_clamped"])
    _check_outputs(res, 3, 2)
    ck = glob.glob(res["project"] + res["experiment"] + "/baseline_vqvae/checkpoints/checkpoint_epoch=1.pt")
    assert len(ck) == 1
    assert glob.glob(res["project"] + res["experiment"] + "/performer/checkpoints/checkpoint_epoch=2.pt")
    # the extraction files against the network API on the same checkpoint: index_quantize -> codes on disk, decode_samples(index_quantize(x)) == eval forward
    cfg = parse_flags(end_to_end.vqvae_flags(res["project"], res["experiment"]) + ["--training_subjects=synthetic:4", "--validation_subjects=synthetic:3",
                                                                                 "--mode=extracting"], run_vqvae.DEFAULTS)
    create_folder_structure(cfg)
    dev = torch.device("cuda", 0)
    net = run_vqvae.build_network(cfg, dev).eval()
    load_network_state(net, check_for_checkpoints(cfg))
    x = torch.stack([run_vqvae._load_volume(f"synthetic_{i:04d}", cfg, None, dev) for i in range(2)])
    with torch.no_grad():
        idx = net.index_quantize(x)[0]
        rec = net.decode_samples([idx])
        fwd = net(x)["reconstruction"][0]
    assert idx.dtype == torch.int64 and tuple(idx.shape) == (2, 10, 14, 10)
    # (eval forward decodes the straight-through value (zq - z) + z, decode_samples the table row itself: equal up to that fp32 rounding ahead of the bf16 decoder)
    dev_max = float((rec.float() - fwd.float()).abs().max() / fwd.float().abs().max())
    print(f"[end_to_end] decode_samples(index_quantize(x)) vs eval forward: max-rel {dev_max:.2e}")
    assert dev_max <= 2e-2, dev_max
    for i in range(2):
        on_disk = np.load(res["codes"][i])
        assert np.array_equal(on_disk.astype(np.int64), idx[i].cpu().numpy())
        r = np.load(res["codes"][i].replace("quantization_0", "reconstruction"))
        assert np.array_equal(r, rec[i, 0].float().cpu().numpy())
    # decoding stage == decode_samples on the sampled code grid
    s0 = torch.from_numpy(np.load(res["samples"][0]).astype(np.int64))[None].to(dev)
    with torch.no_grad():
        d0 = net.decode_samples([s0])[0, 0].float().cpu().numpy()
    name = os.path.basename(res["samples"][0])[:-len(".npy")]
    assert np.array_equal(np.load([f for f in res["decoded"] if name in f][0]), d0)


def test_production_size_chain_with_two_ranks_sharing_the_device(tmp_path):
    from tools import end_to_end
    res = end_to_end.run_chain(str(tmp_path), volumes=4, extract=3, samples=2, vq_batch=1, tr_batch=1, world=2, share_device=True)
    print("[end_to_end world 2, shared device]", res["seconds"], "total", res["total_s"])
    _check_outputs(res, 3, 2)     # 3 inputs over 2 ranks: no duplicates, nothing dropped (even_divisible=False sharding of the inference modes)
    sd = torch.load(glob.glob(res["project"] + res["experiment"] + "/baseline_vqvae/checkpoints/checkpoint_epoch=1.pt")[0], map_location="cpu", weights_only=False)
    assert all(torch.isfinite(v).all() for v in sd["network"].values() if v.is_floating_point())
    assert int(sd["trainer"]["iteration"]) == 2      # 4 volumes / (2 ranks x batch 1)
