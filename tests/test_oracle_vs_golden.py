"""CPU: the oracle (our restatement) against fixtures computed by the reference itself (tests/golden)."""
import ctypes
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden, meta_of, state_from_golden
from oracle import ordering_ref, vqvae_ref


def _cfg(meta):
    kw = meta["net_kwargs"]
    return vqvae_ref.VQVAEConfig(
        n_levels=kw["n_levels"], downsample_parameters=tuple(map(tuple, kw["downsample_parameters"])),
        upsample_parameters=tuple(map(tuple, kw["upsample_parameters"])), n_embed=kw["n_embed"], embed_dim=kw["embed_dim"],
        n_channels=kw["n_channels"], n_res_channels=kw["n_res_channels"], n_res_layers=kw["n_res_layers"],
        commitment_cost=kw["commitment_cost"], vq_decay=kw["vq_decay"])


@pytest.mark.parametrize("name", ["vqvae_cfg1", "vqvae_tiny4"])
def test_vqvae_oracle_eval(name):
    g = load_golden(name)
    cfg = _cfg(meta_of(g))
    st = state_from_golden(g)
    x = torch.from_numpy(g["x"])
    with torch.no_grad():
        out = vqvae_ref.forward(st, cfg, x, training=False)
    assert np.array_equal(out["indices"].numpy(), g["eval/idx"])
    np.testing.assert_allclose(out["z"].numpy(), g["eval/z"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(out["reconstruction"][0].numpy(), g["eval/recon"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(out["quantization_losses"][0].numpy(), g["eval/qloss"], rtol=1e-6)
    with torch.no_grad():
        rec2 = vqvae_ref.decode(st, cfg, vqvae_ref.embed(st, torch.from_numpy(g["eval/idx"])))
    np.testing.assert_allclose(rec2.numpy(), g["eval/decode_samples"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("name", ["vqvae_cfg1", "vqvae_tiny4"])
def test_vqvae_oracle_train(name):
    g = load_golden(name)
    cfg = _cfg(meta_of(g))
    st = state_from_golden(g)
    leaf = {k: v.clone().requires_grad_(True) for k, v in st.items() if "quantizer" not in k}
    stt = dict(st)
    stt.update(leaf)
    x = torch.from_numpy(g["x"])
    for step in (1, 2):
        for p in leaf.values():
            p.grad = None
        out = vqvae_ref.forward(stt, cfg, x, training=True)
        loss = vqvae_ref.mse_loss(out, x)
        loss.backward()
        np.testing.assert_allclose(out["reconstruction"][0].detach().numpy(), g[f"train{step}/recon"], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(loss.item(), g[f"train{step}/loss"], rtol=1e-5)
        np.testing.assert_allclose(vqvae_ref.perplexity(out["indices"], cfg.n_embed).numpy(), g[f"train{step}/perplexity"], rtol=1e-5)
        for nm in ("N", "embed_avg", "weight"):
            np.testing.assert_allclose(stt["quantizer.0.impl." + nm].numpy(), g[f"train{step}/{nm}"], rtol=1e-5, atol=1e-6)
        if step == 1:
            n = 0
            for k in g.files:
                if k.startswith("train1/grad/"):
                    pk = k[len("train1/grad/"):]
                    ref = g[k]
                    got = leaf[pk].grad.numpy()
                    assert np.abs(got - ref).max() <= 1e-4 * np.abs(ref).max() + 1e-8, pk
                    n += 1
            assert n > 5


def _lib():
    path = os.path.join(ROOT, "oracle", "libsa_oracle.so")
    if not os.path.exists(path):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    return ctypes.CDLL(path)


def c_vq_assign(rows, cb):
    lib = _lib()
    M, D = rows.shape
    K = cb.shape[0]
    idx = np.zeros(M, np.int64)
    counts = np.zeros(K, np.float32)
    dw = np.zeros((K, D), np.float32)
    gap = np.zeros(M, np.float32)
    se = ctypes.c_double(0)
    P = ctypes.c_void_p
    lib.sa_oracle_vq_assign(P(rows.ctypes.data), P(cb.ctypes.data), ctypes.c_int64(M), K, D, P(idx.ctypes.data),
                            P(counts.ctypes.data), P(dw.ctypes.data), ctypes.byref(se), P(gap.ctypes.data))
    return idx, counts, dw, se.value, gap


def test_c_quantizer_vs_golden():
    g = load_golden("quantizer")
    lib = _lib()
    cb = g["W0"].copy()
    K, D = cb.shape
    N = np.zeros(K, np.float32)
    avg = cb.copy()
    # eval case
    x = np.ascontiguousarray(np.transpose(g["eval/x"], (0, 2, 3, 4, 1)).reshape(-1, D))
    idx, counts, dw, se, gap = c_vq_assign(x, cb)
    assert np.array_equal(idx.reshape(g["eval/idx"].shape), g["eval/idx"])
    np.testing.assert_allclose(gap, g["eval/top2gap"], rtol=0, atol=2e-4)
    np.testing.assert_allclose(0.25 * se / x.size, g["eval/loss"], rtol=1e-5)
    for s in range(3):
        x = np.ascontiguousarray(np.transpose(g[f"train{s}/x"], (0, 2, 3, 4, 1)).reshape(-1, D))
        idx, counts, dw, se, gap = c_vq_assign(x, cb)
        assert np.array_equal(idx.reshape(g[f"train{s}/idx"].shape), g[f"train{s}/idx"]), s
        zq = cb[idx]
        np.testing.assert_allclose(np.transpose(g[f"train{s}/zq"], (0, 2, 3, 4, 1)).reshape(-1, D), (zq - x) + x, rtol=1e-6, atol=1e-6)
        P = ctypes.c_void_p
        lib.sa_oracle_vq_ema_update(P(N.ctypes.data), P(avg.ctypes.data), P(cb.ctypes.data), P(counts.ctypes.data), P(dw.ctypes.data),
                                    K, D, ctypes.c_float(0.5), ctypes.c_float(1e-5))
        np.testing.assert_allclose(N, g[f"train{s}/N"], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(avg, g[f"train{s}/embed_avg"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(cb, g[f"train{s}/weight"], rtol=2e-5, atol=1e-6)


def test_torch_quantizer_oracle_vs_golden():
    g = load_golden("quantizer")
    cfg = vqvae_ref.VQVAEConfig(n_embed=2048, embed_dim=32, vq_decay=0.5)
    W = torch.from_numpy(g["W0"].copy())
    st = {"quantizer.0.impl.weight": W, "quantizer.0.impl.N": torch.zeros(2048), "quantizer.0.impl.embed_avg": W.clone()}
    for s in range(3):
        zq, loss, idx, aux = vqvae_ref.quantize(st, cfg, torch.from_numpy(g[f"train{s}/x"]), training=True)
        assert np.array_equal(idx.numpy(), g[f"train{s}/idx"])
        np.testing.assert_allclose(loss.numpy(), g[f"train{s}/loss"], rtol=1e-6)
        np.testing.assert_allclose(st["quantizer.0.impl.weight"].numpy(), g[f"train{s}/weight"], rtol=1e-6, atol=1e-7)
    # perplexity fixture
    cfg2 = vqvae_ref.VQVAEConfig(n_embed=64, embed_dim=8, vq_decay=0.9)
    W = torch.from_numpy(g["ppl/W0"].copy())
    st = {"quantizer.0.impl.weight": W, "quantizer.0.impl.N": torch.zeros(64), "quantizer.0.impl.embed_avg": W.clone()}
    zq, loss, idx, aux = vqvae_ref.quantize(st, cfg2, torch.from_numpy(g["ppl/x"]), training=True)
    np.testing.assert_allclose(vqvae_ref.perplexity(idx, 64).numpy(), g["ppl/perplexity"], rtol=1e-6)


def _sha(a):
    return hashlib.sha1(np.asarray(a).astype(np.int64).tobytes()).hexdigest()


def _ordering_cases():
    g = load_golden("ordering")
    shas = json.loads(bytes(g["sha_json"]).decode())
    return g, shas


@pytest.mark.parametrize("impl", ["oracle", "product"])
def test_ordering_vs_golden(impl):
    g, shas = _ordering_cases()
    if impl == "oracle":
        def make(typ, nd, dims, refl=None, tr=(), rot=(), order=("transpose", "rotate_90", "reflect")):
            refl = refl if refl is not None else (False,) * nd
            o, r = ordering_ref.ordering(typ, nd, (1,) + tuple(dims), refl, tr, rot, order)
            return o, r
    else:
        from synthanatomy_amd.networks.transformers.img2seq_ordering import Ordering

        def make(typ, nd, dims, refl=None, tr=(), rot=(), order=("transpose", "rotate_90", "reflect")):
            refl = refl if refl is not None else (False,) * nd
            o = Ordering(typ, nd, (1,) + tuple(dims), refl, tr, rot, order)
            return o.get_sequence_ordering(), o.get_revert_sequence_ordering()
    n = 0
    for key, sha in shas.items():
        typ, dims = key.split("/")
        if typ in ("readme", "mix", "mix2"):
            continue
        dims = tuple(int(v) for v in dims.split("x"))
        o, r = make(typ, len(dims), dims)
        assert _sha(o) == sha, key
        if key in g.files:
            assert np.array_equal(o, g[key])
        if key + "/revert" in g.files:
            assert np.array_equal(r, g[key + "/revert"])
        n += 1
    assert n >= 27
    # the SHA-1 prefixes recorded in SURVEY.md section 8 B1
    assert shas["raster_scan/10x14x10"].startswith("bc3230f5fc48")
    assert shas["hilbert_curve/10x14x10"].startswith("c9b7c71f1d7b")
    assert shas["hilbert_curve/20x28x25"].startswith("68f59ebc88de")
    o, _ = make("raster_scan", 3, (10, 14, 10), None, ((2, 0, 1),), ((0, 1),), ("rotate_90", "transpose"))
    assert _sha(o) == shas["readme/10x14x10"] and shas["readme/10x14x10"].startswith("077ffd6b8ef6")
    assert list(o[:12]) == [130, 270, 410, 550, 690, 830, 970, 1110, 1250, 1390, 120, 260]
    o, _ = make("s_curve", 3, (4, 6, 5), (True, False, True), ((1, 0, 2),), ((1, 2),))
    assert np.array_equal(o, g["mix/4x6x5"])
    o, _ = make("hilbert_curve", 3, (4, 6, 5), (False, True, False), ((0, 2, 1), (1, 0, 2)), ((0, 2), (0, 1)), ("reflect", "transpose", "rotate_90"))
    assert np.array_equal(o, g["mix2/4x6x5"])
    np.random.seed(7)
    o, _ = make("random", 3, (3, 4, 5))
    assert np.array_equal(o, g["random_seed7/3x4x5"])


def test_ordering_errors_product():
    from synthanatomy_amd.networks.transformers.img2seq_ordering import Ordering

    with pytest.raises(AssertionError):
        Ordering("zigzag", 3, (1, 2, 2, 2), (False,) * 3, (), ())
    with pytest.raises(AssertionError):
        Ordering("raster_scan", 3, (2, 2, 2), (False,) * 3, (), ())
    with pytest.raises(ValueError):
        Ordering("raster_scan", 3, (1, 2, 2, 2), (False,) * 3, (), (), ("transpose", "transpose"))
    with pytest.raises(ValueError):
        Ordering("raster_scan", 3, (1, 2, 2, 2), (False,) * 3, (), (), ("shear",))


def test_discriminator_oracle_vs_golden():
    g = load_golden("discriminator")
    st = state_from_golden(g)
    x = torch.from_numpy(g["x"])
    st_train = {k: v.clone() for k, v in st.items()}
    y = vqvae_ref.discriminator_forward(st_train, x, training=True)
    np.testing.assert_allclose(y.numpy(), g["train/logits"], rtol=1e-4, atol=1e-5)
    for k in g.files:
        if k.startswith("train/sd/"):
            np.testing.assert_allclose(st_train[k[len("train/sd/"):]].numpy(), g[k], rtol=1e-5, atol=1e-6)
    y = vqvae_ref.discriminator_forward(st_train, x, training=False)
    np.testing.assert_allclose(y.numpy(), g["eval/logits"], rtol=1e-4, atol=1e-5)
