"""CPU: the C-ABI library builds, loads, and exports every symbol include/synthanatomy_hip.h declares.
No compute call is made here (there is no GPU in this tier, and no CPU fallback exists by design)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "synthanatomy_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(sa_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from synthanatomy_amd import _ffi
    from synthanatomy_amd.build import build
    lib = ctypes.CDLL(build(verbose=False))
    hdr = _header_symbols()
    assert len(hdr) >= 14
    for s in hdr:
        assert hasattr(lib, s), f"{s} declared in the header but not exported"
    assert sorted(_ffi.declared_symbols()) == hdr, "ctypes signatures out of sync with include/synthanatomy_hip.h"
    assert lib.sa_abi_version() == 4 == _ffi.ABI_VERSION


def test_ctypes_struct_layout_matches_header():
    from synthanatomy_amd import _ffi
    # 14 scalars + 6 int[3] + 2 scalars
    assert ctypes.sizeof(_ffi.ConvGeom) == 4 * (14 + 18 + 2)
    assert ctypes.sizeof(_ffi.Epilogue) == 4 * 8 + 6 * 4 + 4 + 4 + 2 * 8  # 4 pointers, 6 ints, slope, padding, 2 extra-output pointers
    assert _ffi.Epilogue.out_pre.offset == 64 and _ffi.Epilogue.out_lp.offset == 72
    assert ctypes.sizeof(_ffi.LocalAttnArgs) == 3 * 8 + 8 * 4 + 11 * 8 + 2 * 4 and _ffi.LocalAttnArgs.o.offset == 56 and _ffi.LocalAttnArgs.L.offset == 144   # sa_local_attn_args
    assert ctypes.sizeof(_ffi.PackDesc) == 2 * 8 + 64 * 4 + 2 * 8 + 8 * 4  # 2 pointers, tap table, 2 strides, 8 ints


def test_product_path_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from synthanatomy_amd import _ffi
    from synthanatomy_amd.networks.vqvae.baseline import BaselineVQVAE
    net = BaselineVQVAE(n_levels=2, downsample_parameters=((4, 2, 1, 1),) * 2, upsample_parameters=((4, 2, 1, 0, 1),) * 2, n_embed=16, embed_dim=8,
                        n_channels=16, n_res_channels=16, n_res_layers=1)
    with pytest.raises(_ffi.HipLibraryError):
        net(torch.rand(1, 1, 8, 8, 8))
    with pytest.raises(_ffi.HipLibraryError):
        net.decode_samples([torch.zeros(1, 2, 2, 2, dtype=torch.long)])


def test_plugin_surface_and_state_dict_keys():
    import torch
    from synthanatomy_amd.networks.vqvae.configure import get_vqvae_network
    cfg = dict(network="baseline_vqvae", no_levels=4, downsample_parameters=((4, 2, 1, 1),) * 4, upsample_parameters=((4, 2, 1, 0, 1),) * 4,
               num_embeddings=(2048,), embedding_dim=(32,), commitment_cost=(0.25,), no_channels=16, no_res_layers=3, dropout=0.0, decay=(0.5,),
               use_subpixel_conv=False)
    net = get_vqvae_network(cfg)
    keys = list(net.state_dict().keys())
    assert len(keys) == 120  # 120 state-dict entries for no_levels=4
    for k in ("encoder.0.0.weight", "encoder.0.2.0.0.weight", "encoder.0.2.2.3.bias", "encoder.0.12.weight", "quantizer.0.impl.weight",
              "quantizer.0.impl.N", "quantizer.0.impl.embed_avg", "quantizer.0.impl.embedding.weight", "decoder.0.0.weight", "decoder.0.1.0.0.weight",
              "decoder.0.2.weight", "decoder.0.11.weight"):
        assert k in keys, k
    assert net.get_last_layer().shape == (8, 1, 4, 4, 4)
    assert net.get_ema_decay() == [0.5] and net.set_ema_decay(0.7) == [0.7] and net.set_ema_decay([0.6]) == [0.6]
    assert net.get_commitment_cost() == [0.25] and net.set_commitment_cost(0.3) == [0.3]
    assert len(net.get_perplexity()) == 1
    with pytest.raises(ValueError):
        get_vqvae_network(dict(cfg, network="nope"))
    with pytest.raises(AssertionError):
        get_vqvae_network(dict(cfg, no_levels=3))
