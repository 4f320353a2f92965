#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by importing the REFERENCE itself.

Run only in the build container (``/root/reference`` does not exist on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_goldens.py

Nothing from the reference is copied: the fixtures hold tensors only (inputs, weights drawn by the
reference's own constructors under a fixed seed, and the outputs its code computes for them).

Import recipe: ``src/networks/vqvae/baseline.py:6`` imports one MONAI symbol that is
never instantiated when ``use_subpixel_conv=False`` -> pre-seed ``sys.modules`` with an empty placeholder.
"""
import hashlib
import json
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def _stub_monai():
    monai = types.ModuleType("monai")
    nets = types.ModuleType("monai.networks")
    blocks = types.ModuleType("monai.networks.blocks")

    class SubpixelUpsample:  # placeholder, never constructed
        pass

    blocks.SubpixelUpsample = SubpixelUpsample
    monai.networks = nets
    nets.blocks = blocks
    sys.modules.update({"monai": monai, "monai.networks": nets, "monai.networks.blocks": blocks})


def _np(sd):
    return {k: v.detach().cpu().numpy().copy() for k, v in sd.items()}


def vqvae_case(name, seed, net_kwargs, in_shape, grad_keys=None):
    from src.networks.vqvae.baseline import BaselineVQVAE

    torch.manual_seed(seed)
    net = BaselineVQVAE(**net_kwargs)
    x = torch.rand(*in_shape)
    out = {}
    sd0 = _np(net.state_dict())
    for k, v in sd0.items():
        out["sd0/" + k] = v
    out["x"] = x.numpy()

    # eval forward (no EMA side-effect)
    net.eval()
    with torch.no_grad():
        z = net.encode(x)[0]
        zq, ql, idx = net.quantizer[0].quantize(z)
        rec = net.decode([zq])
        out["eval/z"] = z.numpy()
        out["eval/idx"] = idx.numpy()
        out["eval/qloss"] = ql.numpy()
        out["eval/recon"] = rec.numpy()
        out["eval/index_quantize"] = net.index_quantize(x)[0].numpy()
        out["eval/decode_samples"] = net.decode_samples([idx]).numpy()
    assert all(np.array_equal(sd0[k], v) for k, v in _np(net.state_dict()).items())

    # two training steps (fwd + mse loss + bwd, SGD-free: we only record grads and EMA state)
    net.train()
    for step in (1, 2):
        net.zero_grad()
        o = net(x)
        loss = torch.nn.functional.mse_loss(o["reconstruction"][0].float(), x.float()) + o["quantization_losses"][0]
        loss.backward()
        out[f"train{step}/recon"] = o["reconstruction"][0].detach().numpy()
        out[f"train{step}/qloss"] = o["quantization_losses"][0].detach().numpy()
        out[f"train{step}/loss"] = loss.detach().numpy()
        out[f"train{step}/perplexity"] = net.get_perplexity()[0].numpy()
        for k, p in net.named_parameters():
            if p.grad is not None and step == 1 and (grad_keys is None or k in grad_keys):
                out[f"train{step}/grad/" + k] = p.grad.numpy().copy()
        q = net.quantizer[0].impl
        out[f"train{step}/N"] = q.N.numpy().copy()
        out[f"train{step}/embed_avg"] = q.embed_avg.numpy().copy()
        out[f"train{step}/weight"] = q.weight.detach().numpy().copy()
    out["meta"] = np.frombuffer(json.dumps({"net_kwargs": net_kwargs, "in_shape": list(in_shape), "seed": seed}).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "keys", len(out), "params", sum(p.numel() for p in net.parameters()))


def quantizer_case():
    from src.networks.vqvae.baseline import Quantizer

    torch.manual_seed(11)
    q = Quantizer(2048, 32, commitment_cost=0.25, decay=0.5)
    out = {"W0": q.impl.weight.detach().numpy().copy()}
    xs = [torch.randn(1, 32, 10, 14, 10) * 0.8 for _ in range(3)]
    q.eval()
    with torch.no_grad():
        zq, loss, idx = q.quantize(xs[0])
        flat = xs[0].permute(0, 2, 3, 4, 1).reshape(-1, 32)
        d = (flat ** 2).sum(1, keepdim=True) - 2 * flat @ q.impl.weight.t() + (q.impl.weight ** 2).sum(1, keepdim=True).t()
        top2 = torch.topk(-d, 2, dim=1)[0]
        out["eval/x"] = xs[0].numpy()
        out["eval/idx"] = idx.numpy()
        out["eval/zq"] = zq.numpy()
        out["eval/loss"] = loss.numpy()
        out["eval/top2gap"] = (top2[:, 0] - top2[:, 1]).numpy()
        assert np.array_equal(out["W0"], q.impl.weight.numpy())
    q.train()
    for s in range(3):
        with torch.no_grad():
            zq, loss, idx = q.quantize(xs[s])
            q(xs[s]) if False else None
        out[f"train{s}/x"] = xs[s].numpy()
        out[f"train{s}/idx"] = idx.numpy()
        out[f"train{s}/zq"] = zq.numpy()
        out[f"train{s}/loss"] = loss.numpy()
        out[f"train{s}/N"] = q.impl.N.numpy().copy()
        out[f"train{s}/embed_avg"] = q.impl.embed_avg.numpy().copy()
        out[f"train{s}/weight"] = q.impl.weight.detach().numpy().copy()
    # perplexity via Quantizer.forward on a fresh module (forward also runs the EMA step)
    torch.manual_seed(12)
    q2 = Quantizer(64, 8, decay=0.9)
    x = torch.randn(2, 8, 4, 4, 4)
    out["ppl/W0"] = q2.impl.weight.detach().numpy().copy()
    out["ppl/x"] = x.numpy()
    q2.train()
    zq, l = q2(x)
    out["ppl/perplexity"] = q2.get_perplexity().numpy()
    out["ppl/zq"] = zq.detach().numpy()
    np.savez_compressed(os.path.join(HERE, "quantizer.npz"), **out)
    print("quantizer ok; min top2 gap", float(out["eval/top2gap"].min()))


def ordering_case():
    from src.networks.transformers.img2seq_ordering import Ordering

    rec = {}
    arrays = {}

    def sha(a):
        return hashlib.sha1(np.asarray(a).astype(np.int64).tobytes()).hexdigest()

    for dims in [(10, 14, 10), (20, 28, 20), (20, 28, 25), (4, 6, 5), (3, 3, 3), (8, 8, 8)]:
        for typ in ["raster_scan", "s_curve", "hilbert_curve"]:
            o = Ordering(typ, 3, (1,) + dims, (False, False, False), (), ())
            key = f"{typ}/{'x'.join(map(str, dims))}"
            rec[key] = sha(o.get_sequence_ordering())
            if np.prod(dims) <= 1400:
                arrays[key] = o.get_sequence_ordering().astype(np.int64)
                arrays[key + "/revert"] = o.get_revert_sequence_ordering().astype(np.int64)
    # 2D
    for dims in [(6, 9), (16, 16), (5, 4)]:
        for typ in ["raster_scan", "s_curve", "hilbert_curve"]:
            o = Ordering(typ, 2, (1,) + dims, (False, False), (), ())
            key = f"{typ}/{'x'.join(map(str, dims))}"
            rec[key] = sha(o.get_sequence_ordering())
            arrays[key] = o.get_sequence_ordering().astype(np.int64)
    # README transform (README.md:82-84)
    o = Ordering("raster_scan", 3, (1, 10, 14, 10), (False, False, False), ((2, 0, 1),), ((0, 1),), ("rotate_90", "transpose"))
    rec["readme/10x14x10"] = sha(o.get_sequence_ordering())
    arrays["readme/10x14x10"] = o.get_sequence_ordering().astype(np.int64)
    # transforms mixture incl. reflect, default order
    o = Ordering("s_curve", 3, (1, 4, 6, 5), (True, False, True), ((1, 0, 2),), ((1, 2),))
    arrays["mix/4x6x5"] = o.get_sequence_ordering().astype(np.int64)
    rec["mix/4x6x5"] = sha(o.get_sequence_ordering())
    o = Ordering("hilbert_curve", 3, (1, 4, 6, 5), (False, True, False), ((0, 2, 1), (1, 0, 2)), ((0, 2), (0, 1)), ("reflect", "transpose", "rotate_90"))
    arrays["mix2/4x6x5"] = o.get_sequence_ordering().astype(np.int64)
    rec["mix2/4x6x5"] = sha(o.get_sequence_ordering())
    # random ordering depends on the global numpy RNG
    np.random.seed(7)
    o = Ordering("random", 3, (1, 3, 4, 5), (False, False, False), (), ())
    arrays["random_seed7/3x4x5"] = o.get_sequence_ordering().astype(np.int64)
    arrays["sha_json"] = np.frombuffer(json.dumps(rec, sort_keys=True).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "ordering.npz"), **arrays)
    print("ordering ok", rec["raster_scan/10x14x10"][:12], rec["hilbert_curve/10x14x10"][:12], rec["readme/10x14x10"][:12])


def sample_case():
    """TransformerBase.sample post-processing with a deterministic fake forward (transformer.py:58-101)."""
    from src.networks.transformers.img2seq_ordering import Ordering
    from src.networks.transformers.transformer import TransformerBase

    dims = (1, 3, 4, 2)
    o = Ordering("s_curve", 3, dims, (False, False, False), ((2, 0, 1),), ())
    V = 11

    class Fake(TransformerBase):
        def __init__(self):
            super().__init__()
            self.ordering = o
            torch.manual_seed(3)
            self.table = torch.randn(64, V + 1)

        def forward(self, x, conditioning=None):
            # logits depend on position and the previous token only -> deterministic chain
            n = x.shape[1]
            pos = torch.arange(n)
            return self.table[(x * 7 + pos[None, :] * 3) % 64]

    f = Fake()
    prefix = torch.full((2, 1), V, dtype=torch.long)
    greedy = f.sample(prefix, sample=False)
    topk = f.sample(prefix, sample=False, top_k=3, temperature=0.7)
    torch.manual_seed(5)
    stoch = f.sample(prefix, sample=True, top_k=4)
    np.savez_compressed(os.path.join(HERE, "sample.npz"), table=f.table.numpy(), greedy=greedy.numpy(), topk=topk.numpy(),
                        stoch_seed5_top4=stoch.numpy(), ordering=o.get_sequence_ordering().astype(np.int64))
    print("sample ok", tuple(greedy.shape))


def discriminator_case():
    from src.networks.discriminator.baseline import BaselineDiscriminator

    torch.manual_seed(21)
    d = BaselineDiscriminator(input_nc=1, ndf=8, n_layers=3)
    x = torch.rand(2, 1, 32, 32, 32)
    out = {"sd0/" + k: v for k, v in _np(d.state_dict()).items()}
    out["x"] = x.numpy()
    d.train()
    y = d(x)
    out["train/logits"] = y.detach().numpy()
    for k, v in _np(d.state_dict()).items():
        if "running" in k:
            out["train/sd/" + k] = v
    d.eval()
    with torch.no_grad():
        out["eval/logits"] = d(x).numpy()
    np.savez_compressed(os.path.join(HERE, "discriminator.npz"), **out)
    print("discriminator ok", tuple(y.shape))


def main():
    assert os.path.isdir(REF), "reference tree not present: goldens can only be regenerated in the build container"
    sys.path.insert(0, REF)
    _stub_monai()
    torch.set_num_threads(8)
    vqvae_case("vqvae_cfg1", 4, dict(n_levels=2, downsample_parameters=((4, 2, 1, 1),) * 2, upsample_parameters=((4, 2, 1, 0, 1),) * 2,
                                     n_embed=256, embed_dim=256, n_channels=32, n_res_channels=32, n_res_layers=3,
                                     p_dropout=0.0, commitment_cost=0.25, vq_decay=0.5), (2, 1, 32, 32, 32),
               grad_keys={"encoder.0.0.weight", "encoder.0.0.bias", "encoder.0.2.1.0.weight", "encoder.0.2.1.3.weight", "encoder.0.2.1.3.bias",
                          "encoder.0.3.weight", "encoder.0.6.bias", "decoder.0.0.bias", "decoder.0.1.0.0.weight", "decoder.0.1.2.3.weight",
                          "decoder.0.2.weight", "decoder.0.5.weight", "decoder.0.5.bias"})
    vqvae_case("vqvae_tiny4", 5, dict(n_levels=4, downsample_parameters=((4, 2, 1, 1),) * 4, upsample_parameters=((4, 2, 1, 0, 1),) * 4,
                                      n_embed=128, embed_dim=32, n_channels=16, n_res_channels=16, n_res_layers=3,
                                      p_dropout=0.0, commitment_cost=0.25, vq_decay=0.5), (1, 1, 32, 48, 32))
    quantizer_case()
    ordering_case()
    sample_case()
    discriminator_case()


if __name__ == "__main__":
    main()
