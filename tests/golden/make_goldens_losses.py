#!/usr/bin/env python3
"""Generate ``losses.npz`` by importing the REFERENCE's loss classes (build container only; ``/root/reference`` is absent on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_goldens_losses.py

* ``src.losses.adversarial.adversarial.AdversarialLoss`` (adversarial.py:11-105) for the three criteria, generator and discriminator form,
  weight 0.005 as the factories set it (configure.py:19-38): loss values and d loss / d logits;
* ``src.losses.vqvae.vqvae.JukeboxLoss(dimensions=3)`` (vqvae.py:522-638): loss value, its spectral part, and d loss / d reconstruction.

Import recipe: both modules import ``src.handlers.general`` (needs ignite / MONAI, absent here) only for the ``TBSummaryTypes`` enum whose
members are used as dictionary keys, and ``vqvae.py`` imports ``lpips.LPIPS`` which ``JukeboxLoss`` never constructs -> ``sys.modules`` is
pre-seeded with placeholders for exactly those two names.  Nothing from the reference is copied: the fixture holds tensors only.
"""
import enum
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def _placeholders():
    handlers = types.ModuleType("src.handlers.general")

    class TBSummaryTypes(enum.Enum):   # only ever used as a dict key
        SCALAR = "scalar"

    handlers.TBSummaryTypes = TBSummaryTypes
    sys.modules["src.handlers.general"] = handlers
    lp = types.ModuleType("lpips")

    class LPIPS:  # placeholder, never constructed by JukeboxLoss
        pass

    lp.LPIPS = LPIPS
    sys.modules["lpips"] = lp


def main():
    assert os.path.isdir(REF), "reference tree not present: goldens can only be regenerated in the build container"
    sys.path.insert(0, REF)
    _placeholders()
    from src.losses.adversarial.adversarial import AdversarialLoss
    from src.losses.vqvae.vqvae import JukeboxLoss

    out = {}
    g = torch.Generator().manual_seed(31)
    fake = torch.randn(2, 1, 3, 4, 3, generator=g)
    real = torch.randn(2, 1, 3, 4, 3, generator=g) + 0.5
    out["adv/logits_fake"], out["adv/logits_real"] = fake.numpy(), real.numpy()
    for crit in ("vanilla", "hinge", "least_square"):
        f = fake.clone().requires_grad_(True)
        lg = AdversarialLoss(criterion=crit, is_discriminator=False, weight=0.005)(f)
        lg.backward()
        out[f"adv/{crit}/generator"], out[f"adv/{crit}/generator_dfake"] = lg.detach().numpy(), f.grad.numpy()
        f = fake.clone().requires_grad_(True)
        r = real.clone().requires_grad_(True)
        ld = AdversarialLoss(criterion=crit, is_discriminator=True, weight=0.005)(f, r)
        ld.backward()
        out[f"adv/{crit}/discriminator"] = ld.detach().numpy()
        out[f"adv/{crit}/discriminator_dfake"], out[f"adv/{crit}/discriminator_dreal"] = f.grad.numpy(), r.grad.numpy()

    y = torch.rand(2, 1, 8, 12, 10, generator=g)
    pred = (y + 0.1 * torch.randn(2, 1, 8, 12, 10, generator=g)).requires_grad_(True)
    ql = torch.tensor(0.0123)
    loss_fn = JukeboxLoss(dimensions=3)
    loss = loss_fn({"reconstruction": [pred], "quantization_losses": [ql]}, y)
    loss.backward()
    out["jukebox/y"], out["jukebox/pred"], out["jukebox/qloss"] = y.numpy(), pred.detach().numpy(), ql.numpy()
    out["jukebox/loss"], out["jukebox/dpred"] = loss.detach().numpy(), pred.grad.numpy()
    spec = [v for k, v in loss_fn.get_summaries()[list(loss_fn.get_summaries())[0]].items() if "Spectral" in k][0]
    out["jukebox/spectral"] = spec.detach().numpy()
    np.savez_compressed(os.path.join(HERE, "losses.npz"), **out)
    print("losses ok:", {k: float(v) for k, v in out.items() if v.ndim == 0})


if __name__ == "__main__":
    main()
