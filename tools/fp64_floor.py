"""Dev: how far the CPU oracle in fp32 is from its own fp64 evaluation on the config-2 encoder gradients (the fp32 error floor quoted in DESIGN.md section 3)."""
import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import vqvae_ref
NET = dict(n_levels=4, downsample_parameters=((4, 2, 1, 1),) * 4, upsample_parameters=((4, 2, 1, 0, 1),) * 4, n_embed=2048, embed_dim=32, n_channels=256,
           n_res_channels=256, n_res_layers=3, p_dropout=0.0, commitment_cost=0.25, vq_decay=0.5)
cfg = vqvae_ref.VQVAEConfig(**NET)
st = vqvae_ref.init_state(cfg, seed=4)
torch.manual_seed(21)
x = torch.rand(1, 1, 32, 48, 32)
with torch.no_grad():
    ev = vqvae_ref.forward({k: v.clone() for k, v in st.items()}, cfg, x, training=False)
    zq = vqvae_ref.embed(st, ev["indices"])
res = {}
for dt in (torch.float32, torch.float64):
    t0 = time.time()
    leaf = {k: v.clone().to(dt).requires_grad_(True) for k, v in st.items() if "quantizer" not in k}
    z = vqvae_ref.encode(leaf, cfg, x.to(dt))
    torch.nn.functional.mse_loss(z, zq.to(dt)).backward()
    res[dt] = {k: v.grad.double() for k, v in leaf.items() if v.grad is not None}
    print(dt, time.time() - t0, flush=True)
fro = lambda a, b: float((a - b).norm() / b.norm())
mx = lambda a, b: float((a - b).abs().max() / b.abs().max())
tab = sorted(((fro(res[torch.float32][k], res[torch.float64][k]), mx(res[torch.float32][k], res[torch.float64][k]), k) for k in res[torch.float64]), reverse=True)
print(tab[:6])
