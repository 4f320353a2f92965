import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from synthanatomy_amd.networks.vqvae.baseline import BaselineVQVAE
NET = dict(n_levels=4, downsample_parameters=((4, 2, 1, 1),) * 4, upsample_parameters=((4, 2, 1, 0, 1),) * 4, n_embed=2048, embed_dim=32, n_channels=256, n_res_channels=256, n_res_layers=3)
torch.manual_seed(4)
ref = BaselineVQVAE(**NET, compute_dtype=torch.float32).cuda()
sd = ref.state_dict()
nets = {"fp32": ref, "f16": BaselineVQVAE(**NET, compute_dtype=torch.bfloat16).cuda(), "bf16": BaselineVQVAE(**NET, compute_dtype=torch.bfloat16, encoder_forward_dtype=torch.bfloat16).cuda()}
for k in ("f16", "bf16"):
    nets[k].load_state_dict(sd)
x = torch.rand(2, 1, 64, 96, 64, generator=torch.Generator().manual_seed(9)).cuda()
gz = None
G = {}
for name, net in nets.items():
    net.train()
    z = net.encode(x)[0]
    if gz is None:
        gz = torch.randn(z.shape, generator=torch.Generator().manual_seed(3)).cuda()
    (z.float() * gz).sum().backward()
    torch.cuda.synchronize()
    G[name] = {k: p.grad.detach().float().clone() for k, p in net.named_parameters() if p.grad is not None}
def fro(a, b): return float((a - b).norm() / (b.norm() + 1e-30))
for k in G["fp32"]:
    print(f"{k:28s} |g| {float(G['fp32'][k].norm()):.3e}  f16 vs fp32 {fro(G['f16'][k], G['fp32'][k]):.3e}  bf16 vs fp32 {fro(G['bf16'][k], G['fp32'][k]):.3e}  f16 vs bf16 {fro(G['f16'][k], G['bf16'][k]):.3e}")
