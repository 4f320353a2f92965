import sys
sys.path.insert(0, "/root/repo")
import torch
from synthanatomy_amd import engine
R = 8400
for K, N in [(512, 512), (512, 1536), (512, 2048), (2048, 512)]:
    w = torch.randn(N, K, 1, 1, 1, device="cuda") * K ** -0.5
    op = engine.ConvOp("conv", K, N, 1, 1, 0, w, torch.zeros(N, device="cuda"), torch.bfloat16)
    x = torch.randn(1, 1, 1, R, K, device="cuda").bfloat16()
    g = torch.randn(1, 1, 1, R, N, device="cuda").bfloat16()
    dw = torch.zeros_like(w); db = torch.zeros(N, device="cuda")
    for _ in range(5):
        op.wgrad(x, g, dw, db)
    torch.cuda.synchronize()
