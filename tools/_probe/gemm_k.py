import sys, os
sys.path.insert(0, "/root/repo")
import torch
from synthanatomy_amd import engine
R = 8400
for K, N, odt in [(64, 512, torch.float32), (128, 512, torch.float32), (256, 512, torch.float32), (512, 512, torch.float32), (1024, 512, torch.float32), (2048, 512, torch.float32),
                  (512, 512, torch.bfloat16), (512, 1536, torch.float32), (512, 2048, torch.bfloat16)]:
    w = torch.randn(N, K, 1, 1, 1, device="cuda") * K ** -0.5
    op = engine.ConvOp("conv", K, N, 1, 1, 0, w, torch.zeros(N, device="cuda"), torch.bfloat16)
    x = torch.randn(1, 1, 1, R, K, device="cuda").bfloat16()
    for _ in range(5):
        op.fprop(x, out_dtype=odt)
    torch.cuda.synchronize()
