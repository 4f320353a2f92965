import sys
sys.path.insert(0, "/root/repo")
import torch
from synthanatomy_amd import engine
for R, K, N, odt in [(8400, 64, 128, torch.bfloat16), (8400, 64, 512, torch.bfloat16), (8400, 64, 2048, torch.bfloat16), (2100, 64, 512, torch.bfloat16), (128, 64, 64, torch.bfloat16),
                     (8400, 64, 512, torch.float32)]:
    w = torch.randn(N, K, 1, 1, 1, device="cuda") * K ** -0.5
    op = engine.ConvOp("conv", K, N, 1, 1, 0, w, None, torch.bfloat16)
    x = torch.randn(1, 1, 1, R, K, device="cuda").bfloat16()
    for _ in range(5):
        op.fprop(x, out_dtype=odt, use_bias=False)
    torch.cuda.synchronize()
