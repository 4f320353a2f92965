#!/usr/bin/env python3
"""Dev probe: what does WRITING an [8400, 512] fp32 tile set (17 MB) cost as a kernel of its own?  (fill, copy and a bf16 -> fp32 cast through torch; read the
kernel durations from rocprofv3 --kernel-trace.)"""
import torch
x = torch.empty(8400, 512, device="cuda")
y = torch.randn(8400, 512, device="cuda")
h = torch.randn(8400, 512, device="cuda").to(torch.bfloat16)
big = torch.empty(8400, 3072, device="cuda")
for _ in range(20):
    x.zero_()
    x.copy_(y)
    x.copy_(h)
    big.zero_()
torch.cuda.synchronize()
