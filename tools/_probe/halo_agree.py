import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch, numpy as np
import test_fullsize_gpu as T
from synthanatomy_amd.networks.vqvae.baseline import BaselineVQVAE
torch.manual_seed(4)
net = BaselineVQVAE(**T.NET, compute_dtype=torch.bfloat16).cuda()
g = torch.Generator(device="cuda").manual_seed(5)
x = torch.rand(1, 1, *T.VOL, generator=g, device="cuda")
def run(env):
    for k, v in env.items(): os.environ[k] = v
    try:
        net.eval()
        with torch.no_grad():
            idx = net.index_quantize(x)[0].clone()
        return T._grads(net, x), idx
    finally:
        for k in env: del os.environ[k]
(l0, g0), i0 = run({})
(l1, g1), i1 = run({"SA_NO_HALO": "1", "SA_NO_FUSED_1X1_BWD": "1"})
(l2, g2), i2 = run({})
print("index flips halo vs nohalo:", int((i0 != i1).sum()), "of", i0.numel(), " rerun:", int((i0 != i2).sum()))
def stats(a, b):
    fro = max((float((a[n].double() - b[n].double()).norm() / (b[n].double().norm() + 1e-30)), n) for n in a)
    mx = max((T._rel(a[n], b[n]), n) for n in a)
    return fro, mx
print("halo vs nohalo", stats(g0, g1)); print("halo vs halo rerun", stats(g0, g2))
