#!/usr/bin/env python3
"""Dev probe: where do the device copies / fills / torch elementwise launches of a Performer (or VQ-VAE) training step come from?  torch profiler with Python stacks,
grouped by (op, innermost repo frame).  usage: tools/find_copies.py [performer|vqvae]"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from torch.profiler import profile, ProfilerActivity

dev = torch.device("cuda", 0)
which = sys.argv[1] if len(sys.argv) > 1 else "performer"
if which == "performer":
    from synthanatomy_amd.losses.transformer import CELoss
    from synthanatomy_amd.networks.transformers.img2seq_ordering import Ordering
    from synthanatomy_amd.networks.transformers.performer import Performer
    from synthanatomy_amd.runtime.ddp import GradReducer
    from synthanatomy_amd.runtime.optim import FlatParams, FusedAdam
    PERF = bench.PERF
    spatial = PERF["spatial"]; N = int(np.prod(spatial)); B = 6
    torch.manual_seed(4)
    order = Ordering("raster_scan", 3, (1,) + spatial, (False, False, False), ((2, 0, 1),), ((0, 1),), ("rotate_90", "transpose"))
    net = Performer(num_tokens=PERF["vocab"] + 1, max_seq_len=N + 1, dim=PERF["dim"], depth=PERF["depth"], heads=PERF["heads"], ordering=order,
                    local_attn_heads=PERF["local_heads"], local_window_size=PERF["window"], feature_redraw_interval=1, use_rezero=True,
                    spatial_position_emb="absolute", spatial_shape=spatial, compute_dtype=torch.bfloat16).to(dev).train()
    flat = FlatParams(net.parameters()); reducer = GradReducer(flat); net.set_grad_sink(reducer)
    opt = FusedAdam(flat, lr=1e-3); opt.on_step.append(net.invalidate_packed_weights)
    loss_fn = CELoss()
    codes = torch.randint(0, PERF["vocab"], (B, N), device=dev)
    seq = codes[:, torch.as_tensor(order.get_sequence_ordering(), device=dev)]
    seq = torch.nn.functional.pad(seq, (1, 0), value=PERF["vocab"])
    x_in, x_tgt = seq[:, :-1].contiguous(), seq[:, 1:].contiguous()
    def step():
        flat.zero_grad(); logits = net(x_in); loss = loss_fn(logits.transpose(1, 2), x_tgt); loss.backward(); opt.step(grad_scale=reducer.finish())
else:
    from synthanatomy_amd.losses.vqvae import MSELoss  # noqa
    from synthanatomy_amd.networks.vqvae.baseline import BaselineVQVAE
    from synthanatomy_amd.runtime.ddp import GradReducer
    from synthanatomy_amd.runtime.optim import FlatParams, FusedAdam
    cfg = bench.VQ if hasattr(bench, "VQ") else None
    raise SystemExit("vqvae: not wired")

for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
cnt = collections.Counter()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for ev in prof.events():
    nm = ev.name
    if not (nm.startswith("aten::copy_") or nm.startswith("aten::fill_") or nm.startswith("aten::zero_") or nm in ("aten::cat", "aten::add", "aten::mul", "aten::to", "aten::_to_copy", "aten::zeros", "aten::clone", "aten::contiguous")):
        continue
    fr = [s for s in (ev.stack or []) if root in s and "find_copies" not in s]
    cnt[(nm, fr[0].replace(root + "/", "") if fr else "?")] += 1
names = collections.Counter(ev.name for ev in prof.events())
for nm, c in names.most_common(60):
    print(f"{c:5d}  {nm[:120]}")
print("----")
for (nm, fr), c in cnt.most_common(50):
    print(f"{c:5d}  {nm:18s} {fr}")
