#!/usr/bin/env python3
"""Dev probe: the Performer training step (README shape, batch 6) eager vs replayed from ONE captured HIP graph (forward + CE + backward + Adam + re-pack)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from synthanatomy_amd.losses.transformer import CELoss
from synthanatomy_amd.networks.transformers.img2seq_ordering import Ordering
from synthanatomy_amd.networks.transformers.performer import Performer
from synthanatomy_amd.runtime.ddp import GradReducer
from synthanatomy_amd.runtime.optim import FlatParams, FusedAdam

dev = torch.device("cuda", 0)
PERF = bench.PERF
spatial = PERF["spatial"]; N = int(np.prod(spatial)); B = 6
torch.manual_seed(4)
order = Ordering("raster_scan", 3, (1,) + spatial, (False, False, False), ((2, 0, 1),), ((0, 1),), ("rotate_90", "transpose"))
net = Performer(num_tokens=PERF["vocab"] + 1, max_seq_len=N + 1, dim=PERF["dim"], depth=PERF["depth"], heads=PERF["heads"], ordering=order,
                local_attn_heads=PERF["local_heads"], local_window_size=PERF["window"], feature_redraw_interval=int(os.environ.get("REDRAW", "1")), use_rezero=True,
                spatial_position_emb="absolute", spatial_shape=spatial, compute_dtype=torch.bfloat16).to(dev).train()
flat = FlatParams(net.parameters())
reducer = GradReducer(flat)
net.set_grad_sink(reducer)
opt = FusedAdam(flat, lr=1e-3)
opt.on_step.append(net.invalidate_packed_weights)
loss_fn = CELoss()
gen = torch.Generator(device=dev).manual_seed(4)
codes = torch.randint(0, PERF["vocab"], (B, N), generator=gen, device=dev)
seq = codes[:, torch.as_tensor(order.get_sequence_ordering(), device=dev)]
seq = torch.nn.functional.pad(seq, (1, 0), value=PERF["vocab"])
x_in, x_tgt = seq[:, :-1].contiguous(), seq[:, 1:].contiguous()
loss_buf = torch.zeros((), device=dev)

def step():
    flat.zero_grad()
    logits = net(x_in)
    loss = loss_fn(logits.transpose(1, 2), x_tgt)
    loss.backward()
    opt.step(grad_scale=reducer.finish())
    loss_buf.copy_(loss.detach())

def timeit(fn, n=30):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

for _ in range(8): step()
print("eager ms/step", round(timeit(step), 3), "loss", float(loss_buf))
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        step()
    for _ in range(5): g.replay()
    print("graph ms/step", round(timeit(g.replay), 3), "loss", float(loss_buf))
except Exception as e:
    print("capture failed:", type(e).__name__, str(e)[:600])
