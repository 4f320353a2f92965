#!/usr/bin/env python3
"""Dev: wall time of Performer.sample() (README model, N = 1400) -- reference-faithful O(N^2) loop vs the stateful O(N) path."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from synthanatomy_amd.networks.transformers.img2seq_ordering import Ordering
from synthanatomy_amd.networks.transformers.performer import Performer

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=2)
ap.add_argument("--shape", default="10,14,10")
ap.add_argument("--depth", type=int, default=24)
ap.add_argument("--mode", default="both")
a = ap.parse_args()
spatial = tuple(int(v) for v in a.shape.split(","))
N = int(np.prod(spatial))
torch.manual_seed(4)
order = Ordering("raster_scan", 3, (1,) + spatial, (False, False, False), ((2, 0, 1),), ((0, 1),), ("rotate_90", "transpose"))
net = Performer(num_tokens=2049, max_seq_len=N, dim=512, depth=a.depth, heads=16, ordering=order, local_attn_heads=8, local_window_size=420,
                feature_redraw_interval=1, use_rezero=True, spatial_position_emb="absolute", spatial_shape=spatial, compute_dtype=torch.bfloat16).cuda().eval()
prefix = torch.full((a.batch, 1), 2048, dtype=torch.long, device="cuda")
for mode in (["quadratic", "stateful"] if a.mode == "both" else a.mode.split(",")):
    kw = {"stateful": mode != "quadratic"}
    if mode == "eager":
        kw["use_graph"] = False
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = net.sample(prefix, sample=False, **kw)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{mode}: {dt:.2f} s for {a.batch} x {N} tokens = {a.batch * N / dt:.1f} tokens/s; checksum {int(out.sum())}", flush=True)
