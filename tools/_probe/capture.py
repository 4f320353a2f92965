import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
warnings.simplefilter("always")
from synthanatomy_amd.networks.transformers.img2seq_ordering import Ordering
from synthanatomy_amd.networks.transformers.performer import Performer
spatial = (2, 3, 4)
order = Ordering("raster_scan", 3, (1,) + spatial, (False, False, False), (), ())
net = Performer(num_tokens=19, max_seq_len=24, dim=32, depth=2, heads=4, ordering=order, local_attn_heads=1, local_window_size=5, feature_redraw_interval=None,
                use_rezero=True, spatial_position_emb="absolute", spatial_shape=spatial).cuda().eval()
prefix = torch.full((2, 1), 18, dtype=torch.long, device="cuda")
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    out = net.sample(prefix, sample=True)
    print('plain sample warnings', len(w))
    out = net.sample(prefix, sample=True, top_k=5, temperature=0.9)
    print('topk sample warnings', len(w))
    for x in w:
        print("WARNING:", str(x.message)[:1500])
print("done", out.shape)
