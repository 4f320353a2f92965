import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import test_performer_gpu as T
P = T.P
def run(rezero, B, shape, window, local):
    n = int(np.prod(shape))
    cfg = P.PerformerConfig(num_tokens=33, max_seq_len=n, dim=32, depth=2, heads=4, dim_head=64, local_attn_heads=local, local_window_size=window, spatial_shape=shape, use_rezero=rezero)
    st = P.init_state(cfg, seed=n)
    if rezero:
        for k in st:
            if k.endswith(".g"): st[k] = torch.tensor(0.4)
    net, o = T._build(cfg, st, rezero=rezero); net.train()
    seqs = P.spatial_index_sequences(shape, o.get_sequence_ordering())
    torch.manual_seed(0)
    tok = torch.randint(0, 33, (B, n)); tgt = torch.randint(0, 32, (B, n))
    leaf = {k: v.clone().requires_grad_(True) for k, v in st.items() if "projection_matrix" not in k}
    stt = dict(st); stt.update(leaf)
    ref = P.forward(stt, cfg, tok, seqs); P.ce_loss(ref, tgt).backward()
    from synthanatomy_amd.losses.transformer import CELoss
    out = net(tok.cuda()); loss = CELoss()(out.transpose(1, 2), tgt.cuda()); loss.backward(); torch.cuda.synchronize()
    params = dict(net.named_parameters())
    errs = sorted(((T._rel(params[k].grad, p.grad), k) for k, p in leaf.items() if p.grad is not None), reverse=True)
    return T._rel(out, ref), errs[:3]
for ex in ("7", "6", "5", "3", "0"):
    os.environ["SA_SCAN_EXACT"] = ex
    for case in [(True, 2, (2, 3, 3), 5, 0), (True, 2, (2, 3, 4), 6, 2)]:
        o, e = run(*case)
        print("EXACT", ex, case, "out %.1e" % o, [(round(x, 5), k.split("layers.")[-1]) for x, k in e], flush=True)
