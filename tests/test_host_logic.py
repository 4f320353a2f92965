"""CPU: host-side logic of the product package (no kernels): batch preparation, sampling post-processing against the
reference fixture, flat parameter storage, bucket construction."""
import numpy as np
import torch

from conftest import load_golden
from oracle import ordering_ref


def test_prepare_batch_matches_oracle_and_hand_vectors():
    from synthanatomy_amd.networks.transformers.img2seq_ordering import Ordering
    from synthanatomy_amd.utils.transformer import prepare_batch, prepare_inference_batch
    o = Ordering("s_curve", 3, (1, 2, 3, 2), (False,) * 3, (), ())
    q = torch.arange(2 * 12, dtype=torch.int32).reshape(2, 2, 3, 2) % 7
    (x_in, cond), x_tgt = prepare_batch({"quantization": q}, o.get_sequence_ordering(), 7)
    ri, rt = ordering_ref.prepare_batch(q.numpy(), o.get_sequence_ordering(), 7)
    assert cond is None and x_in.dtype == torch.int64
    assert np.array_equal(x_in.numpy(), ri) and np.array_equal(x_tgt.numpy(), rt)
    assert x_in[0, 0].item() == 7 and x_in.shape == (2, 12) and x_tgt.shape == (2, 12)
    # hand vector: s_curve over 2x3x2 visits (0,0,0)(0,0,1)(0,1,1)(0,1,0)(0,2,0)(0,2,1)(1,2,0)(1,2,1)(1,1,1)...
    assert x_tgt[0, :8].tolist() == [0, 1, 3, 2, 4, 5, 10 % 7, 11 % 7]
    (x0, c0), t0 = prepare_inference_batch({"quantization": q, "age": torch.tensor([3, 4])}, 7, conditionings=("age",))
    assert x0.tolist() == [[7], [7]] and t0.tolist() == [[7], [7]] and c0[0].shape == (2, 1)


def test_sample_postprocessing_matches_reference_fixture():
    """TransformerBase.sample with the deterministic fake forward of tests/golden/make_goldens.py (reference output pinned)."""
    from synthanatomy_amd.networks.transformers.img2seq_ordering import Ordering
    from synthanatomy_amd.networks.transformers.transformer import TransformerBase
    g = load_golden("sample")
    o = Ordering("s_curve", 3, (1, 3, 4, 2), (False,) * 3, ((2, 0, 1),), ())
    assert np.array_equal(o.get_sequence_ordering(), g["ordering"])
    table = torch.from_numpy(g["table"])

    class Fake(TransformerBase):
        def __init__(self):
            super().__init__()
            self.ordering = o

        def forward(self, x, conditioning=None):
            pos = torch.arange(x.shape[1])
            return table[(x * 7 + pos[None, :] * 3) % 64]

    f = Fake()
    prefix = torch.full((2, 1), 11, dtype=torch.long)
    assert np.array_equal(f.sample(prefix, sample=False).numpy(), g["greedy"])
    assert np.array_equal(f.sample(prefix, sample=False, top_k=3, temperature=0.7).numpy(), g["topk"])
    torch.manual_seed(5)
    assert np.array_equal(f.sample(prefix, sample=True, top_k=4).numpy(), g["stoch_seed5_top4"])


def test_flat_params_and_buckets():
    from synthanatomy_amd.runtime.ddp import GradReducer
    from synthanatomy_amd.runtime.optim import FlatParams
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(*s)) for s in [(5, 3), (7,), (2, 2, 2), (1,), (33,)]]
    ref = [p.detach().clone() for p in ps]
    flat = FlatParams(ps)
    assert all(torch.equal(p.detach(), r) for p, r in zip(ps, ref))
    assert all(p.data_ptr() % 16 == 0 for p in ps) and flat.numel % 4 == 0
    ps[1].grad.add_(1.0)
    assert float(flat.grad.sum()) == 7.0
    flat.zero_grad()
    assert float(flat.grad.abs().sum()) == 0.0
    red = GradReducer(flat, bucket_bytes=64)
    covered = sorted((lo, hi) for lo, hi, _, _ in red.buckets)
    assert covered[0][0] == 0 and covered[-1][1] == flat.numel
    assert all(a[1] == b[0] for a, b in zip(covered, covered[1:]))  # contiguous, non-overlapping
    assert red.buckets[0][3] == len(ps) - 1  # first bucket = the LAST parameters (backward order)
    for p in reversed(ps):
        red.ready(p)
    assert red.finish() == 1.0
