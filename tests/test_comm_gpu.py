"""``sa_comm_*`` (csrc/comm.hip): RCCL behind the C ABI, on a ONE-rank communicator (one GPU per test box; RCCL between two GPUs has not run -- DESIGN section 5).
With one rank every collective is the identity on the data, which still exercises the dlopen'd API, the communicator life cycle, the dtype mapping and the stream
ordering (the collective reads what the preceding kernel on the stream wrote)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_one_rank_collectives_through_the_c_abi():
    from synthanatomy_amd.runtime.comm import NativeComm
    uid = NativeComm.unique_id()
    assert len(uid) == 128 and any(uid)
    comm = NativeComm(uid, 0, 1)
    assert (comm.rank, comm.world) == (0, 1)
    for dt in (torch.float32, torch.bfloat16, torch.float16):
        x = torch.randn(1 << 20, device="cuda").to(dt)
        ref = (x.float() * 2).to(dt)
        y = x.mul(2)                       # enqueued on the same stream in front of the collective
        comm.all_reduce_sum(y)
        assert torch.equal(y, ref)
        out = torch.empty_like(y)
        comm.reduce_scatter_sum(y, out)
        assert torch.equal(out, ref)
        out2 = torch.zeros_like(y)
        comm.all_gather(y, out2)
        torch.cuda.synchronize()
        assert torch.equal(out2, ref)
    # the EMA statistics exchange of the quantizer as a non-torch host would issue it: ONE all-reduce of the packed [K + K D] buffer
    K, D = 2048, 32
    stats = torch.rand(K + K * D, device="cuda")
    keep = stats.clone()
    comm.all_reduce_sum(stats)
    torch.cuda.synchronize()
    assert torch.equal(stats, keep)
    comm.close()
    with pytest.raises(AssertionError):
        NativeComm(b"short", 0, 1)


def test_argument_checks_do_not_touch_rccl():
    from synthanatomy_amd import _ffi
    lib = _ffi.lib()
    assert lib.sa_comm_unique_id(None) == _ffi.SA_EINVAL
    assert lib.sa_comm_all_reduce_sum(None, None, 0, 0, None) == _ffi.SA_EINVAL
    assert lib.sa_comm_rank(None) == _ffi.SA_EINVAL
