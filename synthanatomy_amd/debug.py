"""Developer switches: alternative kernels / launch plans for the SAME result (A/B timing, cross-family parity tests).

Two kinds, both read from the environment ONCE (library load / module import) and afterwards changed only through ``override``:

* library flags -- the ``SA_DBG_*`` word of include/synthanatomy_hip.h (``sa_set_debug_flags``): which HIP kernel a launcher picks;
* host switches -- which launch plan the Python chains pick (fused residual block vs two launches, ...).

    with debug.override(no_halo=True, no_fused_res=True):
        y_ref = stage.fwd(x, None)
"""
from __future__ import annotations

import contextlib
import os

LIB_FLAGS = dict(no_halo=1 << 0, no_halo256=1 << 1, no_halo256_fuse=1 << 2, no_dma=1 << 3, no_small_tiles=1 << 4, no_fused_db=1 << 5,
                 no_wgrad_halo9=1 << 6, im2col_direct=1 << 7, scan_valu=1 << 8, local_attn_exact=1 << 9, halo256_4w=1 << 13, dense_narrow=1 << 15, deterministic=1 << 16, no_kgroups=1 << 17, favor_seq_always=1 << 18, no_cells256=1 << 21, no_class_launch=1 << 22)
SCAN_EXACT_SHIFT = 10   # scan_exact=0..7

_HOST_ENV = dict(no_fused_res="SA_NO_FUSED_RES", no_fused_1x1_bwd="SA_NO_FUSED_1X1_BWD", no_conv1_gemm="SA_NO_CONV1_GEMM",
                 no_conv1_fused="SA_NO_CONV1_FUSED", convt1_direct="SA_CONVT1_DIRECT", no_batched_pack="SA_NO_BATCHED_PACK",
                 no_fused_sums="SA_NO_FUSED_SUMS", no_fused_qkv="SA_NO_FUSED_QKV", no_fused_favor="SA_NO_FUSED_FAVOR", no_convt1_fused_bwd="SA_NO_CONVT1_FUSED_BWD", no_fused_epilogues="SA_NO_FUSED_EPILOGUES", no_convt1_fused_fwd="SA_NO_CONVT1_FUSED_FWD", no_lp_mirrors="SA_NO_LP_MIRRORS", no_decode_bf16_weights="SA_NO_DECODE_BF16_WEIGHTS", no_attn_step_merge="SA_NO_ATTN_STEP_MERGE", no_attn_colaunch="SA_NO_ATTN_COLAUNCH", no_side_wgrad="SA_NO_SIDE_WGRAD", no_side_wgrad_vqvae="SA_NO_SIDE_WGRAD_VQVAE",
                 ddp_single_rank="SA_DDP_SINGLE_RANK",   # collectives issued on a ONE-rank process group too (RCCL on a one-GPU box: tests/test_rccl_single_rank_gpu.py)
                 no_f16_forward="SA_NO_F16_FORWARD",     # VQ-VAE encoder forward on bf16 operands like the rest (default in throughput mode: float16, the reference's AMP dtype)
                 no_sample_step="SA_NO_SAMPLE_STEP",     # stateful sampler: the decision + sequence update as torch ops instead of sa_sample_step (A/B, equality test)
                 opt_in_backward="SA_OPT_IN_BACKWARD",   # CLIs: FusedAdam(in_backward=reducer) -- per-bucket optimizer slices + re-packs behind the gradients (measured slower on one GPU)
                 share_device="SA_SHARE_DEVICE",         # test aid: every rank on cuda:0 over gloo (RCCL refuses two ranks per device) -- the N > 1 code path of the CLIs on a one-GPU box
                 no_proj_bf16="SA_NO_PROJ_BF16",         # Performer throughput mode: keep the fp32 FAVOR+ projection operand (default: its bf16 copy -> lo(P) = 0, two products instead of three)
                 keep_ipc_mode="SA_KEEP_IPC_MODE")       # do NOT default HSA_ENABLE_IPC_MODE_LEGACY=0 (runtime/ddp.ipc_mode_default; hosts whose driver wants legacy IPC)
_host = {k: os.environ.get(v) is not None for k, v in _HOST_ENV.items()}


def host(name: str) -> bool:
    return _host[name]


def deterministic() -> bool:
    """The reference's --deterministic flag (SA_DETERMINISTIC / run_*.py --deterministic=True / debug.override(deterministic=True)): fixed-order reductions
    instead of fp32 atomics (csrc/deterministic.hip) and the unfused first / last layer routes."""
    from . import _ffi
    return bool(_ffi.lib().sa_get_debug_flags() & LIB_FLAGS["deterministic"])


def set_deterministic(on: bool = True):
    from . import _ffi
    lib = _ffi.lib()
    f = lib.sa_get_debug_flags()
    lib.sa_set_debug_flags((f | LIB_FLAGS["deterministic"]) if on else (f & ~LIB_FLAGS["deterministic"]))


@contextlib.contextmanager
def override(**kw):
    """Temporarily set switches (library flags, ``scan_exact=<0..7>`` and host switches) for the calling process."""
    from . import _ffi
    lib = _ffi.lib()
    old_flags = lib.sa_get_debug_flags()
    flags = old_flags
    old_host = dict(_host)
    for k, v in kw.items():
        if k in LIB_FLAGS:
            flags = (flags | LIB_FLAGS[k]) if v else (flags & ~LIB_FLAGS[k])
        elif k == "scan_exact":
            flags = (flags & ~(7 << SCAN_EXACT_SHIFT)) | ((int(v) & 7) << SCAN_EXACT_SHIFT)
        elif k in _host:
            _host[k] = bool(v)
        else:
            raise KeyError(f"unknown debug switch {k!r}")
    lib.sa_set_debug_flags(flags)
    try:
        yield
    finally:
        lib.sa_set_debug_flags(old_flags)
        _host.clear()
        _host.update(old_host)
