"""``sa_comm_*`` from Python: RCCL through the C ABI (csrc/comm.hip), for hosts that do not bring ``torch.distributed``.

The training loops of this repository reduce gradients and EMA statistics through ``torch.distributed`` (``runtime/ddp.py``: backend ``nccl`` = RCCL); this class is
the same exchange as a non-torch host would bind it -- the 128-byte id of rank 0 handed to the other ranks by the host, one communicator per process, every
collective enqueued on the caller's HIP stream.  ``tests/test_comm_gpu.py`` runs it on a one-rank communicator (the test pool has one GPU per box)."""
from __future__ import annotations

import ctypes

import torch

from .. import _ffi


class NativeComm:
    ID_BYTES = 128

    @staticmethod
    def unique_id() -> bytes:
        buf = ctypes.create_string_buffer(NativeComm.ID_BYTES)
        NativeComm._ck(_ffi.lib().sa_comm_unique_id(buf), "sa_comm_unique_id")
        return buf.raw

    @staticmethod
    def _ck(rc, what):
        if rc != 0:
            msg = _ffi.lib().sa_comm_last_error()
            raise RuntimeError(f"{what} failed (rc {rc}): {msg.decode() if msg else ''}")

    def __init__(self, uid: bytes, rank: int, world: int):
        assert len(uid) == self.ID_BYTES
        self._h = ctypes.c_void_p()
        self._ck(_ffi.lib().sa_comm_init(ctypes.byref(self._h), ctypes.create_string_buffer(uid, self.ID_BYTES), rank, world), "sa_comm_init")
        self.rank, self.world = _ffi.lib().sa_comm_rank(self._h), _ffi.lib().sa_comm_world(self._h)

    def all_reduce_sum(self, t: torch.Tensor):
        assert t.is_cuda and t.is_contiguous()
        self._ck(_ffi.lib().sa_comm_all_reduce_sum(self._h, _ffi.ptr(t), t.numel(), _ffi.dtype_id(t.dtype), _ffi.stream()), "sa_comm_all_reduce_sum")
        return t

    def reduce_scatter_sum(self, send: torch.Tensor, recv: torch.Tensor):
        assert send.is_contiguous() and recv.is_contiguous() and send.numel() == recv.numel() * self.world and send.dtype == recv.dtype
        self._ck(_ffi.lib().sa_comm_reduce_scatter_sum(self._h, _ffi.ptr(send), _ffi.ptr(recv), recv.numel(), _ffi.dtype_id(recv.dtype), _ffi.stream()),
                 "sa_comm_reduce_scatter_sum")
        return recv

    def all_gather(self, send: torch.Tensor, recv: torch.Tensor):
        assert send.is_contiguous() and recv.is_contiguous() and recv.numel() == send.numel() * self.world and send.dtype == recv.dtype
        self._ck(_ffi.lib().sa_comm_all_gather(self._h, _ffi.ptr(send), _ffi.ptr(recv), send.numel(), _ffi.dtype_id(send.dtype), _ffi.stream()), "sa_comm_all_gather")
        return recv

    def close(self):
        if self._h:
            self._ck(_ffi.lib().sa_comm_destroy(self._h), "sa_comm_destroy")
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
