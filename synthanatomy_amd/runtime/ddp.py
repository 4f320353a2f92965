"""Batch-sharded data parallelism over RCCL/xGMI, one process per GPU.

Replaces the reference's ``DistributedDataParallel(..., bucket_cap_mb=12.5)`` wrappers (run_vqvae.py:71-77,
run_transformer.py:98-105).  Gradients are written by the wgrad kernels straight into the flat gradient buffer
(runtime/optim.py); ``GradReducer`` cuts that buffer into a few large contiguous buckets in BACKWARD order and launches
one asynchronous all-reduce per bucket on a side stream as soon as every parameter of the bucket has been produced, so
the reduction overlaps the remaining backward kernels.  xGMI is a point-to-point mesh (7 links x ~153 GB/s per GPU): a
ring all-reduce is bound by one link, so few large buckets (default 32 MiB) beat DDP's many 12.5 MiB ones.  The 1/world
averaging is folded into the Adam kernel's ``grad_scale``.
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch
import torch.distributed as dist

from .. import debug
from .optim import FlatParams


def ipc_mode_default(env=None, verbose: bool = True) -> bool:
    """RCCL between processes exchanges device-memory handles; the MI355X hosts this build targets only support dmabuf IPC, which the HIP runtime
    selects when ``HSA_ENABLE_IPC_MODE_LEGACY=0`` is in the environment BEFORE its first device call (without it: ``hipIpcGetMemHandle: invalid
    argument`` in the first collective).  Applied only when the variable is unset and ``SA_KEEP_IPC_MODE`` is not given, and said so once on rank 0
    -- a host whose driver wants legacy IPC exports ``HSA_ENABLE_IPC_MODE_LEGACY=1`` (respected) or ``SA_KEEP_IPC_MODE=1``.  Returns whether the
    default was applied.  Documented in INTEGRATION.md ("Multi-process environment")."""
    env = os.environ if env is None else env
    if "HSA_ENABLE_IPC_MODE_LEGACY" in env or debug.host("keep_ipc_mode"):
        return False
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    if verbose and int(env.get("WORLD_SIZE", "1")) > 1 and int(env.get("RANK", "0")) == 0:
        print("[synthanatomy_amd] HSA_ENABLE_IPC_MODE_LEGACY was unset: defaulting to 0 (dmabuf IPC) for RCCL; SA_KEEP_IPC_MODE=1 leaves it alone", flush=True)
    return True


DEFAULT_TIMEOUT_S = 300.0


def init_distributed(backend: Optional[str] = None, timeout_s: Optional[float] = None):
    """torchrun-style bootstrap (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*); replaces deepspeed.init_distributed
    (run_vqvae.py:831-842).  Returns (rank, local_rank, world_size).  ``SA_SHARE_DEVICE=1`` (test aid) puts every rank on cuda:0 over gloo:
    the N > 1 code path of the CLIs on a one-GPU box (RCCL refuses two ranks per device).

    Failure detection (the one knob the reference leaves at its default): every collective carries ``timeout_s`` (argument, else
    ``SA_DIST_TIMEOUT_S``, else 300 s instead of torch's 10 / 30 minutes), and the RCCL watchdog is told to tear the process down on an asynchronous
    error or a timed-out collective (``TORCH_NCCL_ASYNC_ERROR_HANDLING=1`` unless the caller exported another policy) -- so when a rank dies the others
    FAIL within the timeout (gloo raises in the waiting call; RCCL aborts the communicator and the process) and the launcher (torchrun) ends the job,
    instead of seven ranks spinning in an all-reduce kernel.  ``tests/test_ddp_gloo.py::test_dead_rank_fails_the_others_within_the_timeout``."""
    import datetime
    ipc_mode_default()
    if timeout_s is None:
        raw = os.environ.get("SA_DIST_TIMEOUT_S")
        try:
            timeout_s = float(raw) if raw not in (None, "") else DEFAULT_TIMEOUT_S
        except ValueError:
            raise ValueError(f"SA_DIST_TIMEOUT_S={raw!r}: expected the collective timeout in seconds (a number > 0)") from None
    if not timeout_s > 0:
        raise ValueError(f"collective timeout must be > 0 seconds, got {timeout_s!r} (SA_DIST_TIMEOUT_S / init_distributed(timeout_s=...))")
    os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if debug.host("share_device"):
        local = 0
        backend = backend or "gloo"
    single = world == 1 and "RANK" in os.environ and "MASTER_PORT" in os.environ and debug.host("ddp_single_rank")   # one-GPU box: RCCL on a one-rank group
    if (world > 1 or single) and not dist.is_initialized():
        if torch.cuda.is_available():
            # load (or, on a cold node, build) the HIP library BEFORE the rendezvous: a rank that compiles for minutes behind init_process_group would run the
            # others into the collective timeout
            from .. import _ffi
            _ffi.lib()
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"  # "nccl" is RCCL on ROCm
        if torch.cuda.is_available():
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=timeout_s))
    elif torch.cuda.is_available():
        torch.cuda.set_device(local)
    return rank, local, world


_HOST_STAGED = {}   # backend name -> True when collectives on device tensors must be staged through host memory


def all_reduce_sum(t: torch.Tensor, group=None):
    """SUM all-reduce of `t` in place on the CURRENT stream.  RCCL ("nccl") reduces device memory directly.  The gloo backend is only used
    by the tests that run several ranks on ONE device (RCCL refuses duplicate devices): it reduces device tensors through its own pinned
    staging when the build supports that, else through an explicit host copy here."""
    if not t.is_cuda or dist.get_backend(group) != "gloo":
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return
    staged = _HOST_STAGED.get("gloo")
    if staged is None:
        try:
            probe = torch.zeros(1, device=t.device)
            dist.all_reduce(probe, op=dist.ReduceOp.SUM, group=group)
            staged = False
        except RuntimeError:
            staged = True
        _HOST_STAGED["gloo"] = staged
    if not staged:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return
    h = t.detach().cpu()          # synchronises the current stream: everything queued before the collective has been produced
    dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
    t.copy_(h)


def _reduce_scatter_sum(out: torch.Tensor, inp: torch.Tensor, group=None):
    """``out`` = this rank's 1/world slice of the SUM of ``inp`` over ranks (``out`` may alias that slice of ``inp``: the in-place form RCCL
    reduces without a staging copy).  gloo has no reduce-scatter: the CPU / shared-device tests get the same result from an all-reduce."""
    if dist.get_backend(group) == "gloo":
        all_reduce_sum(inp, group)
        if out.data_ptr() != inp.data_ptr() + dist.get_rank(group) * out.numel() * out.element_size():
            r = dist.get_rank(group)
            out.copy_(inp[r * out.numel():(r + 1) * out.numel()])
        return
    dist.reduce_scatter_tensor(out, inp, op=dist.ReduceOp.SUM, group=group)


def _all_gather(out: torch.Tensor, inp: torch.Tensor, group=None):
    """``out`` = the ranks' ``inp`` slices in rank order (``inp`` may alias this rank's slice of ``out``)."""
    if dist.get_backend(group) == "gloo":
        if out.is_cuda and _HOST_STAGED.get("gloo"):
            h = [torch.empty(inp.numel(), dtype=inp.dtype) for _ in range(dist.get_world_size(group))]
            dist.all_gather(h, inp.detach().cpu(), group=group)
            out.copy_(torch.cat(h))
            return
        parts = [torch.empty_like(inp) for _ in range(dist.get_world_size(group))]
        dist.all_gather(parts, inp.clone(), group=group)
        out.copy_(torch.cat(parts))
        return
    dist.all_gather_into_tensor(out, inp, group=group)


MODES = ("all_reduce", "reduce_scatter")
TRANSPORTS = ("fp32", "bf16")


class GradReducer:
    """Gradient sink of the backward chains + the bucketed reduction over ranks.

    ``mode``      ``"all_reduce"`` (default; what DDP issues, run_vqvae.py:72-77): one SUM all-reduce per bucket.
                  ``"reduce_scatter"``: every bucket is reduce-scattered (each rank receives the sum of its 1/world slice) and the reduced slices are
                  all-gathered back -- on the fully connected xGMI mesh both halves are direct peer-to-peer exchanges over all 7 links at once instead
                  of a ring bound by one link; the flat buffer ends up with the same sums, so the optimizer is unchanged.
    ``transport`` ``"fp32"`` (default) or ``"bf16"``: the bucket crosses the links as bf16 (half the bytes; the sum is formed in bf16, so gradients carry
                  bf16 rounding -- opt-in).
    Environment defaults: ``SA_DDP_MODE``, ``SA_DDP_TRANSPORT`` (read at construction)."""

    def __init__(self, flat: FlatParams, bucket_bytes: int = 32 << 20, process_group=None, mode: Optional[str] = None, transport: Optional[str] = None):
        self.flat, self.group = flat, process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.mode = mode or os.environ.get("SA_DDP_MODE", "all_reduce")
        self.transport = transport or os.environ.get("SA_DDP_TRANSPORT", "fp32")
        if self.mode not in MODES or self.transport not in TRANSPORTS:
            raise ValueError(f"GradReducer: mode {self.mode!r} / transport {self.transport!r}; choices are {MODES} / {TRANSPORTS}")
        # buckets are contiguous [lo, hi) ranges of the flat buffer, built from the END (backward produces the last
        # parameters first)
        n = len(flat.params)
        ends = [flat.offsets[i + 1] if i + 1 < n else flat.numel for i in range(n)]
        self.buckets: List[List[int]] = []   # [lo, hi, first_param, last_param]
        hi_p = n - 1
        while hi_p >= 0:
            lo_p = hi_p
            while lo_p > 0 and (ends[hi_p] - flat.offsets[lo_p - 1]) * 4 <= bucket_bytes:
                lo_p -= 1
            self.buckets.append([flat.offsets[lo_p], ends[hi_p], lo_p, hi_p])
            hi_p = lo_p - 1
        self.bucket_of = {}
        for b, (_, _, lo_p, hi_p) in enumerate(self.buckets):
            for i in range(lo_p, hi_p + 1):
                self.bucket_of[i] = b
        self._pending = None
        self._side = None
        self._works = []
        self._lp = None          # bf16 transport: one staging buffer of the largest bucket's size (buckets are reduced one after the other on the side stream)
        self.on_bucket = None    # optional callback(lo, hi, scale) run on the side stream after a bucket's gradients are final (reduced): the optimizer's
                                 # slice for that range (runtime/optim.FusedAdam in_backward mode), so that nothing but the last bucket is serial after backward.
                                 # It WRITES parameters and packed operands, which the data-gradient launches of the bucket's last layer still read -- and a chain
                                 # reports a layer's parameters before it queues that layer's data gradient -- so the call is deferred to the next ready() /
                                 # finish(): by then everything that reads the bucket's parameters has been queued, and the side stream waits for it
        self._deferred = []
        self._final = []         # optimizer slices of buckets completed by a report that did NOT follow the flush-then-ready discipline: released in finish() only
        self._report = 0         # token handed out by flush(); ready(p, report) proves the caller flushed before this layer's report
        self._tainted = set()    # buckets that received a ready() without the current token
        self.timing = False      # bench.py: record HIP events around every bucket's collective and around the wait in finish()
        self._ev = []            # per finished step: (bucket (start, end) events on the side stream, main-stream arrival, side-stream end)
        self._bucket_ev = []
        self.reset()

    @property
    def active(self) -> bool:
        """True when a reported bucket launches a collective (more than one rank, or the one-rank RCCL test mode): the backward chains ask this to
        decide from which stream a bucket's parameters are reported."""
        return self.world > 1 or (dist.is_initialized() and debug.host("ddp_single_rank")) or self.on_bucket is not None

    def reset(self):
        self._pending = [hi - lo + 1 for (_, _, lo, hi) in self.buckets]
        self._works = []
        self._tainted = set()

    # ---- gradient sink protocol used by the backward chains --------------------------------------------------
    def buffer(self, p: torch.nn.Parameter) -> Optional[torch.Tensor]:
        if id(p) not in self.flat.index:
            return None
        return self.flat.grad_view(p)

    def _flush_deferred(self):
        if not self._deferred:
            return
        todo, self._deferred = self._deferred, []
        cuda = self.flat.grad.is_cuda
        if cuda:
            if self._side is None:
                self._side = torch.cuda.Stream(device=self.flat.grad.device)
            ev = torch.cuda.Event()
            ev.record()
            with torch.cuda.stream(self._side):
                self._side.wait_event(ev)
                for lo, hi, scale in todo:
                    self.on_bucket(lo, hi, scale)
        else:
            for lo, hi, scale in todo:
                self.on_bucket(lo, hi, scale)

    def flush(self):
        """Called by a backward chain at the START of a layer's report (before its ready() calls): everything queued so far belongs to layers whose launches
        are complete, so the optimizer slices deferred by earlier reports may go.  (A bucket boundary can fall between the weight and the bias of ONE layer:
        flushing inside ready() would step that layer's weight before its data-gradient launch, which reads the packed weight, has been queued.)
        Returns the report token to pass to this layer's ready() calls."""
        self._flush_deferred()
        self._report += 1
        return self._report

    def ready(self, p: torch.nn.Parameter, report: Optional[int] = None):
        """``report``: the token of the flush() that opened this layer's report.  The optimizer-in-backward path (``on_bucket``) overwrites parameters and
        packed operands behind a bucket's gradients; that is only safe early for buckets whose EVERY parameter was reported flush-then-ready by a chain
        that queues nothing reading them afterwards.  A bucket that receives a ready() without the current token (autograd hooks of the embeddings, the
        adversarial trainer, any third-party sink user) keeps its optimizer slice until finish()."""
        i = self.flat.index.get(id(p))
        if i is None:
            return
        b = self.bucket_of[i]
        if report is None or report != self._report:
            self._tainted.add(b)
        self._pending[b] -= 1
        if self._pending[b] == 0:
            self._launch(b)

    def _defer(self, b: int, lo: int, hi: int, scale: float):
        (self._final if b in self._tainted else self._deferred).append((lo, hi, scale))

    def _collective(self, view: torch.Tensor):
        """The bucket's reduction on the CURRENT stream (device) or synchronously (host tensors of the gloo tests)."""
        buf = view
        if self.transport == "bf16":
            if self._lp is None or self._lp.device != view.device:
                self._lp = torch.empty(max(hi - lo for lo, hi, _, _ in self.buckets), dtype=torch.bfloat16, device=view.device)
            buf = self._lp[:view.numel()]
            buf.copy_(view)
        if self.mode == "all_reduce" or self.world == 1:
            all_reduce_sum(buf, self.group)
        else:
            w = self.world
            c = buf.numel() // w
            if c:
                r = dist.get_rank(self.group)
                body = buf[:c * w]
                mine = body[r * c:(r + 1) * c]
                _reduce_scatter_sum(mine, body, self.group)
                _all_gather(body, mine, self.group)
            if buf.numel() > c * w:            # fewer than `world` trailing elements
                all_reduce_sum(buf[c * w:], self.group)
        if buf is not view:
            view.copy_(buf)

    def _collects(self) -> bool:
        return self.world > 1 or (dist.is_initialized() and debug.host("ddp_single_rank"))

    def _launch(self, b: int):
        if not self.active:
            return
        lo, hi = self.buckets[b][0], self.buckets[b][1]
        view = self.flat.grad[lo:hi]
        if not self._collects():          # one rank, optimizer in backward: no collective, only the bucket's (deferred) optimizer slice
            self._defer(b, lo, hi, 1.0)
            return
        if view.is_cuda:
            if self._side is None:
                self._side = torch.cuda.Stream(device=view.device)
            ev = torch.cuda.Event()
            ev.record()
            with torch.cuda.stream(self._side):
                self._side.wait_event(ev)
                if self.timing:
                    e0 = torch.cuda.Event(enable_timing=True)
                    e0.record()
                self._collective(view)
                if self.timing:
                    e1 = torch.cuda.Event(enable_timing=True)
                    e1.record()
                    self._bucket_ev.append((e0, e1))
            if self.on_bucket is not None:
                self._defer(b, lo, hi, 1.0 / self.world)
        elif self.mode == "all_reduce" and self.transport == "fp32" and self.on_bucket is None:
            self._works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            self._collective(view)
            if self.on_bucket is not None:
                self._defer(b, lo, hi, 1.0 / self.world)

    def finish(self) -> float:
        """Wait for every bucket (launching any bucket whose parameters never reported, e.g. unused ones) and return the
        scale (1/world) the optimizer must apply."""
        for b, left in enumerate(self._pending):
            if left > 0:
                self._pending[b] = 0
                self._tainted.add(b)
                self._launch(b)
        self._deferred += self._final
        self._final = []
        self._flush_deferred()
        if self._side is not None:
            if self.timing and self._bucket_ev:
                arrive = torch.cuda.Event(enable_timing=True)
                arrive.record()                      # the main stream has queued everything of backward and now has to wait
                self._ev.append((self._bucket_ev, arrive, self._bucket_ev[-1][1]))
                self._bucket_ev = []
            torch.cuda.current_stream().wait_stream(self._side)
        for w in self._works:
            w.wait()
        self.reset()
        return 1.0 / self.world

    def comm_stats(self):
        """After a device synchronisation: per-step averages of the collectives' time on the side stream (`comm_ms`), the part of it the main
        stream had to wait for after backward had been queued (`exposed_ms`) and the rest (`hidden_ms`, overlapped with backward kernels)."""
        if not self._ev:
            return None
        tot = exp = 0.0
        for buckets, arrive, end in self._ev:
            tot += sum(a.elapsed_time(b) for a, b in buckets)
            exp += max(0.0, arrive.elapsed_time(end))
        n = len(self._ev)
        self._ev = []
        exp = min(exp, tot)
        return {"steps": n, "comm_ms": round(tot / n, 3), "exposed_ms": round(exp / n, 3), "hidden_ms": round((tot - exp) / n, 3),
                "buckets": len(self.buckets), "bytes_per_step": int(self.flat.numel * (2 if self.transport == "bf16" else 4)),
                "mode": self.mode, "transport": self.transport}
