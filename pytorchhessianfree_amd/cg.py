"""Drop-in ``cg`` for the reference's ``hessianfree/cg.py:9-231`` driving the HIP
PCG kernels of ``libhfpcg.so``.

Same signature, same return value ``(x_iters, m_iters, reason)``, same warnings.
What is different is *where* the work happens:

* ``x, r, p`` are persistent contiguous HBM vectors updated in place by three
  fused kernels per iteration (``hf_pcg_iterate``) instead of ~25 ATen launches
  that allocate a fresh N-vector each (cg.py:205-224);
* ``alpha``, ``beta``, ``||r||``, ``m_i`` and all four termination tests
  (cg.py:95-115) live on the device -- the host never reads a scalar inside the
  loop; it polls a pinned flag the device sets on termination and bounds its
  run-ahead with a lagged event.  Kernels enqueued after termination are no-ops,
  so the returned iterate is exactly the terminating one;
* iterates the caller asked for (``store_x_at_iters``) are written into a slab
  by the update kernel itself (the reference gets them for free from its
  allocate-per-iteration style, cg.py:208-210).

Fast paths, chosen by operator TYPE (any other callable still works):

* ``A`` is a :class:`DampedCurvature` -> the damping ``+ lambda*p``
  (optimizer.py:266) is fused into K1/K2;
* ``M`` is a :class:`DiagonalPreconditioner` -> ``(diag+lambda)^-alpha`` is
  built once (preconditioners.py:124 recomputes the power every call) and the
  multiply is fused into K2/K3.

There is NO CPU fallback: CPU tensors raise ``RuntimeError``.
"""

import ctypes
import os
import weakref
from math import ceil, log
from warnings import warn

import torch

from . import _lib

_LAG = 2  # how many iterations the host may run ahead of the device
# ... and when one iteration is a single graph launch of a whole curvature product (>= 0.1 ms): every
# iteration enqueued beyond the terminating one is a wasted product -- one in flight keeps the device busy
# (measured: headline unchanged, a default step() of config 2 0.6 ms shorter)
_LAG_FUSED = 1


class DampedCurvature:
    """``x -> mvp(x) + damping * x`` (the ``A`` of optimizer.py:266) as an object,
    so that :func:`cg` can fuse the damping into its kernels."""

    def __init__(self, mvp, damping, lockstep=False):
        self.mvp = mvp
        self.damping = float(damping)
        # data-parallel callers: every rank must leave the PCG loop at the same
        # iteration (``mvp`` ends in a collective even if it does not say so)
        self.lockstep = bool(lockstep)

    def __call__(self, x):
        return self.mvp(x) + self.damping * x


class DiagonalPreconditioner:
    """``x -> (diag + damping)^-exponent * x`` (preconditioners.py:108-127) with
    the power evaluated once by the ``hf_precond_build`` kernel."""

    def __init__(self, diag_vec, damping, exponent=0.75):
        self.diag = diag_vec
        self.damping = float(damping)
        self.exponent = float(exponent)
        if diag_vec.is_cuda:
            self.minv = torch.empty_like(diag_vec, memory_format=torch.contiguous_format)
            _lib.precond_build(self.minv, diag_vec.contiguous(), damping, exponent)
        else:  # host-side use (e.g. unit tests of the recipe); cg() itself needs a GPU
            self.minv = (diag_vec + damping) ** -exponent

    def __call__(self, x):
        return torch.mul(self.minv, x)


def storing_grid(max_iter, gamma=1.3):
    """Iterations ``ceil(gamma^j) - 1`` at which iterates are kept for
    CG-backtracking (cg.py:152-170).  The powers are taken in float32 by torch on
    an integer ``arange`` exactly as the reference does -- a double-precision
    table differs."""
    if gamma < 1.0:
        raise ValueError(f"Invalid gamma = {gamma}")
    j_max = ceil(log(max_iter + 1) / log(gamma))
    js = torch.arange(j_max + 1)
    return sorted(set((torch.ceil(gamma**js) - 1).int().tolist()))


class _IterationGraph:
    """``[curvature product] -> K1 -> K2 -> K3`` as ONE hipGraph launch per PCG
    iteration (``hf_pcg_graph_*``).  Built once per (operator, workspace,
    preconditioner mode) from the operator's captured product graph (or without a
    product, for data-parallel runs where an all-reduce separates the two halves);
    the K1-K3 kernel arguments are refreshed at the start of every solve."""

    def __init__(self, ws, raw_graph, Bp, damping, mode, timing):
        self.ws, self.mode, self.timing = ws, mode, bool(timing)
        self.handle = _lib.c_void_p()
        _lib.check(
            ws.lib.hf_pcg_graph_create(ctypes.byref(self.handle), ws.handle,
                                       _lib.c_void_p(raw_graph) if raw_graph else None,
                                       _lib.c_void_p(Bp.data_ptr()), float(damping), 1 if timing else 0),
            "hf_pcg_graph_create")
        self.fresh = True

    def refresh(self, Bp, damping):
        if self.fresh:  # created with this solve's arguments
            self.fresh = False
            return
        _lib.check(self.ws.lib.hf_pcg_graph_update(self.handle, _lib.c_void_p(Bp.data_ptr()),
                                                   float(damping)), "hf_pcg_graph_update")

    def launch(self, stream, timed=False):
        _lib.check(self.ws.lib.hf_pcg_graph_launch(self.handle, 1 if timed else 0, stream),
                   "hf_pcg_graph_launch")

    def collect(self):
        _lib.check(self.ws.lib.hf_pcg_graph_collect_timing(self.handle), "hf_pcg_graph_collect_timing")

    def __del__(self):
        try:
            self.ws.lib.hf_pcg_graph_destroy(self.handle)
        except Exception:
            pass


_TIMING_SAMPLE = 16  # with kernel timing on, every 16th fused iteration runs the event-carrying graph


def _iteration_graph(matvec, ws, Bp, damping, mode, with_product):
    """The operator's cached :class:`_IterationGraph` for this workspace, or ``None``
    when the operator is not a captured graph / fusion is switched off
    (``HF_FUSE_ITERATION=0``)."""
    if os.environ.get("HF_FUSE_ITERATION", "1") == "0":
        return None
    raw = getattr(matvec, "raw_graph", None)
    if raw is None:
        return None
    cache = matvec.__dict__.setdefault("_iteration_graphs", {})
    key = (id(ws), mode, bool(with_product), bool(ws.timing))
    g = cache.get(key)
    if g is None:
        g = cache[key] = _IterationGraph(ws, raw() if with_product else None, Bp, damping, mode, ws.timing)
    g.refresh(Bp, damping)
    return g


class _Workspace:
    """Per (device, N, dtype): the native handle and the two vectors that never
    leave the solver (r, p)."""

    _cache = {}

    def __init__(self, device, n, dtype):
        lib = _lib.load()
        self.lib = lib
        self.device, self.n, self.dtype = device, n, dtype
        self.handle = _lib.c_void_p()
        with torch.cuda.device(device):
            _lib.check(
                lib.hf_pcg_create(ctypes.byref(self.handle), n, _lib.dtype_code(dtype),
                                  int(os.environ.get("HF_PCG_BLOCKS", "0"))),
                "hf_pcg_create",
            )
        self.r = torch.empty(n, dtype=dtype, device=device)
        self.p = torch.empty(n, dtype=dtype, device=device)
        self.timing = False
        self._events = [torch.cuda.Event() for _ in range(_LAG + 2)]
        self.checked_groups = weakref.WeakSet()  # (by group OBJECT: an id() is recycled once a group is gone)

    def event(self, it):
        """Recycled completion events: at most _LAG + 1 are pending at any time."""
        return self._events[it % len(self._events)]

    @classmethod
    def get(cls, device, n, dtype):
        key = (device.index if device.index is not None else torch.cuda.current_device(), n, dtype)
        ws = cls._cache.get(key)
        if ws is None:
            ws = cls._cache[key] = cls(device, n, dtype)
        return ws

    def __del__(self):
        try:
            self.lib.hf_pcg_destroy(self.handle)
        except Exception:
            pass


def _check_ranks_agree(ws, group):
    """Data-parallel solves stay in lockstep only if every rank computes bitwise
    identical scalars, i.e. runs the PCG kernels with the same grid (= the same
    partial-sum order).  Checked once per (workspace, group): ranks on GPUs with
    different CU counts or different ``HF_PCG_BLOCKS`` are refused up front instead of
    dead-locking in a later all-reduce."""
    if group in ws.checked_groups:
        return
    dist = torch.distributed
    mine = int(os.environ.get("HF_PCG_BLOCKS", "0")) or 2 * torch.cuda.get_device_properties(
        ws.device).multi_processor_count
    t = torch.tensor([mine, -mine], dtype=torch.int64, device=ws.device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    hi, lo = int(t[0]), -int(t[1])
    if hi != lo:
        raise RuntimeError(
            f"data-parallel PCG needs the same kernel grid on every rank (got {lo}..{hi} blocks): "
            "set HF_PCG_BLOCKS to one value on all ranks")
    ws.checked_groups.add(group)


def _as_operand(t, like, name):
    """Contiguous, right dtype/device, 16-byte aligned view or copy of ``t``."""
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"`{name}` should be a torch.Tensor, not {type(t)}.")
    t = t.detach()
    if t.device != like.device or t.dtype != like.dtype or t.shape != like.shape:
        raise RuntimeError(f"`{name}` must match `b` in device, dtype and shape")
    if not t.is_contiguous() or t.data_ptr() % 16:
        t = t.clone(memory_format=torch.contiguous_format)
    return t


def enable_kernel_timing(device, n, dtype, enable=True):
    """Record HIP events around K1/K2/K3 of every ``hf_pcg_iterate`` of the
    workspace for (device, n, dtype) (used by bench.py's roofline leg)."""
    ws = _Workspace.get(torch.device(device), n, dtype)
    _lib.check(ws.lib.hf_pcg_timing_enable(ws.handle, 1 if enable else 0), "timing_enable")
    ws.timing = bool(enable)
    return ws


def read_kernel_timing(ws):
    k1, k2, k3 = _lib.c_double(), _lib.c_double(), _lib.c_double()
    cnt = _lib.c_int64()
    _lib.check(
        ws.lib.hf_pcg_timing_read(
            ws.handle, ctypes.byref(k1), ctypes.byref(k2), ctypes.byref(k3), ctypes.byref(cnt)
        ),
        "timing_read",
    )
    return {"k1_ms": k1.value, "k2_ms": k2.value, "k3_ms": k3.value, "n": cnt.value}


def cg(
    A,
    b,
    x0=None,
    M=None,
    max_iter=None,
    tol=1e-5,
    atol=None,
    martens_conv_crit=False,
    store_x_at_iters=[],
    verbose=False,
):
    """Preconditioned CG for ``A x = b`` on the GPU; arguments and return value as
    the reference's ``hessianfree.cg.cg`` (cg.py:9-65).

    Returns ``(x_iters, m_iters, reason)``: ``x_iters`` has ``n_iters + 1``
    entries (``None`` where the iterate was not requested, the final iterate
    always present), ``m_iters`` the quadratic's values (0-dim tensors) if
    ``martens_conv_crit`` else ``None``, ``reason`` one of the four strings of
    cg.py:103-115.
    """
    _lib.require_device_tensor(b, "b")
    lib = _lib.load()
    if verbose:
        print("\nStarting cg...")

    b = _as_operand(b, b, "b")
    if b.dim() != 1:
        raise RuntimeError("`b` must be a 1-D vector")
    n, dtype, device = b.numel(), b.dtype, b.device
    max_iter = n if max_iter is None else int(max_iter)
    if max_iter < 1:
        raise ValueError(f"Invalid max_iter: {max_iter}")

    # snapshot table (cg.py:181-183); slots only for iterations that can occur
    if store_x_at_iters is None:
        store_x_at_iters = storing_grid(max_iter)
    store = sorted({int(i) for i in store_x_at_iters if 0 <= int(i) <= max_iter})

    ws = _Workspace.get(device, n, dtype)
    stream = _lib.current_stream_ptr(device)
    width = 16 // b.element_size()
    stride = (n + width - 1) // width * width

    # fresh per-solve storage handed to the caller afterwards
    x = torch.zeros(n, dtype=dtype, device=device) if x0 is None else _as_operand(x0, b, "x0").clone()
    slab = torch.empty((len(store), stride), dtype=dtype, device=device) if store else None
    store_dev = torch.tensor(store, dtype=torch.int64, device=device) if store else None
    m_hist = torch.empty(max_iter + 1, dtype=dtype, device=device) if martens_conv_crit else None

    if M is None:
        mode, minv = _lib.HF_M_NONE, None
    elif isinstance(M, DiagonalPreconditioner) and M.minv.is_cuda:
        mode, minv = _lib.HF_M_DIAG, _as_operand(M.minv, b, "M.minv")
    else:
        mode, minv = _lib.HF_M_EXTERNAL, None

    if isinstance(A, DampedCurvature):
        matvec, damping = A.mvp, A.damping  # (attributes `group` / `collective` of the mvp are honoured)
    else:
        matvec, damping = A, 0.0
    # a hipGraph-captured operator reads its input from a fixed buffer: make that
    # buffer the search direction p itself (K3 then writes straight into it)
    p_vec = ws.p
    ib = getattr(matvec, "input_buffer", None)
    if (
        isinstance(ib, torch.Tensor) and ib.shape == b.shape and ib.dtype == dtype
        and ib.device == device and ib.is_contiguous() and ib.data_ptr() % 16 == 0
    ):
        p_vec = ib

    def ptr(t):
        return _lib.c_void_p(t.data_ptr()) if t is not None else None

    _lib.check(
        lib.hf_pcg_begin(
            ws.handle, ptr(x), ptr(ws.r), ptr(p_vec), ptr(b), ptr(minv), mode, max_iter,
            float(tol), -1.0 if atol is None else float(atol), 1 if martens_conv_crit else 0,
            ptr(store_dev), len(store), 1 if (store and store[0] == 0) else 0, ptr(slab),
            stride, ptr(m_hist),
        ),
        "hf_pcg_begin",
    )

    # ---- initialisation (cg.py:186-192): r = A(x0) - b, p = -M r ---------------
    # (the reference evaluates A(x0) even for x0 = 0; so do we -- the damped
    # operator is evaluated un-fused here, exactly as written in optimizer.py:266)
    Ax0 = _as_operand(A(x), b, "A(x0)")
    _lib.check(lib.hf_pcg_init(ws.handle, ptr(Ax0), stream), "hf_pcg_init")
    if mode == _lib.HF_M_EXTERNAL:
        y = _as_operand(M(ws.r), b, "M(r)")
        _lib.check(lib.hf_pcg_init_external(ws.handle, ptr(y), stream), "hf_pcg_init_external")

    if verbose:
        print(f"Starting iterations (max_iter = {max_iter})...")

    # ---- iterations: enqueue, never read a scalar ------------------------------
    # If the operator ends in a collective (data-parallel matvec), every rank must
    # leave the loop at the SAME iteration or the next all-reduce deadlocks.  The
    # opportunistic poll ("break as soon as the flag is seen") is then replaced by
    # a deterministic rule: at host iteration i wait for the event of iteration
    # i-lag and stop iff the device terminated at an iteration <= i-lag.  Every rank
    # thus performs exactly n_iters+lag operator calls, with no pipeline bubble.
    group = getattr(matvec, "group", None)
    lockstep = group is not None or bool(getattr(matvec, "collective", False)) or bool(
        getattr(A, "lockstep", False))
    if group is not None:
        _check_ranks_agree(ws, group)

    # A hipGraph-captured operator whose buffers the solver adopted runs the whole
    # iteration -- product, K1, K2, K3 -- as ONE graph launch; with a process group the
    # all-reduce separates the product graph from a K1-K3 graph.
    fused = None
    if mode != _lib.HF_M_EXTERNAL and p_vec is ib and hasattr(matvec, "replay_local"):
        fused = _iteration_graph(matvec, ws, matvec.output_buffer, damping, mode,
                                 with_product=group is None)

    status = _lib.Status()
    events = []
    timed_pending = False
    lag = min(_LAG, max(1, _LAG_FUSED)) if fused is not None else _LAG
    for it in range(1, max_iter + 1):
        if verbose:
            print(f"  cg-iteration {it}")
        if fused is not None:
            # (every 16th iteration -- and the 3rd, so that short Martens-terminated solves are sampled too)
            timed = fused.timing and (it % _TIMING_SAMPLE == 0 or it == 3)
            if timed and timed_pending:
                fused.collect()  # iteration it-16: long finished, does not stall the pipeline
            if group is None:
                matvec.calls += 1
            else:
                fused_reduce = getattr(matvec, "replay_and_reduce", None)
                if fused_reduce is not None:  # (the operator overlaps its collective with its own sweep)
                    fused_reduce()
                else:
                    matvec.replay_local()
                    matvec.reduce(matvec.output_buffer)
                matvec.calls += 1
            fused.launch(stream, timed)
            timed_pending = timed_pending or timed
        else:
            Bp = _as_operand(matvec(p_vec), b, "A(p)")
            if mode == _lib.HF_M_EXTERNAL:
                _lib.check(lib.hf_pcg_curvature(ws.handle, ptr(Bp), damping, stream), "curvature")
                _lib.check(lib.hf_pcg_update_xr(ws.handle, ptr(Bp), damping, stream), "update_xr")
                y = _as_operand(M(ws.r), b, "M(r)")
                _lib.check(lib.hf_pcg_update_p(ws.handle, ptr(y), stream), "update_p")
            else:
                _lib.check(lib.hf_pcg_iterate(ws.handle, ptr(Bp), damping, stream), "hf_pcg_iterate")
        ev = ws.event(it)
        ev.record()
        events.append(ev)
        if lockstep:
            if it > lag:
                events.pop(0).synchronize()  # iteration it-lag has completed
                lib.hf_pcg_poll(ws.handle, ctypes.byref(status))
                if status.done and 0 < status.n_iters <= it - lag:
                    break
            continue
        # lagged, non-blocking look at the device's termination flag
        lib.hf_pcg_poll(ws.handle, ctypes.byref(status))
        if status.done:
            break
        if len(events) > lag:
            events.pop(0).synchronize()
            # ... and again once iteration it-lag is known to have finished: without it the NEXT iteration is
            # enqueued before its termination is seen -- a whole curvature product past the end of every solve
            # (measured on the ResNet-18 session: 14 launches for a 12-iteration solve, 9.8 ms; now 13)
            lib.hf_pcg_poll(ws.handle, ctypes.byref(status))
            if status.done:
                break
    if timed_pending:
        fused.collect()

    _lib.check(lib.hf_pcg_finish(ws.handle, ctypes.byref(status), stream), "hf_pcg_finish")
    if not status.done:
        raise RuntimeError("PCG kernels did not terminate within max_iter launches")
    n_iters = int(status.n_iters)
    reason = _lib.REASONS[int(status.reason)]

    # non-positive curvature warnings (cg.py:133-139), issued after the fact
    if status.nonpos_count:
        cap = 32
        its, vals = (_lib.c_int64 * cap)(), (_lib.c_double * cap)()
        k = lib.hf_pcg_read_nonpos(ws.handle, its, vals, cap)
        for j in range(max(k, 0)):
            msg = f"Directional curvature pAp = {vals[j]:.3e} <= 0 detected in cg-"
            msg += f"iteration {its[j]}. This is a violation to the assumption "
            msg += "of positive definiteness."
            warn(msg)
    if verbose:
        print(f"Residual norm required for termination: {status.res_bound:.6e}")
        print(reason)

    # ---- results in the reference's format ------------------------------------
    x_iters = [None] * (n_iters + 1)
    for slot, it in enumerate(store):
        if it <= n_iters:
            x_iters[it] = slab[slot, :n]
    if x_iters[-1] is None:
        x_iters[-1] = x  # the final iterate is always returned (cg.py:229-230)
    m_iters = list(m_hist[: n_iters + 1].unbind(0)) if martens_conv_crit else None
    return x_iters, m_iters, reason
