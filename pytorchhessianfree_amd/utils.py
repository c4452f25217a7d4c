"""Flat-vector <-> parameter conversions (reference ``hessianfree/utils.py:8-76``)
plus the "flatten once into contiguous HBM" arena the solver works on."""

from warnings import warn

import torch

from . import _lib


def vector_to_trainparams(vec, parameters):
    """Make every TRAINABLE parameter a view of its slice of ``vec`` (frozen
    parameters are skipped and do not consume entries) -- utils.py:8-38."""
    if not isinstance(vec, torch.Tensor):
        raise TypeError(f"`vec` should be a torch.Tensor, not {type(vec)}.")
    offset = 0
    for param in parameters:
        if not param.requires_grad:
            continue
        count = param.numel()
        param.data = vec[offset : offset + count].view_as(param).data
        offset += count
    if offset != len(vec):
        warn("Not all entries of `vec` have been used.")


def vector_to_parameter_list(vec, parameters):
    """List of views of ``vec`` shaped like ``parameters`` (which stay untouched)
    -- utils.py:41-76."""
    if not isinstance(vec, torch.Tensor):
        raise TypeError(f"`vec` should be a torch.Tensor, not {type(vec)}.")
    views, offset = [], 0
    for param in parameters:
        count = param.numel()
        views.append(vec[offset : offset + count].view_as(param).data)
        offset += count
    if offset != len(vec):
        warn("Not all entries of `vec` have been used.")
    return views


class ParameterArena:
    """One contiguous vector ``theta`` holding all trainable parameters, with the
    parameters re-bound as views into it.

    The reference reaches the same state after its first step (its
    ``vector_to_trainparams`` leaves parameters as views of ``params_vec + step``,
    optimizer.py:293, :349-350) but re-creates the vector with ``torch.cat`` and a
    fresh allocation for every trial step.  Here the arena is persistent:
    ``write(base, step, alpha)`` overwrites it in place with the fused
    ``hf_axpy_out`` kernel (12 N bytes, no allocation, no re-binding)."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        ref = self.params[0]
        self.n = sum(p.numel() for p in self.params)
        self.theta = torch.empty(self.n, dtype=ref.dtype, device=ref.device)
        self._bind(copy_in=True)

    def _bind(self, copy_in):
        offset = 0
        with torch.no_grad():
            for p in self.params:
                count = p.numel()
                view = self.theta[offset : offset + count].view_as(p)
                if copy_in:
                    view.copy_(p.data)
                p.data = view
                offset += count

    def ensure_bound(self):
        """Re-adopt parameters whose storage was replaced from outside (e.g. by
        ``model.to(...)`` or a hand-written ``param.data = ...``)."""
        offset, ok = 0, True
        esize = self.theta.element_size()
        for p in self.params:
            if p.data_ptr() != self.theta.data_ptr() + offset * esize or p.device != self.theta.device:
                ok = False
                break
            offset += p.numel()
        if not ok:
            ref = self.params[0]
            if ref.device != self.theta.device or ref.dtype != self.theta.dtype:
                self.theta = torch.empty(self.n, dtype=ref.dtype, device=ref.device)
            self._bind(copy_in=True)

    def snapshot(self):
        return self.theta.clone()

    def write(self, base, step, alpha=1.0):
        """theta <- base + alpha*step  (elementwise ``base + (alpha*step)``, two
        roundings, as ``params_vec + lr * step_vec`` in optimizer.py:349)."""
        if self.theta.is_cuda:
            _lib.axpy_out(self.theta, base, step, alpha)
            # the kernel wrote behind autograd's back: bump the parameters' version
            # counters (``p.data = view`` keeps each parameter's own counter) so that
            # a stale graph that saved the old weights fails loudly in backward
            # instead of silently using the new ones
            for p in self.params:
                torch.autograd.graph.increment_version(p)
        else:
            with torch.no_grad():
                torch.add(base, alpha * step if alpha != 1.0 else step, out=self.theta)
        return self.theta
