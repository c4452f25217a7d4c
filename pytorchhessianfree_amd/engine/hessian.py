"""The terms of a Hessian product that do not depend on the adjoint chain (optimizer.py:450-455 by
forward-over-reverse), issued on a parallel graph branch.

Mixin of ``FusedGGNEngine`` (engine/core.py); see that class for the sweeps' overall structure."""

import os

import torch

from .. import _lib
from .common import _ptr


class _HessianExtras:
    def _hessian_extras(self, u):
        """The terms of a Hessian product that do NOT depend on the adjoint chain -- only on the tangent sweep's
        results and the step's first-order cotangents: the scale's  sum_rows g_z * rstd * t_a  (t_a = the sum of the
        tangent convolution's slabs, still in place) as `rb` more partial rows of the gw buffer, and conv_D(g_a, V) /
        conv_W(t_x, g_a) as MORE SLABS of the same buffers (the consumers sum them anyway)."""
        if u.dead:  # (frozen, behind frozen layers only: no tangent reaches it, no cotangent is needed behind it)
            return
        if u.bn is not None and not u.train:  # (train mode: the tangent sweep left this sum, engine/tangent.py)
            n, k, oh, ow = u.a.shape
            _lib.check(_lib.load().hf_chan_affine_bwd_ex(
                None, _ptr(u.gw[u.rb:]), None, None, _ptr(u.tbuf), u.sT, u.tbuf.shape[1], None, 1, 0, _ptr(u.g1),
                _ptr(self._zeros(k)), _ptr(u.rstd), None, None, n, k, oh * ow, 1, u.rb, _lib.HF_F32,
                _lib.current_stream_ptr(self.dev)), "hf_chan_affine_bwd_ex")
        if not u.im2col and not u.first and u.sD:  # (u.sD == 0: a tangent-free input -- both terms vanish)
            c = u.x.shape[1]
            if self._extras_mode == 2:  # (conv_D(g_a, V) rides in the chain's launch: only the weight term here)
                self._conv_slabs(2, u.wbuf[u.sW:], u.xcat, u.ga1, u.geo, u.sW, act_ld=2 * c)
            else:
                _lib.conv_dw_slabs((1, u.dbuf[u.sD:], u.ga1, u.vT, u.geo, u.sD, 0, 0),
                                   (2, u.wbuf[u.sW:], u.xcat, u.ga1, u.geo, u.sW, 2 * c, 0), self.dev)

    # Those extras are half of a Hessian product's launches and none of them is on the adjoint sweep's dependency
    # chain: they are issued on a SECOND STREAM forked off after the tangent sweep (inside a hipGraph capture: a
    # parallel branch of the graph), in the adjoint's unit order; the chain waits per unit for the data-gradient
    # slabs it is about to sum (an event per unit) and once, before the gather, for the rest.
    # Forms: 1 = all extras on the side branch, the chain waits per unit for the data-gradient slabs it needs (one
    # cross-branch dependency per unit); 2 = conv_D(g, V) inside the chain's own grouped launch, ONLY results nobody on
    # the chain reads on the side branch: one fork, one join.  One form per engine family (``_extras_default``), by
    # measurement (profiles/r04_hessian_parallel_branch.jsonl: ResNet-18 sequential 852 / form 1 922-930 / form 2
    # 947-959 matvecs/s; All-CNN-C 493 / 521 / 283 -- its 128-wide tiles spill a three-problem argument block); in
    # sequence only where the caller already runs this engine on one of several parallel branches.
    _extras_parallel = False
    _extras_mode = 0
    _extras_default = 2

    def _extras_fork(self):
        # (``_extras_allowed = False``: the caller already runs this engine on one of several parallel branches --
        # session.AccumulatedSession -- and a fork inside a forked capture branch crashes hipStreamEndCapture
        # on this stack: segfault in capture_end, round-4 batch r4f)
        # (measured, profiles/r04_hessian_parallel_branch.jsonl: ResNet-18 form 1 922-930, form 2 947-959 matvecs/s;
        # All-CNN-C form 1 521, form 2 283 -- its 128-wide tile configurations spill a three-problem argument block)
        mode = self._extras_default
        self._extras_parallel = getattr(self, "_extras_allowed", True)
        self._extras_mode = mode if self._extras_parallel else 0
        if not self._extras_parallel:
            return
        self._side_setup()
        cur = torch.cuda.current_stream(self.dev)
        self._xfork.record(cur)
        self._xside.wait_event(self._xfork)
        self._xwait = {}
        with torch.cuda.stream(self._xside):
            for u in reversed(self.units):
                self._hessian_extras(u)
                if self._extras_mode == 1 and getattr(u, "sD", 0) and not u.im2col and not u.first:
                    ev = self._xev[id(u)]
                    ev.record(self._xside)
                    self._xwait[u.dbuf.data_ptr()] = ev
            self._xjoin2 = getattr(self, "_xjoin2", None) or torch.cuda.Event()
            self._xjoin2.record(self._xside)

    def _extras_wait(self, srcs):
        """Before a unit sums data-gradient slabs: the side branch's share of them must be there."""
        if self._extras_parallel and self._second:
            cur = torch.cuda.current_stream(self.dev)
            for buf, _n, _l in srcs:
                ev = self._xwait.get(buf.data_ptr())
                if ev is not None:
                    cur.wait_event(ev)

    def _extras_join(self):
        if self._extras_parallel:
            torch.cuda.current_stream(self.dev).wait_event(self._xjoin2)
            self._extras_parallel = False
        self._extras_mode = 0

    # ---- side branch for launches that are off the adjoint sweep's dependency chain -------------------------
    # (inside a hipGraph capture: a parallel branch of the graph; eager: a second stream)
    def _side_setup(self):
        if getattr(self, "_xside", None) is None:
            self._xside = torch.cuda.Stream(device=self.dev)
            self._xfork = torch.cuda.Event()
            self._xev = {id(u): torch.cuda.Event() for u in self.units}

    def _swap_first_order(self):
        for u in self.units:
            u.g, u.g1 = u.g1, u.g
            u.ga, u.ga1 = u.ga1, u.ga
