"""Diagonal of the empirical Fisher from one adjoint sweep + per-sample weight-gradient launches
(preconditioners.py:11-105).

Mixin of ``FusedGGNEngine`` (engine/core.py); see that class for the sweeps' overall structure."""

import torch

from .. import _lib
from .common import _ptr


class _DiagEF:
    # ---- diagonal of the empirical Fisher (preconditioners.py:11-105) ---------------------------------------------
    def diag_ef(self, reduction="mean", out=None):
        """``sum_i g_i^2`` (``sum``) / ``(1/N) sum_i g_i^2`` (``mean``) over the per-sample gradients ``g_i`` of the
        engine's CURRENT batch (``set_batch`` + ``forward_own`` must have run): the quantity of the reference's
        ``diag_EF_autograd`` / ``diag_EF_backpack`` (one backward pass per sample / BackPACK's ``SumGradSquared``),
        from ONE adjoint sweep of the whole batch -- in eval mode the samples do not interact, so the batch
        cotangents ARE the per-sample cotangents -- followed, per sample, by the weight-gradient convolutions on that
        sample's rows and one squaring gather (``hf_pack_ex`` mode 1).  A tagged L2 term (each per-sample loss of
        the reference carries it whole) enters in closed form: sum (a_i + b)^2 = sum a_i^2 + 2 b sum a_i + N b^2."""
        if self.loss_spec is None or self.train_bn:
            raise RuntimeError("engine.diag_ef needs a softmax cross-entropy / MSE loss and eval-mode BatchNorm")
        if self.loss_spec["reduction"] != reduction:
            raise RuntimeError("engine.diag_ef: the loss's reduction differs from the requested one")
        n = self.x_in.shape[0]
        if out is None:
            out = torch.empty(self.n, dtype=torch.float32, device=self.dev)
        out.zero_()
        l2, self._l2 = self._l2, None
        try:
            mean_grad = self.gradient()  # fills the units' cotangents (g / ga, or g1 / ga1 of a Hessian engine)
        finally:
            self._l2 = l2
        scale = float(n) if reduction == "mean" else 1.0   # the sweep's cotangents carry the loss's 1/N
        first_order = self.hessian
        lib, st = _lib.load(), _lib.current_stream_ptr(self.dev)
        self._diag_buffers()
        tensors, perms, splits = self._diag_pack
        for i in range(n):
            for u in self.units:
                if u.dead:  # (frozen, behind frozen layers only: the sweep left no cotangent here, nothing is gathered)
                    continue
                ga = (u.ga1 if first_order else u.ga)[i:i + 1]
                if u.pw is None:
                    pass  # (frozen weight: no entry)
                elif u.im2col:
                    self._conv_slabs(2, u.wps, u.cols_pad[i:i + 1], ga, u.geo_w1, u.sW1, out_c=u.jcols)
                else:
                    self._conv_slabs(2, u.wps, u.x[i:i + 1], ga, u.geo1, u.sW1)
                k, hw = u.a.shape[1], u.a.shape[2] * u.a.shape[3]
                if u.pg is None and u.pb is None:
                    continue
                if u.bn is not None:
                    g = (u.g1 if first_order else u.g)[i:i + 1]
                    _lib.check(lib.hf_chan_affine_bwd_ex(
                        None, _ptr(u.gws), _ptr(u.gbs), None, _ptr(g), 1, 0, None, 1, 0, _ptr(u.a[i:i + 1]), _ptr(u.mean),
                        _ptr(u.rstd), None, None, 1, k, hw, 1, 1, _lib.HF_F32, st), "hf_chan_affine_bwd_ex")
                elif u.pb is not None:
                    _lib.check(lib.hf_chan_affine_bwd_ex(
                        None, None, _ptr(u.gbs), None, _ptr(ga), 1, 0, None, 1, 0, None, None, None, None, None, 1, k, hw,
                        1, 1, _lib.HF_F32, st), "hf_chan_affine_bwd_ex")
            _lib.pack_ex(out, tensors, perms, splits, scale=scale, live=self._pack_live, mode=1)
        if self.fc is not None:
            # linear head: the per-sample weight gradient is the outer product g_i x feat_i, so the sum of its squares
            # is (g o g)^T (feat o feat)
            g = self._dlogits() * scale
            nf = self.fc.weight.numel()
            out[self._offs[self.pfw]: self._offs[self.pfw] + nf].copy_(((g * g).t() @ (self.feat * self.feat)).reshape(-1))
            if self.pfb is not None:
                out[self._offs[self.pfb]: self._offs[self.pfb] + g.shape[1]].copy_((g * g).sum(0))
        if l2 is not None:
            sum_a = mean_grad * (scale / self.weight)  # sum_i a_i
            b = l2 * self._theta()
            out.add_(2.0 * b * sum_a).add_(float(n) * b * b)
        if reduction == "mean":
            out.div_(float(n))
        return out

    def _diag_buffers(self):
        """Per-sample (n = 1) geometry, split counts and slab buffers of ``diag_ef`` (allocated on first use)."""
        if getattr(self, "_diag_pack", None) is not None:
            return
        f32, dev = torch.float32, self.dev
        tensors, perms, splits = [None] * len(self.params), {}, {}
        for u in self.units:
            k = u.a.shape[1]
            if u.bn is not None and not self.hessian and u.g is None:
                u.g, u.needs_g = torch.empty_like(u.a), True  # (every unit's masked cotangent is needed per sample)
            if u.im2col:
                rows1 = u.a.shape[2] * u.a.shape[3]
                u.geo_w1 = (rows1,) + tuple(u.geo_w[1:])
                u.sW1 = self._plan_stem(2, u.geo_w1)
            else:
                u.geo1 = (1,) + tuple(u.geo[1:])
                n_, h, w, c, k_, r, s_, sd, pd = u.geo1
                u.sW1 = _lib.conv_plan(2, 1, h, w, c, k_, r, s_, sd, pd)
            u.wps = torch.zeros((u.sW1, u.conv.weight.numel()), dtype=f32, device=dev)  # dead taps stay 0
            if u.pw is not None:
                tensors[u.pw] = u.wps[0]
                if not u.im2col:
                    k_, c, r, s_ = u.conv.weight.shape
                    if r * s_ > 1:
                        perms[u.pw] = (c, r * s_)
                if u.sW1 > 1:
                    splits[u.pw] = (u.sW1, u.wps.shape[1])
            u.gws = torch.zeros((1, k), dtype=f32, device=dev)
            u.gbs = torch.zeros((1, k), dtype=f32, device=dev)
            if u.pg is not None:
                tensors[u.pg] = u.gws[0]
            if u.pb is not None:
                tensors[u.pb] = u.gbs[0]
        if self.fc is not None:  # (filled in closed form afterwards: zeros here)
            self._diag_zero_fw = torch.zeros(self.fc.weight.numel(), dtype=f32, device=dev)
            tensors[self.pfw] = self._diag_zero_fw
            if self.pfb is not None:
                self._diag_zero_fb = torch.zeros(self.fc.weight.shape[0], dtype=f32, device=dev)
                tensors[self.pfb] = self._diag_zero_fb
        self._pack_args()  # (makes sure _pack_live exists)
        self._diag_pack = (tensors, perms, splits)
