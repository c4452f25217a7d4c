"""The network's forward pass on the engine's static buffers, own kernels only (reference: the ``forward()`` of
optimizer.py:216-229 and every ``tfunc`` call, :288-294), eval- and train-mode BatchNorm; the recorded-activation load.

Mixin of ``FusedGGNEngine`` (engine/core.py); see that class for the sweeps' overall structure."""

import torch
from torch import nn

from .. import _lib
from .common import _pair, _ptr


class _Forward:
    def _bn_forward(self, u, splits, update_running=True):
        n, k, oh, ow = u.a.shape
        res = u.res
        if u.train:
            # ONE pass for the partial sums (the convolution's slabs summed into ``a``, per-channel sum a / sum a^2 in
            # fp64 per row block); the normalising launch adds them up in its prologue -- 2 launches per unit
            bn, st = u.bn, _lib.current_stream_ptr(self.dev)
            move = update_running and bn.track_running_stats
            count = float(n * oh * ow)
            _lib.check(_lib.load().hf_bn_stats_rows(
                _ptr(u.a), _ptr(u.tbuf), splits, u.tbuf.shape[1], _ptr(u.stat_part), n * oh * ow, k, u.rb,
                _lib.HF_F32, st), "hf_bn_stats_rows")
            _lib.check(_lib.load().hf_bn_forward_train(
                _ptr(u.y), _ptr(u.yout2), 2 * k if u.yout2 is not None else 0, _ptr(u.a), _ptr(u.stat_part), u.rb,
                _ptr(u.mean_t), _ptr(u.rstd), _ptr(bn.running_mean) if move else None,
                _ptr(bn.running_var) if move else None, count, float(bn.eps), float(bn.momentum) if move else -1.0,
                _ptr(u.scale), _ptr(u.shift), _ptr(res), 0, 1 if u.relu else 0, n * oh * ow, k, _lib.HF_F32, st),
                "hf_bn_forward_train")
            if move:
                bn.num_batches_tracked.add_(1)
            return
        _lib.check(_lib.load().hf_bn_forward(
            _ptr(u.y), _ptr(u.yout2), 2 * k if u.yout2 is not None else 0, _ptr(u.a), _ptr(u.tbuf), splits,
            u.tbuf.shape[1], _ptr(u.mean), _ptr(u.rstd), _ptr(u.scale), _ptr(u.shift),
            _ptr(res), 0, 1 if u.relu else 0, n * oh * ow, k, _lib.HF_F32, _lib.current_stream_ptr(self.dev)),
            "hf_bn_forward")

    def _conv_forward(self, u):
        if u.im2col:  # 1x1 product over the im2col; the weight is read from the parameter itself
            self._conv_slabs(0, u.tbuf, u.cols, u.conv.weight.detach(), u.geo, u.sF)
            return
        n, h, w, c, k, r, s, st, pd = u.geo
        self._conv_slabs(0, u.tbuf, u.x, u.wcat, u.geo, u.sF, mat_ld=2 * c)

    def forward_own(self, refresh=False, update_running=True):
        """The network's forward pass on the engine's static buffers, own kernels only: every
        activation lands where the sweeps read it (dense output = ReLU mask / weight-gradient operand
        / residual, and the x half of the consumer's [t_x | x] operand), max-pool positions, logits
        and -- for a softmax cross-entropy -- probabilities and the loss value.  ``refresh``: first bring
        the W halves of all [W | v_W] operands up to the current parameters (``refresh_weights()``) -- carried
        by the stem's convolution launch where possible (the stem reads its weight from the parameter)."""
        s = self.stem
        flat = self._flat_params
        carried = (refresh and flat is not None and flat.data_ptr() == self.params[0].data_ptr()
                   and self._conv_carrying_scatter(s, s.conv.weight.detach(), s.sF, flat, 0))
        if not carried:
            if refresh:
                self.refresh_weights()
            self._conv_forward(s)
        self._bn_forward(s, s.sF, update_running)
        ks, st_, pd, _dl, _cm = self.pool_args
        pn, ph, pw, poh, pow_, c0 = self._pool_geometry()
        (kh, kw), (sh, sw), (pph, ppw) = _pair(ks), _pair(st_ if st_ is not None else ks), _pair(pd)
        _lib.check(_lib.load().hf_maxpool_forward_nhwc(
            _ptr(self.pool_out), _ptr(self.pool_t[:, c0:]), 2 * c0, _ptr(self.pool_idx32), _ptr(s.y), pn, ph, pw,
            poh, pow_, c0, kh, kw, sh, sw, pph, ppw, _lib.HF_F32, _lib.current_stream_ptr(self.dev)),
            "hf_maxpool_forward_nhwc")
        for chain, ds, _x in self.blocks:
            if ds is not None:
                self._conv_forward(ds)
                self._bn_forward(ds, ds.sF, update_running)
            for u in chain:
                self._conv_forward(u)
                self._bn_forward(u, u.sF, update_running)
        tail = self.tail
        if self._head_hw > 1:
            torch.mean(tail.y, dim=(2, 3), out=self.feat)
        fw = self.fc.weight.detach()
        if self.pfb is not None:
            torch.addmm(self.fc.bias.detach(), self.feat, fw.t(), out=self.logits)
        else:
            torch.mm(self.feat, fw.t(), out=self.logits)
        if getattr(self, "loss_spec", None) is not None:
            self._loss_head()
        self._at = "own"
        return self.logits

    def _loss_head(self):
        """Softmax probabilities and the cross-entropy value of the current logits (the reference's
        ``forward()[0]``, optimizer.py:216-229; same ATen ops as ``F.cross_entropy``)."""
        if self.loss_spec["kind"] == "mse":  # (same ATen op as ``F.mse_loss`` / ``nn.MSELoss``)
            val = torch.nn.functional.mse_loss(self.logits, self._targets, reduction=self.loss_spec["reduction"])
        else:
            torch.softmax(self.logits, 1, out=self._p)
            lsm = torch.log_softmax(self.logits, 1)
            val = torch.nn.functional.nll_loss(lsm, self._targets, reduction=self.loss_spec["reduction"])
        if getattr(self, "_l2", None) is not None:
            theta = self._theta()
            val = val + 0.5 * torch.dot(self._l2 * theta, theta)
        self.loss_buf.copy_(val)

    def _load_recorded(self, outputs):
        """Overwrite the engine's activations with the ones the MODEL's forward pass recorded, so that a
        product of the engine and one of the autograd operator linearise at bitwise the same point -- same
        ReLU masks, same pooling positions.  Two correct fp32 forward passes may decide a ReLU whose input
        is within rounding of zero differently, and one such sign moves a product of a deep net by ~1e-4 of
        its max-norm: not an error of either, but it would drown the comparison below."""
        for u in self.units:
            u.a.copy_(u.ra)
            u.y.copy_(u.ry)
            if u.yout2 is not None:
                u.yout2.copy_(u.ry)
            if u.train:  # (the own forward pass writes its batch statistics into the same buffers)
                u.mean_t.copy_(u.rec_stats[0])
                u.rstd.copy_(u.rec_stats[1])
        if self.pool_args is not None:
            ks, st_, pd, dl, cm = self.pool_args
            _, idx = torch.nn.functional.max_pool2d(self.stem.ry, ks, st_, pd, dl, cm, return_indices=True)
            self.pool_idx32.copy_(idx.permute(0, 2, 3, 1))
            self.pool_out.copy_(self._rec_pool)
            c0 = self.pool_out.shape[1]
            self.pool_t[:, c0:].copy_(self._rec_pool)
        if self.fc is not None and self._head_hw > 1:
            torch.mean(self.tail.y, dim=(2, 3), out=self.feat)
        self.logits.copy_(outputs.detach())
        if getattr(self, "loss_spec", None) is not None:
            self._loss_head()
        self._at = "recorded"
