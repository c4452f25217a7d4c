"""Tangent sweep J v: per unit the tangent convolution of ``[t_x | x]`` with ``[W | v_W]`` and the BatchNorm / bias
tangent (+ residual tangent, ReLU mask); the classifier head with the loss Hessian (optimizer.py:457-462, R-op).

Mixin of ``FusedGGNEngine`` (engine/core.py); see that class for the sweeps' overall structure."""

import os

import torch

from .. import _lib
from .common import _ptr


class _TangentSweep:
    def _bn_tangent(self, u, v, add, add_ld):
        """t_y = mask * (sum(T slabs) * w*rstd + xhat * v_w + v_b + add), into the consumer's operand."""
        n, k, oh, ow = u.a.shape
        vg = v[self._offs[u.pg]: self._offs[u.pg] + k] if u.pg is not None else None
        vb = v[self._offs[u.pb]: self._offs[u.pb] + k] if u.pb is not None else None
        if u.train:
            # reduction (partial rows: by the convolution's epilogue, else by its own launch), then the elementwise
            # pass adds them up in its prologue
            lib, st = _lib.load(), _lib.current_stream_ptr(self.dev)
            px, p1, nparts = u.gw, u.gb, u.rb
            if self.hessian:
                # the adjoint of a Hessian product reads these sums again (hf_bn_train_hessian_coeffs): rows of their
                # own -- and  rstd * sum_rows g_z * t_a  (g_z: the step's first-order masked cotangent) as the second
                # `rb` rows of gw, where the gather expects the scale's second-order share
                px, p1 = u.hx, u.h1
                _lib.check(lib.hf_chan_affine_bwd_ex(
                    None, _ptr(u.gw[u.rb:]), None, None, _ptr(u.tbuf), u.sT, u.tbuf.shape[1], None, 1, 0, _ptr(u.g1),
                    _ptr(self._zeros(k)), _ptr(u.rstd), None, None, n, k, oh * ow, 1, u.rb, _lib.HF_F32, st),
                    "hf_chan_affine_bwd_ex")
            if u.tsum:
                px, p1, nparts = u.tpx, u.tp1, u.tp1.shape[0]
            u.tsums = (px, p1, nparts)  # (a Hessian product's adjoint reads them again)
            if not u.tsum:
                _lib.check(lib.hf_chan_affine_bwd_ex(
                    None, _ptr(px), _ptr(p1), None, _ptr(u.tbuf), u.sT, u.tbuf.shape[1], None, 1, 0, _ptr(u.a),
                    _ptr(u.mean), _ptr(u.rstd), None, None, n, k, oh * ow, 1, u.rb, _lib.HF_F32, st),
                    "hf_chan_affine_bwd_ex")
            _lib.check(lib.hf_chan_affine_train(
                _ptr(u.tout), _ptr(u.tbuf), _ptr(u.a), _ptr(u.mean), _ptr(u.rstd), _ptr(u.scale), _ptr(px),
                _ptr(p1), nparts, _ptr(vg), _ptr(vb), float(n * oh * ow), _ptr(add), _ptr(u.y) if u.relu else None,
                n, k, oh * ow, u.tout_ld, add_ld, u.sT, u.tbuf.shape[1], _lib.HF_F32, st), "hf_chan_affine_train")
            return
        _lib.check(_lib.load().hf_chan_affine_ex(
            _ptr(u.tout), _ptr(u.tbuf), _ptr(u.a), _ptr(u.mean), _ptr(u.rstd), _ptr(u.scale),
            _ptr(vg), _ptr(vb), _ptr(add), _ptr(u.y) if u.relu else None, 0, n, k, oh * ow, 1, u.tout_ld, add_ld,
            u.sT, u.tbuf.shape[1], _lib.HF_F32, _lib.current_stream_ptr(self.dev)), "hf_chan_affine_ex")

    def _train_pair_ok(self, u1, u2):
        return u1.train and u2.train and os.environ.get("HF_BN_TRAIN_PAIR", "1") != "0"

    def _affine_train_problem(self, q, u, out, out_ld, a, a_splits, a_slab, px, p1, nparts, vq, vr, mask):
        n, k, oh, ow = u.a.shape
        q.out, q.a, q.x = out.data_ptr(), a.data_ptr(), u.a.data_ptr()
        q.mean, q.rstd, q.w = u.mean.data_ptr(), u.rstd.data_ptr(), u.scale.data_ptr()
        q.part_x, q.part_1, q.nparts = px.data_ptr(), p1.data_ptr(), nparts
        q.vq = vq.data_ptr() if vq is not None else None
        q.vr = vr.data_ptr() if vr is not None else None
        q.count, q.add, q.mask_src = float(n * oh * ow), None, (mask.data_ptr() if mask is not None else None)
        q.n, q.c, q.hw, q.out_ld, q.a_splits, q.a_slab = n, k, oh * ow, out_ld, a_splits, a_slab

    def _bn_tangent_pair_train(self, u1, u2, v):
        """Train mode, prologue form: the elementwise passes of two units without residual input in ONE launch (their
        partial sums came from the convolutions' epilogue)."""
        arr = (_lib.AffineTrainProblem * 2)()
        for q, u in zip(arr, (u1, u2)):
            k = u.a.shape[1]
            vg = v[self._offs[u.pg]: self._offs[u.pg] + k] if u.pg is not None else None
            vb = v[self._offs[u.pb]: self._offs[u.pb] + k] if u.pb is not None else None
            self._affine_train_problem(q, u, u.tout, u.tout_ld, u.tbuf, u.sT, u.tbuf.shape[1], u.tpx, u.tp1,
                                       u.tp1.shape[0], vg, vb, u.y if u.relu else None)
        _lib.check(_lib.load().hf_chan_affine_train_pair(_lib.ctypes.cast(arr, _lib.c_void_p), _lib.HF_F32,
                                                         _lib.current_stream_ptr(self.dev)), "hf_chan_affine_train_pair")

    def _bn_tangent_pair(self, u1, u2, v):
        """The BatchNorm tangents of two units without residual input in ONE launch."""
        arr = (_lib.AffineProblem * 2)()
        for q, u in zip(arr, (u1, u2)):
            n, k, oh, ow = u.a.shape
            q.out, q.a, q.x = u.tout.data_ptr(), u.tbuf.data_ptr(), u.a.data_ptr()
            q.mean, q.rstd, q.w = u.bn.running_mean.data_ptr(), u.rstd.data_ptr(), u.bn.weight.data_ptr()
            q.q = v.data_ptr() + 4 * self._offs[u.pg] if u.pg is not None else None  # (frozen scale / shift: no tangent)
            q.r = v.data_ptr() + 4 * self._offs[u.pb] if u.pb is not None else None
            q.add, q.mask_src, q.relu_self = None, (u.y.data_ptr() if u.relu else None), 0
            q.n, q.c, q.hw, q.out_ld, q.add_ld = n, k, oh * ow, u.tout_ld, 0
            q.a_splits, q.a_slab = u.sT, u.tbuf.shape[1]
        _lib.check(_lib.load().hf_chan_affine_pair(_lib.ctypes.cast(arr, _lib.c_void_p), _lib.HF_F32,
                                                   _lib.current_stream_ptr(self.dev)), "hf_chan_affine_pair")

    def _tangent_stem(self, v, carry_scatter=False):
        """Stem: conv(x, v_W) as a 1x1 convolution on the im2col'd input (the input has no tangent; v_W is a
        slice of ``v`` itself).  ``carry_scatter``: the launch also carries the scatter of every other layer's v_W
        into its ``[W | v_W]`` operand (``hf_conv2d_nhwc_slabs_unpack``) -- nothing in the stem reads those."""
        s = self.stem
        if s.dead:  # (frozen stem: its output carries no tangent -- only the scatter it would have carried is left)
            if carry_scatter and self._slot_list:
                _lib.unpack_tangent(v, self._slot_list)
            return
        if s.pw is None:  # (frozen stem weight under a trainable BatchNorm: conv(x, 0) = 0)
            vw = self._zeros(s.conv.weight.numel())
        else:
            vw = v[self._offs[s.pw]: self._offs[s.pw] + s.conv.weight.numel()]
        if not (carry_scatter and self._conv_carrying_scatter(s, vw, s.sT, v, 1)):
            if carry_scatter:
                _lib.unpack_tangent(v, self._slot_list)  # v_W halves of all [W | v_W] operands: one launch
            self._conv_slabs(0, s.tbuf, s.cols, vw, s.geo, s.sT)
        self._bn_tangent(s, v, None, 0)
        pn, ph, pw, poh, pow_, c0 = self._pool_geometry()
        _lib.check(_lib.load().hf_maxpool_tangent_nhwc(
            _ptr(self.pool_t), _ptr(s.tout), _ptr(self.pool_idx32), pn, ph, pw, poh, pow_, c0, 2 * c0,
            _lib.HF_F32, _lib.current_stream_ptr(self.dev)), "hf_maxpool_tangent_nhwc")

    def _tangent_convs(self, units):
        """The tangent convolutions ``conv([t_x | x], [W | v_W])`` of one or two units in ONE launch.  In front of a
        train-mode BatchNorm the launch's epilogue also writes the per-channel partial sums of its output tiles
        (``hf_conv2d_nhwc_group_slabs_bnsum``): the reduction launch between convolution and elementwise pass is gone
        (``u.tsum``: ``_bn_tangent`` then adds ``u.tp1 / u.tpx`` up instead of ``u.gb / u.gw``)."""
        probs = [(0, u.tbuf, u.xcat, u.wcat, self._tgeo(u), u.sT, 0, 0) for u in units]
        if any(u.epi for u in units):
            sums = [(u.a, u.mean, u.rstd, u.tpx, u.tp1) if u.epi else None for u in units]
            if _lib.conv_group_slabs_bnsum(probs, sums, self.dev):
                for u in units:
                    u.tsum = u.epi
                return
            for u in units:  # (a geometry the 64x64-tile instantiations do not cover: not tried again)
                u.epi = False
        for u in units:
            u.tsum = False
        if len(units) == 1:
            u = units[0]
            self._conv_slabs(0, u.tbuf, u.xcat, u.wcat, self._tgeo(u), u.sT)
        else:
            _lib.conv_group_slabs(probs, self.dev)

    def _tangent_blocks(self, v):
        group = self._grouping()
        for chain, ds, _x in self.blocks[self.dead_blocks:]:  # (dead blocks: frozen, behind frozen layers -- no tangent)
            head = chain[0]
            paired = False
            if ds is not None and group:
                # the downsample branch and the block's first convolution read the same operand:
                # both tangent convolutions in ONE launch
                self._tangent_convs([ds, head])
                alone = head.res_unit is None and not head.res_identity and len(chain) > 1
                paired = alone and not head.train and not ds.train
                if paired:  # ... and both BatchNorm tangents in one
                    self._bn_tangent_pair(ds, head, v)
                elif alone and self._train_pair_ok(ds, head) and ds.tsum and head.tsum:
                    self._bn_tangent_pair_train(ds, head, v)  # (train mode: the same, prologue form)
                    paired = True
                else:
                    self._bn_tangent(ds, v, None, 0)
            elif ds is not None:
                self._tangent_convs([ds])
                self._bn_tangent(ds, v, None, 0)
            for u in chain:
                if u is head and paired:
                    continue
                if not (u is head and ds is not None and group):
                    self._tangent_convs([u])
                add, add_ld = None, 0
                if u.res_unit is not None:
                    add, add_ld = u.res_unit.tout, u.res_unit.tout_ld
                elif u.res_identity:
                    c = head.x.shape[1]
                    add, add_ld = head.xcat[:, :c], 2 * c
                self._bn_tangent(u, v, add, add_ld)

    # ---- classifier head: logits' tangent, loss Hessian, the head's gradients ----------------------
    def _head(self, v):
        """Returns the cotangent of the last unit's output and the head's weight / bias gradients."""
        tail = self.tail
        t_last = tail.tout
        hw = t_last.shape[2] * t_last.shape[3]
        fw = self.fc.weight
        nf = fw.numel()
        v_fw = v[self._offs[self.pfw]: self._offs[self.pfw] + nf].view_as(fw)
        v_fb = None if self.pfb is None else v[self._offs[self.pfb]: self._offs[self.pfb] + fw.shape[0]]
        if self.hessian:
            # forward-over-reverse through the linear head: besides H_L J v, the first-order cotangent g of the
            # logits meets the tangents of the layer's two operands (g V -> features, g^T t_feat -> weight)
            t_feat = t_last.flatten(1) if hw == 1 else t_last.mean(dim=(2, 3))
            if self.pfb is not None:
                Jv = torch.addmm(v_fb, t_feat, fw.detach().t())
            else:
                Jv = t_feat @ fw.detach().t()
            Jv = torch.addmm(Jv, self.feat, v_fw.t())
            HJv = self._loss_hessian(Jv)
            g_fw = torch.addmm(self._gl1.t() @ t_feat, HJv.t(), self.feat)
            g_fb = HJv.sum(0) if self.pfb is not None else None
            g_feat = torch.addmm(self._gl1 @ v_fw, HJv, fw.detach())
            return self._feature_cotangent(g_feat), g_fw, g_fb
        if self._head_fused(hw, v_fw):
            # ONE launch: logits' tangent, softmax-CE Hessian, the three gradients
            g_feat, g_fw, g_fb = self._head_bufs
            _lib.check(_lib.load().hf_linear_ce_head(
                _ptr(g_feat), _ptr(g_fw), _ptr(g_fb) if self.pfb is not None else None, _ptr(t_last),
                _ptr(self.feat), _ptr(fw), _ptr(v_fw), _ptr(v_fb), _ptr(self._ce[0]), float(self._ce[1]),
                g_feat.shape[0], g_feat.shape[1], fw.shape[0], _lib.HF_F32,
                _lib.current_stream_ptr(self.dev)), "hf_linear_ce_head")
            if self.pfb is None:
                g_fb = None
        else:
            t_feat = t_last.flatten(1) if hw == 1 else t_last.mean(dim=(2, 3))
            if self.pfb is not None:
                Jv = torch.addmm(v_fb, t_feat, fw.detach().t())
            else:
                Jv = t_feat @ fw.detach().t()
            Jv = torch.addmm(Jv, self.feat, v_fw.t())
            HJv = self._loss_hessian(Jv)
            g_fw = HJv.t() @ self.feat
            g_fb = HJv.sum(0) if self.pfb is not None else None
            g_feat = HJv @ fw.detach()
        return self._feature_cotangent(g_feat), g_fw, g_fb

    def _head_fused(self, hw, v_fw):
        """Whether ``hf_linear_ce_head`` applies: closed-form softmax-CE Hessian, a 1x1 final map
        (the pooling is then the identity), a small dense head, 16-byte aligned operands."""
        ok = getattr(self, "_head_ok", None)
        if ok is None:
            fw = self.fc.weight
            k, f = fw.shape
            ok = (
                self._ce is not None and hw == 1
                and fw.is_contiguous() and fw.dtype == torch.float32 and k <= 64 and f <= 512 and f % 4 == 0
                and self.feat.is_contiguous() and tuple(self.feat.shape) == (self.logits.shape[0], f)
                and self.logits.shape[0] <= 4096 and ((2 * k + 4) * f + 4 * k) * 4 <= 64 * 1024
                and self._offs[self.pfw] % 4 == 0 and self._ce[0].is_contiguous()
            )
            if ok:
                b = self.logits.shape[0]
                g = _lib.load().hf_linear_ce_head_slabs(b)  # partial sums per workgroup, added up by hf_pack_ex
                kw = dict(dtype=torch.float32, device=self.dev)
                self._head_bufs = (torch.empty((b, f), **kw), torch.empty((g, k, f), **kw),
                                   torch.empty((g, k), **kw))
            self._head_ok = ok
        return ok and v_fw.data_ptr() % 16 == 0

    def _grouping(self):
        # (Hessian products carry extra terms per unit: the plain one-unit launches)
        return not self.hessian
