"""Adjoint sweep J^T u: per unit the BatchNorm adjoint and data + weight gradient in one launch; the gather of all
parameter gradients into the flat vector -- which also leaves the PCG curvature scalar (optimizer.py:457-462, L-op).

Mixin of ``FusedGGNEngine`` (engine/core.py); see that class for the sweeps' overall structure."""

import os

import torch

from .. import _lib
from .common import _cl, _pair, _ptr


class _AdjointSweep:
    def _bn_adjoint_pair_train(self, u1, srcs1, u2, srcs2):
        """Train mode, prologue form: both units' reduction passes in one launch, both elementwise passes in one."""
        self._bn_adjoint_pair(u1, srcs1, u2, srcs2, train=True)
        arr = (_lib.AffineTrainProblem * 2)()
        for q, u in zip(arr, (u1, u2)):
            self._affine_train_problem(q, u, u.ga, 0, u.g, 1, 0, u.gw, u.gb, u.rb, None, None, None)
        _lib.check(_lib.load().hf_chan_affine_train_pair(_lib.ctypes.cast(arr, _lib.c_void_p), _lib.HF_F32,
                                                         _lib.current_stream_ptr(self.dev)), "hf_chan_affine_train_pair")

    def _bn_adjoint_pair(self, u1, srcs1, u2, srcs2, train=False):
        """The BatchNorm adjoints of two units (row-major kernel) in ONE launch.  ``train``: the reduction pass of the
        train-mode adjoint (masked cotangent and partial sums only)."""
        arr = (_lib.BnAdjointProblem * 2)()
        for q, u, srcs in zip(arr, (u1, u2), (srcs1, srcs2)):
            if not 1 <= len(srcs) <= 2:
                raise RuntimeError(f"{u.name}: {len(srcs)} consumers")
            (a, sa, la) = srcs[0]
            (b, sb, lb) = srcs[1] if len(srcs) == 2 else (None, 1, 0)
            n, k, oh, ow = u.a.shape
            q.gx, q.gw, q.gb, q.gres = (None if train else u.ga.data_ptr()), u.gw.data_ptr(), u.gb.data_ptr(), u.g.data_ptr()
            q.gy, q.gy_splits, q.gy_slab = a.data_ptr(), sa, la
            q.gy2, q.gy2_splits, q.gy2_slab = (None if b is None else b.data_ptr()), sb, lb
            q.x, q.mean, q.rstd, q.w = u.a.data_ptr(), u.mean.data_ptr(), u.rstd.data_ptr(), u.bn.weight.data_ptr()
            q.mask_src = u.y.data_ptr() if u.relu else None
            q.n, q.c, q.hw, q.row_blocks = n, k, oh * ow, u.rb
        _lib.check(_lib.load().hf_chan_affine_bwd_pair(_lib.ctypes.cast(arr, _lib.c_void_p), _lib.HF_F32,
                                                       _lib.current_stream_ptr(self.dev)), "hf_chan_affine_bwd_pair")

    def _adjoint_unit(self, u, srcs):
        """srcs: up to two (tensor, splits, slab_stride) cotangents of the unit's output."""
        second = self._second and u.bn is not None
        self._bn_adjoint(u, srcs, reduce_only=second and u.train)
        if not second:
            self._conv_adjoint(u)
            return
        # ---- Hessian product: the tangent of the backward sweep through this unit --------------------------
        lib, st = _lib.load(), _lib.current_stream_ptr(self.dev)
        n, k, oh, ow = u.a.shape
        v = self._v
        v_gamma = v[self._offs[u.pg]: self._offs[u.pg] + k] if u.pg is not None else self._zeros(k)  # (frozen scale)
        if not self._extras_parallel:
            self._hessian_extras(u)
        if u.train:
            # batch statistics: the per-channel finalisation of the five row reductions (this pass's g_z' sums, the
            # tangent sweep's a' sums), then  g_a' = c0 g_a + c1 g_z + c2 g_z' + c3 a' + c4 xhat + c5
            rb = u.rb
            tx, t1, nparts_t = u.tsums
            _lib.check(lib.hf_bn_train_hessian_coeffs(
                _ptr(u.hcoef), _ptr(u.gw[2 * rb:]), _ptr(u.gw), _ptr(u.gb), _ptr(u.gw[rb:]), rb, _ptr(tx), _ptr(t1),
                nparts_t, _ptr(u.gg1), _ptr(u.gb1), _ptr(u.scale), _ptr(v_gamma), _ptr(u.rstd), float(n * oh * ow), k,
                _lib.HF_F32, st), "hf_bn_train_hessian_coeffs")
            _lib.check(lib.hf_bn_train_hessian_apply(
                _ptr(u.gah), _ptr(u.ga1), _ptr(u.g1), _ptr(u.g), _ptr(u.tbuf), u.sT, u.tbuf.shape[1], _ptr(u.a),
                _ptr(u.mean), _ptr(u.rstd), _ptr(u.hcoef), n * oh * ow, k, _lib.HF_F32, st),
                "hf_bn_train_hessian_apply")
        else:
            # the convolution's cotangent tangent:  g_a' + g_z * rstd * v_gamma
            _lib.check(lib.hf_chan_affine_ex(
                _ptr(u.gah), _ptr(u.g1), None, None, _ptr(u.rstd), _ptr(v_gamma), None, None, _ptr(u.ga), None, 0, n,
                k, oh * ow, 1, 0, 0, 1, 0, _lib.HF_F32, st), "hf_chan_affine_ex")
        if self._extras_mode == 2 and not u.im2col and not u.first and u.sD:
            # the chain's launch also computes conv_D(g_a, V) -- the one extra term the chain itself needs next
            _lib.conv_group_slabs([(1, u.dbuf, u.gah, u.wT, u.geo, u.sD, 0, 0), (2, u.wbuf, u.x, u.gah, u.geo, u.sW, 0, 0),
                                   (1, u.dbuf[u.sD:], u.ga1, u.vT, u.geo, u.sD, 0, 0)], self.dev)
        else:
            self._conv_adjoint(u, u.gah)

    def _dslabs(self, u):
        """How many data-gradient slabs the consumers of ``u``'s input cotangent must sum in the current sweep."""
        return u.nD if self._second else u.sD

    def _bn_adjoint(self, u, srcs, ga=None, reduce_only=False):
        """``reduce_only`` (train mode inside a Hessian product): the masked cotangent and its per-channel sums only --
        the elementwise pass is ``hf_bn_train_hessian_apply``'s."""
        ga = u.ga if ga is None else ga
        if not 1 <= len(srcs) <= 2:
            raise RuntimeError(f"{u.name}: {len(srcs)} consumers")
        self._extras_wait(srcs)
        (a, sa, la) = srcs[0]
        (b, sb, lb) = srcs[1] if len(srcs) == 2 else (None, 1, 0)
        lib, st = _lib.load(), _lib.current_stream_ptr(self.dev)
        n, k, oh, ow = u.a.shape
        if u.train:
            # pass 1: g = mask * (sum of the cotangents' slabs) and its per-channel sums (the parameter
            # gradients); pass 2: g_a = rstd*w * [g - mean(g) - xhat * mean(xhat*g)] (the batch statistics'
            # share), by the elementwise kernel with the corrections folded into its per-channel vectors
            # (pass 2 adds the partial rows up in its prologue)
            _lib.check(lib.hf_chan_affine_bwd_ex(
                None, _ptr(u.gw), _ptr(u.gb), _ptr(u.g), _ptr(a), sa, la, _ptr(b), sb, lb, _ptr(u.a),
                _ptr(u.mean), _ptr(u.rstd), _ptr(u.scale), _ptr(u.y) if u.relu else None, n, k, oh * ow, 1,
                u.rb, _lib.HF_F32, st), "hf_chan_affine_bwd_ex")
            if reduce_only:
                return
            _lib.check(lib.hf_chan_affine_train(
                _ptr(ga), _ptr(u.g), _ptr(u.a), _ptr(u.mean), _ptr(u.rstd), _ptr(u.scale), _ptr(u.gw),
                _ptr(u.gb), u.rb, None, None, float(n * oh * ow), None, None, n, k, oh * ow, 0, 0, 1, 0,
                _lib.HF_F32, st), "hf_chan_affine_train")
            return
        # g = mask * (sum of both cotangents' slabs) -> u.g; g * w*rstd -> u.ga; per-channel sums
        bn = u.bn is not None
        _lib.check(lib.hf_chan_affine_bwd_ex(
            _ptr(ga), _ptr(u.gw) if bn else None, _ptr(u.gb) if u.pb is not None else None,
            _ptr(u.g) if u.needs_g else None, _ptr(a), sa, la, _ptr(b), sb, lb, _ptr(u.a) if bn else None,
            _ptr(u.mean), _ptr(u.rstd), _ptr(u.scale), _ptr(u.y) if u.relu else None, n, k,
            oh * ow, 1, u.rb, _lib.HF_F32, st), "hf_chan_affine_bwd_ex")

    def _conv_adjoint(self, u, ga=None):
        """Data + weight gradient of the unit's convolution from ``ga`` (default ``u.ga``), one launch."""
        ga = u.ga if ga is None else ga
        lib, st = _lib.load(), _lib.current_stream_ptr(self.dev)
        if u.im2col:
            self._conv_slabs(2, u.wbuf, u.cols_pad, ga, u.geo_w, u.sW, out_c=u.jcols)
            return
        if u.first or u.no_dgrad:  # the network input (or the output of frozen layers) needs no gradient
            self._conv_slabs(2, u.wbuf, u.x, ga, u.geo, u.sW)
            return
        # (Measured and rejected, round 4: the weight gradient -- off the adjoint chain, only the gather reads it -- as
        # its own launch on a side branch, the chain's launch computing the data gradient alone: one cross-branch
        # dependency PER UNIT costs far more than the shorter chain saves -- ResNet-18 1 518 -> 1 037 matvecs/s,
        # ResNet-50 topology 314 -> 253, All-CNN-C 753 -> 711.  A side branch pays when it forks ONCE: the Hessian
        # products' extras below.)
        n_, h, w, c, k_, r, s, sd, pd = u.geo
        _lib.check(lib.hf_conv2d_nhwc_backward_slabs(
            _ptr(u.dbuf), _ptr(u.wbuf), _ptr(ga), _ptr(u.x), _ptr(u.wT), n_, h, w, c, k_, r, s, sd[0], sd[1],
            pd[0], pd[1], u.sD, u.dbuf.shape[1], u.sW, u.wbuf.shape[1], _lib.HF_F32, st),
            "hf_conv2d_nhwc_backward_slabs")

    # ---- adjoint sweep -------------------------------------------------------------------------
    def _adjoint_blocks(self, g_last, first=None, last_block=0, incoming=None):
        """Walks the blocks ``first`` (default: the last one) ... ``last_block`` backwards; returns the
        two cotangents of the pooled stem output -- or, when the walk stops before block 0, the state
        (``incoming``) a later call continues from (the product in two phases, ``local_phases``)."""
        group = self._grouping()
        tail = self.tail
        if incoming is None:
            incoming = {id(tail): [(g_last, 1, 0)]}
        pool_srcs = None
        first = len(self.blocks) - 1 if first is None else first
        stop = max(last_block, self.dead_blocks)  # (dead blocks: nobody needs their cotangents)
        for bi in range(first, stop - 1, -1):
            chain, ds, _x = self.blocks[bi]
            head, last = chain[0], chain[-1]
            for k in range(len(chain) - 1, -1, -1):
                u = chain[k]
                if k == 0 and ds is not None and group:
                    # both BatchNorm adjoints, then the data + weight gradients of the block's first
                    # convolution AND of its downsample branch in ONE launch (four problems)
                    if u.rb > 1 and ds.rb > 1 and not u.train and not ds.train:
                        self._bn_adjoint_pair(u, incoming.pop(id(u)), ds, [(last.g, 1, 0)])
                    elif u.rb > 1 and ds.rb > 1 and self._train_pair_ok(u, ds):
                        self._bn_adjoint_pair_train(u, incoming.pop(id(u)), ds, [(last.g, 1, 0)])
                    else:
                        self._bn_adjoint(u, incoming.pop(id(u)))
                        self._bn_adjoint(ds, [(last.g, 1, 0)])
                    if u.no_dgrad:  # (the block input carries no tangent: the two weight gradients only)
                        _lib.conv_group_slabs(
                            [(2, u.wbuf, u.x, u.ga, u.geo, u.sW, 0, 0), (2, ds.wbuf, ds.x, ds.ga, ds.geo, ds.sW, 0, 0)],
                            self.dev)
                    else:
                        _lib.conv_group_slabs(
                            [(1, u.dbuf, u.ga, u.wT, u.geo, u.sD, 0, 0), (2, u.wbuf, u.x, u.ga, u.geo, u.sW, 0, 0),
                             (1, ds.dbuf, ds.ga, ds.wT, ds.geo, ds.sD, 0, 0),
                             (2, ds.wbuf, ds.x, ds.ga, ds.geo, ds.sW, 0, 0)], self.dev)
                else:
                    self._adjoint_unit(u, incoming.pop(id(u)))
                if k > 0:
                    incoming.setdefault(id(chain[k - 1]), []).append((u.dbuf, self._dslabs(u), u.dbuf.shape[1]))
            if head.no_dgrad:  # (everything in front of this block is frozen: the sweep ends here)
                if ds is not None and not group:
                    self._adjoint_unit(ds, [(last.g, 1, 0)])
                return None
            # the block input receives conv1's data gradient and the residual branch's cotangent
            srcs = [(head.dbuf, self._dslabs(head), head.dbuf.shape[1])]
            if ds is not None:
                if not group:
                    self._adjoint_unit(ds, [(last.g, 1, 0)])
                srcs.append((ds.dbuf, self._dslabs(ds), ds.dbuf.shape[1]))
            else:
                srcs.append((last.g, 1, 0))
            if bi > 0:
                incoming[id(self.blocks[bi - 1][0][-1])] = srcs
            else:
                pool_srcs = srcs
        return pool_srcs if last_block == 0 else incoming

    def _adjoint_stem(self, pool_srcs):
        """Block 0's input is the pooled stem output: sum its two cotangents, undo the max-pool,
        then the stem's own adjoint."""
        s = self.stem
        if s.dead or pool_srcs is None:  # (frozen stem: nothing to differentiate)
            return
        ks, st_, pd, dl, cm = self.pool_args
        pn, ph, pw, poh, pow_, c0 = self._pool_geometry()
        (a, sa, la), (b, sb, lb) = pool_srcs
        self._extras_wait(pool_srcs)
        # slab sums of both cotangents and the max-pool adjoint (gather form) in one launch
        g_stem = self._g_stem
        (kh, kw), (sh, sw), (pph, ppw) = _pair(ks), _pair(st_ if st_ is not None else ks), _pair(pd)
        _lib.check(_lib.load().hf_maxpool_adjoint_nhwc(
            _ptr(g_stem), _ptr(a), sa, la, _ptr(b), sb, lb, _ptr(self.pool_idx32), pn, ph, pw, poh, pow_,
            c0, kh, kw, sh, sw, pph, ppw, _lib.HF_F32, _lib.current_stream_ptr(self.dev)),
            "hf_maxpool_adjoint_nhwc")
        self._adjoint_unit(s, [(g_stem, 1, 0)])

    def _feature_cotangent(self, g_feat):
        tail = self.tail
        if self._head_hw == 1:
            return g_feat.view(tail.y.shape)
        return _cl((g_feat / self._head_hw).view(g_feat.shape[0], -1, 1, 1).expand(tail.y.shape))

    def _gather(self, out, g_fw, g_fb, first_order=False):
        """All parameter gradients into the flat vector (weight-gradient slabs summed on the way)."""
        tensors, perms, splits = self._pack_args(first_order)
        tensors = list(tensors)
        if g_fw.dim() == 3:  # the head kernel's per-workgroup partial sums: slabs for hf_pack_ex
            splits = dict(splits)
            tensors[self.pfw] = g_fw[0]
            splits[self.pfw] = (g_fw.shape[0], g_fw[0].numel())
            if self.pfb is not None:
                tensors[self.pfb] = g_fb[0]
                splits[self.pfb] = (g_fb.shape[0], g_fb.shape[1])
        else:
            tensors[self.pfw] = g_fw
            if self.pfb is not None:
                tensors[self.pfb] = g_fb
        _lib.pack_ex(out, tensors, perms, splits, scale=self.weight, live=self._pack_live)
        return out

    def _gather_range(self, out, g_fw, g_fb, lo, hi):
        """``_gather`` for the parameters ``lo ... hi-1`` only (a contiguous range of the flat vector)."""
        tensors, perms, splits = self._pack_args()
        tensors, splits = list(tensors), dict(splits)
        if g_fw is not None:
            if g_fw.dim() == 3:
                tensors[self.pfw] = g_fw[0]
                splits[self.pfw] = (g_fw.shape[0], g_fw[0].numel())
                if self.pfb is not None:
                    tensors[self.pfb] = g_fb[0]
                    splits[self.pfb] = (g_fb.shape[0], g_fb.shape[1])
            else:
                tensors[self.pfw] = g_fw
                if self.pfb is not None:
                    tensors[self.pfb] = g_fb
        sub = lambda d: {i - lo: val for i, val in d.items() if lo <= i < hi}  # noqa: E731
        end = self._offs[hi] if hi < len(self.params) else self.n
        _lib.pack_ex(out[self._offs[lo]:end], tensors[lo:hi], sub(perms), sub(splits), scale=self.weight,
                     live=sub(self._pack_live))

    def _pack_args(self, first_order=False):
        """(tensors, perms, splits) of ``hf_pack_ex``.  ``first_order``: a gradient sweep of a Hessian
        engine fills only the first ``sW`` weight-gradient slabs of each layer."""
        if first_order and self.hessian:
            tensors, perms, splits = self._pack_args()
            splits = dict(splits)
            for u in self.units:
                if u.pw is not None and u.nW != u.sW:
                    if u.sW > 1:
                        splits[u.pw] = (u.sW, u.wbuf.shape[1])
                    else:
                        splits.pop(u.pw, None)
                if u.pg is not None and u.gw_rows != u.rb:
                    if u.rb > 1:
                        splits[u.pg] = (u.rb, u.cout)
                    else:
                        splits.pop(u.pg, None)
            return tensors, perms, splits
        if getattr(self, "_pack", None) is None:
            tensors, perms, splits = [None] * len(self.params), {}, {}
            self._pack_live = {}
            for u in self.units:
                if u.pw is None:  # (frozen weight: its gradient slabs, if the fused launch wrote any, are not gathered)
                    for pi, buf, rows in ((u.pg, u.gw, u.gw_rows), (u.pb, u.gb, u.rb)):
                        if pi is not None:
                            tensors[pi] = buf[0]
                            if rows > 1:
                                splits[pi] = (rows, u.cout)
                    continue
                tensors[u.pw] = u.wbuf[0]
                if not u.im2col:
                    k, c, r, s_ = u.conv.weight.shape
                    if r * s_ > 1:
                        perms[u.pw] = (c, r * s_)  # stored (O, H, W, I); 1x1 kernels: already in order
                        if u.live:
                            self._pack_live[u.pw] = u.live
                if u.nW > 1:
                    splits[u.pw] = (u.nW, u.wbuf.shape[1])
                for pi, buf, rows in ((u.pg, u.gw, u.gw_rows), (u.pb, u.gb, u.rb)):
                    if pi is not None:
                        tensors[pi] = buf[0]
                        if rows > 1:
                            splits[pi] = (rows, u.cout)
            self._pack = (tensors, perms, splits)
        return self._pack
