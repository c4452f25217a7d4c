"""The plain conv-ReLU stack engine (All-CNN-C, BASELINE.json configs[3])."""

import torch
from torch import nn

from .. import _lib
from .common import _Unit, _Unsupported, _cl, _ptr, _same
from .core import FusedGGNEngine


class PlainStackEngine(FusedGGNEngine):
    """The same sweeps for a plain stack ``[Dropout] conv(+bias) [ReLU] ... -> AdaptiveAvgPool2d(1) ->
    flatten`` with a softmax cross-entropy on the pooled map -- the All-CNN-C of the reference's
    examples (examples/example_utils.py:59-83, BASELINE.json configs[3]).  Per layer: tangent
    convolution -> bias tangent + ReLU mask (into the next layer's operand) / masked slab sum + bias
    gradient -> data + weight gradient; the head (average pool, loss Hessian, broadcast) is ONE launch
    (``hf_pool_ce_head``).  3 launches per layer and product, bitwise repeatable."""

    mode = ("fused curvature engine (plain conv-ReLU stack): own deterministic convolutions (split-K slabs summed "
            "by the consumer kernel), bias / ReLU fused, 4 launches per layer")
    # Hessian products (optimizer.py:450-455) by forward-over-reverse on the same kernels: the tangent
    # sweep, then the TANGENT OF THE BACKWARD SWEEP -- per layer, besides the GGN's conv_D(g', W) and
    # conv_W(x, g'), the two terms that carry the network's own curvature, conv_D(g, V) and conv_W(t_x, g)
    # (g: first-order cotangent of the step, g': its tangent; ReLU masks are piecewise constant), all four
    # in ONE grouped launch whose extra results are simply more split-K slabs for the consumers to sum.
    supports_hessian = True
    _extras_default = 1

    def _layout(self, model):
        x_in = getattr(self.outputs, "_hf_input", None)
        if x_in is None or x_in.dim() != 4:
            raise _Unsupported("no recorded input")
        if model.training:
            raise _Unsupported("the model must be in eval mode")
        leaves = [m for m in model.modules() if not list(m.children())]

        def io(m):
            rec = getattr(m, "_hf_io", None)
            if rec is None or len(rec) != 2 or rec[0] is None:
                raise _Unsupported(f"{type(m).__name__} has no record of this forward pass")
            return rec

        units, cur, prev, pooled, i = [], x_in.detach(), "input", None, 0
        while i < len(leaves):
            m = leaves[i]
            if isinstance(m, (nn.Dropout, nn.Dropout2d, nn.Identity)):
                i += 1
                continue
            if pooled is not None:
                raise _Unsupported(f"{type(m).__name__} after the pooling layer")
            if type(m) is nn.Conv2d:
                if m.groups != 1 or tuple(m.dilation) != (1, 1) or not getattr(m, "_hf_channels_last", False):
                    raise _Unsupported(f"conv {len(units)}: needs prepare_model(channels_last=True), groups = dilation = 1")
                cx, cy = io(m)
                if not _same(cx, cur):
                    raise _Unsupported(f"conv {len(units)}: input is not the previous activation")
                u = _Unit(f"conv{len(units)}", m, None)
                y, u.relu = cy, False
                nxt = leaves[i + 1] if i + 1 < len(leaves) else None
                if type(nxt) is nn.ReLU and not nxt.inplace:
                    rx, ry = io(nxt)
                    if _same(rx, cy):
                        y, u.relu = ry, True
                        i += 1
                u.kx, u.ky, u.rx, u.ry, u.ra = cx.data_ptr(), y.data_ptr(), cx, y, cy
                u.a, u.y, u.needs_g = _cl(cy), _cl(y), False
                u.pw = self._param(m.weight)
                u.pb = self._param(m.bias) if m.bias is not None else None
                u.pg = None
                u.src, prev, cur = prev, u, y
                units.append(u)
            elif isinstance(m, nn.AdaptiveAvgPool2d) and m.output_size in (1, (1, 1)):
                px, py = io(m)
                if not units or not _same(px, cur):
                    raise _Unsupported("the pooling layer does not follow the last convolution")
                pooled = py
            else:
                raise _Unsupported(f"unsupported layer {type(m).__name__}")
            i += 1
        if pooled is None or not units:
            raise _Unsupported("not a conv stack that ends in global average pooling")
        out = self.outputs.detach()
        if out.data_ptr() != pooled.data_ptr() or tuple(out.shape) != (pooled.shape[0], pooled.shape[1]):
            raise _Unsupported("the network output is not the flattened pooled map")
        first = units[0]
        first.first = True
        first.im2col = first.a.shape[1] > 0 and first.rx.shape[1] < 4
        self.units, self.tail, self.blocks = units, units[-1], []
        self.stem, self.pool_args, self.fc, self.pfw, self.pfb = None, None, None, None, None
        self.model_ref, self._in_shape = model, tuple(x_in.shape)
        used = {i for u in units for i in (u.pw, u.pb) if i is not None}
        if used != set(range(len(self.params))):
            raise _Unsupported("the parameter list has entries the engine's layers do not cover")
        # frozen layers at the input end (round 6, as FusedGGNEngine._mark_dead_prefix): dead for both sweeps; the first
        # layer with a trainable parameter reads a tangent-free input and needs no data gradient
        self.dead_units = 0
        for u in units:
            if u.pw is not None or u.pb is not None:
                break
            u.dead = True
            self.dead_units += 1
        if self.dead_units == len(units):
            raise _Unsupported("every layer is frozen")
        if self.dead_units:
            units[self.dead_units].no_dgrad = True

    def _allocate_pool(self):
        pass

    def _allocate_head(self):
        f32, dev = torch.float32, self.dev
        n, k, h, w = self.tail.y.shape
        self._head_hw = h * w
        self.logits = torch.empty((n, k), dtype=f32, device=dev)
        self._p = torch.empty_like(self.logits)
        self.loss_buf = torch.zeros((), dtype=f32, device=dev)
        self._g_last = torch.empty_like(self.tail.y)
        if k > 1024:
            raise _Unsupported("more than 1024 classes")

    # ---- forward -----------------------------------------------------------------------------
    def forward_own(self, refresh=False, update_running=True):  # (no BatchNorm in a plain stack: nothing to move)
        first, flat = self.units[0], self._flat_params
        carried = (refresh and flat is not None and flat.data_ptr() == self.params[0].data_ptr()
                   and self._conv_carrying_scatter(first, first.conv.weight.detach(), first.sF, flat, 0))
        if refresh and not carried:
            self.refresh_weights()
        for u in self.units:
            if not (carried and u is first):
                self._conv_forward(u)
            self._bn_forward(u, u.sF)
        torch.mean(self.tail.y, dim=(2, 3), out=self.logits)
        if getattr(self, "loss_spec", None) is not None:
            self._loss_head()
        self._at = "own"
        return self.logits

    # ---- product ------------------------------------------------------------------------------
    def _tangent_sweep(self, v):
        first = self.units[0]
        carried = False
        if first.im2col and not first.dead:  # (the first layer's launch carries the v_W scatter of all the others)
            vw = (v[self._offs[first.pw]: self._offs[first.pw] + first.conv.weight.numel()] if first.pw is not None
                  else self._zeros(first.conv.weight.numel()))
            carried = self._conv_carrying_scatter(first, vw, first.sT, v, 1)
        if self._slot_list and not carried:
            _lib.unpack_tangent(v, self._slot_list)  # v_W halves of all [W | v_W] operands: one launch
        for u in self.units:
            if u.dead:  # (frozen, behind frozen layers only: no tangent)
                continue
            if u.im2col:  # no input tangent: conv(x, v_W) as a 1x1 product over the im2col
                vw = (v[self._offs[u.pw]: self._offs[u.pw] + u.conv.weight.numel()] if u.pw is not None
                      else self._zeros(u.conv.weight.numel()))
                if not (carried and u is first):
                    self._conv_slabs(0, u.tbuf, u.cols, vw, u.geo, u.sT)
            else:
                self._conv_slabs(0, u.tbuf, u.xcat, u.wcat, self._tgeo(u), u.sT)
            self._bn_tangent(u, v, None, 0)

    def _adjoint_sweep(self, g_last, first_order=False):
        """``first_order``: the gradient's sweep (cotangents kept in ``ga1`` for later Hessian products);
        otherwise the product's sweep -- for a Hessian engine with the two extra convolutions per layer."""
        second = self.hessian and not first_order
        srcs = [(g_last, 1, 0)]
        self._second = second  # (the chain waits for the side branch's slabs, _extras_wait)
        for u in reversed(self.units):
            if u.dead:  # (nobody needs a cotangent behind the first trainable layer)
                break
            ga = u.ga1 if (self.hessian and first_order) else u.ga
            self._bn_adjoint(u, srcs, ga)
            if second and not u.im2col and not u.first and u.sD:
                # (two launches of two problems each; four in one grouped launch ran 3x slower -- with four
                # by-value problem descriptions hipcc spills them to scratch memory.  The second one -- conv_D(g, V),
                # conv_W(t_x, g): no dependence on this chain -- runs on the side branch, see _extras_fork)
                if self._extras_mode == 2:
                    _lib.conv_group_slabs([(1, u.dbuf, ga, u.wT, u.geo, u.sD, 0, 0), (2, u.wbuf, u.x, ga, u.geo, u.sW, 0, 0),
                                           (1, u.dbuf[u.sD:], u.ga1, u.vT, u.geo, u.sD, 0, 0)], self.dev)
                else:
                    _lib.conv_dw_slabs((1, u.dbuf, ga, u.wT, u.geo, u.sD, 0, 0),
                                       (2, u.wbuf, u.x, ga, u.geo, u.sW, 0, 0), self.dev)
                if not self._extras_parallel:
                    self._hessian_extras(u)
            else:
                self._conv_adjoint(u, ga)
            if u.sD:
                srcs = [(u.dbuf, u.nD if second else u.sD, u.dbuf.shape[1])]

    def local(self, v, out=None):
        if out is None:
            out = torch.empty(self.n, dtype=torch.float32, device=self.dev)
        v = v.detach()
        if not v.is_contiguous():
            v = v.contiguous()
        self._tangent_sweep(v)
        if self.hessian and self._vt_slots:
            _lib.unpack_tangent(v, self._vt_slots, half=2)  # V as (I, H, W, O): the operand of conv_D(g, V)
        n, k = self.logits.shape
        _lib.check(_lib.load().hf_pool_ce_head(
            _ptr(self._g_last), None, _ptr(self.tail.tout), _ptr(self._ce[0]), float(self._ce[1]), n, self._head_hw, k,
            _lib.HF_F32, _lib.current_stream_ptr(self.dev)), "hf_pool_ce_head")
        if self.hessian:
            self._extras_fork()
        try:
            self._adjoint_sweep(self._g_last)
            if self.hessian:
                self._extras_join()
        finally:
            self._second = False
        self._gather(out, None, None)
        if self.hessian and self._l2 is not None:  # the regulariser's Hessian: coef on its tensors' entries
            out.addcmul_(self._l2, v, value=self.weight)
        return out

    def gradient(self, out=None):
        if out is None:
            out = torch.empty(self.n, dtype=torch.float32, device=self.dev)
        n, k, h, w = self.tail.y.shape
        g = (self._p - self._onehot) * (self._ce[1] / self._head_hw)  # d loss / d (last map), per pixel
        self._g_last.permute(0, 2, 3, 1).copy_(g.view(n, 1, 1, k).expand(n, h, w, k))
        self._adjoint_sweep(self._g_last, first_order=True)
        self._gather(out, None, None, first_order=True)
        if self._l2 is not None:
            out.addcmul_(self._l2, self._theta(), value=self.weight)
        return out

    def _gather(self, out, g_fw, g_fb, first_order=False):
        tensors, perms, splits = self._pack_args(first_order)
        _lib.pack_ex(out, list(tensors), perms, splits, scale=self.weight, live=self._pack_live)
        return out

    def _loss_setup(self, loss, outputs):
        FusedGGNEngine._loss_setup(self, loss, outputs)
        if self.loss_spec is None or self.loss_spec["kind"] != "ce":
            raise _Unsupported("the plain-stack engine needs a plain softmax cross-entropy loss")
