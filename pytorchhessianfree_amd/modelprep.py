"""Optional, opt-in model preparation for faster curvature products
(``prepare_model(model)``; every patch keeps the ``Parameter`` objects, their order, the
state_dict and the flat-vector layout, and falls back to the stock op for CPU tensors,
training mode or layers it does not recognise).

A GGN product differentiates the network twice (BackPACK's R-op is a backward of a
backward, ``optimizer.py:461``); PyTorch's generic double-backward formulas issue many
small kernels, and on an MI355X a product of a ResNet-18 is bound by the NUMBER of
kernels (~4.5 us each inside a hipGraph), not by bytes or flops.  The patches:

* ``fuse_eval_batchnorm`` -- an eval-mode BatchNorm is the per-channel affine map
  ``y = xhat * w + b``, ``xhat = (x - running_mean) * rsqrt(running_var + eps)``.  The
  generic double-backward of ``native_batch_norm`` is ~40 kernels per layer and product
  (~800 of the ~1050 of a stock ResNet-18 product); here forward, backward and
  backward-of-backward are ONE HIP kernel each (``hf_chan_affine`` / ``hf_chan_affine_bwd``).
* ``fuse_residual_blocks`` -- ``relu(bn(.))`` / ``relu(bn(.) + identity)`` of ResNet blocks
  as one such layer; the block output's two consumers return their cotangents separately
  and the kernel adds them.
* ``fuse_conv_tangent`` -- a conv layer's tangent map ``conv(v_x, W) + conv(x, v_W)`` as ONE
  convolution over concatenated input channels; ``v_W`` arrives by one multi-tensor
  scatter per product (``hf_unpack_tangent``), ``v_x`` is written in place by the
  preceding fused layer; optionally everything in NHWC (no MIOpen layout transposes);
  convolutions on 1x1 maps as GEMMs on the kernel's centre tap; bias gradients by a
  reduction kernel that is safe inside a hipGraph.
* ``fuse_bn_relu`` -- a ``ReLU`` module fed by a fused BatchNorm (ResNet stem, Sequential
  stacks) becomes part of that layer.
* ``skip_identity_pools`` -- ``AdaptiveAvgPool2d(1)`` of a 1x1 map issues no kernel.

``first_order_only`` / ``tangent_owner`` are the two contexts the curvature operators hold
while recording ``u -> J^T u`` and while sweeping a tangent through it.
"""

import os
import threading
import types

import torch
from torch import nn

from . import _lib


class _Mode:
    first_order_only = False
    owner = None      # operator running its tangent sweep (see ``tangent_owner``)
    vector = None     # the flat vector that sweep differentiates against
    prefilled = False
    producers = {}    # data_ptr -> (ctx, tensor) of this sweep's fused-layer tangent outputs
    consume_now = False  # adjoint sweep whose per-parameter results are gathered at once
    session = None       # persistent engine session that may answer the model's forward pass


class first_order_only:
    """Context manager: inside it, the backward of the fused layers is recorded
    WITHOUT its dependence on the layer input / weight (only on the incoming
    cotangent).  ``GGNOperator`` records ``u -> J^T u`` under it: a GGN product only
    ever differentiates that map with respect to ``u``, but autograd cannot know and
    would evaluate the (discarded) second-order terms on every product."""

    def __enter__(self):
        self._old = _Mode.first_order_only
        _Mode.first_order_only = True

    def __exit__(self, *exc):
        _Mode.first_order_only = self._old
        return False


class consumed_at_once:
    """Context manager a curvature operator holds around its adjoint sweep: the
    per-parameter gradients are gathered into the flat vector before anything else runs, so
    a conv layer may return its persistent (partly constant-zero) weight-gradient buffer
    instead of a fresh tensor.  Outside it every backward returns its own tensor."""

    def __enter__(self):
        self._old = _Mode.consume_now
        _Mode.consume_now = True

    def __exit__(self, *exc):
        _Mode.consume_now = self._old
        return False


def _dims(x):
    n, c = x.shape[0], x.shape[1]
    hw = x.numel() // (n * c) if x.numel() else 1
    return n, c, hw


def _p(t):
    return _lib.c_void_p(t.data_ptr()) if t is not None else None


def _is_cl(t):
    """True iff ``t`` is a 4-D tensor stored NHWC and NOT also NCHW-contiguous (for
    C == 1 or H == W == 1 the two layouts coincide and NCHW is reported)."""
    return t.dim() == 4 and not t.is_contiguous() and t.is_contiguous(memory_format=torch.channels_last)


def _dense(t):
    """NCHW- or NHWC-dense version of ``t`` (the fused kernels handle both)."""
    if t is None or t.is_contiguous() or _is_cl(t):
        return t
    return t.contiguous()


def _like(t, ref):
    """``t`` in the memory layout of ``ref`` (copy only if it differs)."""
    if t is None:
        return None
    if _is_cl(ref):
        return t.contiguous(memory_format=torch.channels_last)
    return t.contiguous()


def _slice_ld(t, like):
    """Leading dimension if ``t`` (shaped like ``like``) is the first-channels slice of a
    wider buffer in ``like``'s layout -- NHWC: (row, c) at row*ld + c; NCHW: (n, c, hw) at
    n*ld + c*HW + hw --, 0 if it is dense in that layout, None if it is neither."""
    if t.shape != like.shape or t.dim() != 4 or t.dtype != like.dtype:
        return None
    n, c, h, w = t.shape
    sn, sc, sh, sw = t.stride()
    if _is_cl(like):
        ld = sw if w > 1 else (sh if h > 1 else (sn if n > 1 else c))
        ok = (c == 1 or sc == 1) and (w == 1 or sw == ld) and (h == 1 or sh == w * ld) and (n == 1 or sn == h * w * ld)
        dense = c
    else:
        ld = sn if n > 1 else c * h * w
        ok = (w == 1 or sw == 1) and (h == 1 or sh == w) and (c == 1 or sc == h * w)
        dense = c * h * w
    if not ok or ld < dense:
        return None
    return 0 if ld == dense else ld


def _affine(a, x, mean, rstd, w, q, r, like, add=None, mask_src=None, relu_self=False, out=None):
    """out = act(a*(w*rstd) + xhat*q + r + add) (nullable terms), one launch; all
    activation-sized operands share ``like``'s layout (NCHW or NHWC).  ``out`` / ``add``
    may be first-channels slices of wider buffers in that layout (``_slice_ld``)."""
    out_ld = 0
    if out is None:
        out = torch.empty_like(like)
    else:
        out_ld = _slice_ld(out, like)
        if out_ld is None:
            raise RuntimeError("_affine: unusable output slice")
    add_ld = 0
    if add is not None:
        add_ld = _slice_ld(add, like)
        if add_ld is None:
            add, add_ld = _like(add, like), 0
    n, c, hw = _dims(like)
    _lib.check(
        _lib.load().hf_chan_affine(
            _p(out), _p(a), _p(x), _p(mean), _p(rstd), _p(w), _p(q), _p(r), _p(add), _p(mask_src),
            1 if relu_self else 0, n, c, hw, 1 if _is_cl(like) else 0, out_ld, add_ld,
            _lib.dtype_code(like.dtype), _lib.current_stream_ptr(like.device)),
        "hf_chan_affine")
    return out


def _affine_bwd(gy, x, mean, rstd, w, mask_src=None, need_gres=False, gy2=None):
    n, c, hw = _dims(x)
    gx = torch.empty_like(x)
    gres = torch.empty_like(x) if need_gres else None
    gw = torch.empty(c, dtype=x.dtype, device=x.device)
    gb = torch.empty(c, dtype=x.dtype, device=x.device)
    _lib.check(
        _lib.load().hf_chan_affine_bwd(
            _p(gx), _p(gw), _p(gb), _p(gres), _p(gy), _p(gy2), _p(x), _p(mean), _p(rstd), _p(w),
            _p(mask_src), n, c, hw, 1 if _is_cl(x) else 0, _lib.dtype_code(x.dtype),
            _lib.current_stream_ptr(x.device)),
        "hf_chan_affine_bwd")
    return gx, gw, gb, gres


def _bshape(x):
    return [1, -1] + [1] * (x.dim() - 2)




class _ChanAffineBwd(torch.autograd.Function):
    """(gy [, gy2]; x, w, y) -> (gx, gw, gb, gres) with g = (gy + gy2) * [y > 0] when the
    layer ends in a ReLU.  Linear in the cotangents; its transpose is one ``_affine``
    launch.  ``gy2`` is the cotangent of the output's twin (``_ChanAffine`` with
    ``twin=True``): the two consumers of a residual block's output hand their cotangents
    over separately and the kernel adds them, instead of autograd's own add kernel."""

    @staticmethod
    def forward(ctx, gy, gy2, x, w, mean, rstd, y, has_res):
        if gy is None:
            gy, gy2 = gy2, None
        gy, gy2 = _like(gy, x), _like(gy2, x)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(gy, gy2, x, w, mean, rstd, y)
        ctx.has_res = has_res
        gx, gw, gb, gres = _affine_bwd(gy, x, mean, rstd, w, mask_src=y,
                                       need_gres=has_res and (y is not None or gy2 is not None), gy2=gy2)
        if has_res and gres is None:
            gres = gy  # no ReLU, one cotangent: the residual branch receives it as it is
        return gx, gw, gb, gres

    @staticmethod
    def backward(ctx, vgx, vgw, vgb, vgres):
        gy, gy2, x, w, mean, rstd, y = ctx.saved_tensors
        if vgx is None and vgw is None and vgb is None and vgres is None:
            return (None,) * 8
        # d/d gy : one fused launch (this is the layer's tangent map) ...
        # ... written straight into the next convolution's [v_x | x] operand once that
        # layer has claimed this output (``_claim_direct``); the residual branch reads its
        # operand from such a slice without a copy as well
        direct = getattr(ctx, "_hf_direct", None)
        v_gy = _affine(_like(vgx, x), x, mean, rstd, w,
                       None if vgw is None else vgw.contiguous(),
                       None if vgb is None else vgb.contiguous(), like=x, add=vgres,
                       mask_src=y, out=direct)
        if direct is None and _Mode.owner is not None:
            _Mode.producers[v_gy.data_ptr()] = (ctx, v_gy)  # holding v_gy pins the address
        v_x = v_w = None
        # second-order terms, only for Hessian products (plain ATen, rare path);
        # the ReLU mask is piecewise constant, so it only gates the cotangent
        if (ctx.needs_input_grad[2] and vgw is not None) or (ctx.needs_input_grad[3] and vgx is not None):
            g = gy if gy2 is None else gy + gy2
            g = g if y is None else g * (y > 0)
            if ctx.needs_input_grad[2] and vgw is not None:
                v_x = g * (vgw * rstd).view(_bshape(x))
            if ctx.needs_input_grad[3] and vgx is not None:
                red = [d for d in range(x.dim()) if d != 1]
                v_w = (vgx * g).sum(red) * rstd
        return (v_gy if ctx.needs_input_grad[0] else None, v_gy if ctx.needs_input_grad[1] else None,
                v_x, v_w, None, None, None, None)


class _ChanAffine(torch.autograd.Function):
    """y = act(xhat * w + b + res) with fixed statistics; act = ReLU or identity.  With
    ``twin`` the output is returned twice (the second an alias of the first) so that its two
    consumers' cotangents reach ``backward`` separately."""

    @staticmethod
    def forward(ctx, x, w, b, mean, rstd, res, relu, twin=False):
        x = _dense(x)
        res = _like(res, x)
        y = _affine(None, x, mean, rstd, None, w, b, like=x, add=res, relu_self=relu)
        ctx.save_for_backward(x, w, mean, rstd, y if relu else None)
        ctx.has_res = res is not None
        ctx.set_materialize_grads(False)  # an unused twin must not cost a zero-fill and an add
        # the twin shares y's storage but is not a view OF y (detach): tagging y with it
        # must not close a reference cycle that would keep the whole graph alive
        return (y, y.detach()) if twin else y

    @staticmethod
    def backward(ctx, gy, gy2=None):
        x, w, mean, rstd, y = ctx.saved_tensors
        if _Mode.first_order_only:
            x, w = x.detach(), w.detach()
        if gy is None and gy2 is None:
            return (None,) * 8
        gx, gw, gb, gres = _ChanAffineBwd.apply(gy, gy2, x, w, mean, rstd, y, ctx.has_res)
        return gx, gw, gb, None, None, gres, None, None


def _bn_usable(bn, x):
    return (
        isinstance(bn, (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d))
        and not bn.training and bn.track_running_stats and bn.running_var is not None
        and x.is_cuda and bn.weight is not None and bn.bias is not None
        and x.dtype in (torch.float32, torch.float64)
        # (the input ranks the stock layer itself accepts -- anything else goes to its forward, which raises its own
        # error: an unbatched [C, H, W] sample, as the reference's per-sample loop preconditioners.py:91-99 feeds it,
        # is a ValueError for nn.BatchNorm2d, not a layer that silently takes dimension 1 for the channels)
        and x.dim() in {nn.BatchNorm1d: (2, 3), nn.BatchNorm2d: (4,), nn.BatchNorm3d: (5,)}[
            next(k for k in (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d) if isinstance(bn, k))]
    )


def fused_bn_act(bn, x, res=None, relu=False, twin=False):
    """``act(bn(x) + res)`` -- one HIP launch per pass when ``bn`` is an eval-mode
    BatchNorm on a GPU tensor, the stock ops otherwise.  ``twin``: tag the result with an
    alias (``y._hf_twin``) for its second consumer, see ``_ChanAffine``."""
    if not _bn_usable(bn, x):
        bn._hf_io = None
        fwd = getattr(bn, "_hf_stock_forward", None) or bn.forward
        y = fwd(x)
        if res is not None:
            y = y + res
        y = torch.relu(y) if relu else y
        if (bn.training and isinstance(bn, nn.BatchNorm2d) and x.is_cuda and x.dim() == 4 and bn.weight is not None
                and bn.bias is not None and x.dtype == torch.float32):
            # train mode (stock ops above): what the curvature engine needs of this pass -- the batch
            # statistics the layer normalised with, as a 6-entry record
            with torch.no_grad():
                var, mean = torch.var_mean(x.detach(), dim=(0, 2, 3), unbiased=False)
                rstd = torch.rsqrt(var + bn.eps)
            bn._hf_io = (x.detach(), None if res is None else res.detach(), y.detach(), relu, rstd, mean)
        return y
    rstd = torch.rsqrt(bn.running_var + bn.eps)
    if not twin:
        y = _ChanAffine.apply(x, bn.weight, bn.bias, bn.running_mean, rstd, res, relu)
        bn._hf_io = (x.detach(), None if res is None else res.detach(), y.detach(), relu, rstd)
        return y
    y, y2 = _ChanAffine.apply(x, bn.weight, bn.bias, bn.running_mean, rstd, res, relu, True)
    y._hf_twin = y2
    bn._hf_io = (x.detach(), None if res is None else res.detach(), y.detach(), relu, rstd)
    return y


def _fused_forward(self, x):
    y = fused_bn_act(self, x)
    if _bn_usable(self, x):
        y._hf_bn_src = (self, x)  # lets a directly following ReLU module fuse with this layer
    elif self.training and getattr(self, "_hf_io", None) is not None:
        y._hf_bn_train_src = self  # train mode: a following ReLU module completes this layer's record
    return y


def _relu_forward(self, x):
    src = getattr(x, "_hf_bn_src", None)
    if src is None:
        y = self._hf_stock_forward(x)
        bn = getattr(x, "_hf_bn_train_src", None)
        if bn is not None and getattr(bn, "_hf_io", None) is not None and len(bn._hf_io) == 6:
            rec = bn._hf_io  # (train-mode BatchNorm, stock ops): the layer's output is the ReLU's
            bn._hf_io = (rec[0], rec[1], y.detach(), True, rec[4], rec[5])
        return y
    bn, xin = src
    return fused_bn_act(bn, xin, relu=True)  # the unfused BatchNorm output drops out of the graph


def fuse_bn_relu(model):
    """``ReLU`` modules whose input is the output of a fused eval-mode BatchNorm (a ResNet
    stem, ``Sequential(conv, bn, relu)`` stacks) evaluate ``relu(bn(x))`` as ONE fused layer:
    the separately computed BatchNorm output is left unused, so every curvature product
    runs one kernel per pass instead of two.  Needs ``fuse_eval_batchnorm``.  Returns the
    number of modules patched."""
    count = 0
    for m in model.modules():
        # (an in-place ReLU rewrites the BatchNorm output that other consumers may still read;
        # the fused layer would leave it untouched -- such modules keep their stock forward)
        if type(m) is nn.ReLU and not m.inplace and not hasattr(m, "_hf_stock_forward"):
            m._hf_stock_forward = m.forward
            m.forward = types.MethodType(_relu_forward, m)
            count += 1
    return count


# ------------------------------------------------------------------------------------
# residual blocks: BN + ReLU and BN + identity + ReLU as one layer each
# ------------------------------------------------------------------------------------
def _second_consumer_view(x):
    """The alias a fused layer attached to its output for its second consumer (the
    residual branch), or ``x`` itself: cotangents of ``x`` and of the alias reach the
    producing layer separately and are added inside its adjoint kernel."""
    twin = getattr(x, "_hf_twin", None)
    return x if twin is None else twin


def _basic_block_forward(self, x):
    x2 = _second_consumer_view(x)
    idt = x2 if self.downsample is None else self.downsample(x2)
    out = fused_bn_act(self.bn1, self.conv1(x), relu=True)
    return fused_bn_act(self.bn2, self.conv2(out), res=idt, relu=True, twin=True)


def _bottleneck_forward(self, x):
    x2 = _second_consumer_view(x)
    idt = x2 if self.downsample is None else self.downsample(x2)
    out = fused_bn_act(self.bn1, self.conv1(x), relu=True)
    out = fused_bn_act(self.bn2, self.conv2(out), relu=True)
    return fused_bn_act(self.bn3, self.conv3(out), res=idt, relu=True, twin=True)


def _verified_block_forward(fused):
    """Wrap a fused block forward: the FIRST call also runs the module's own (stock) forward
    on the same input and compares.  Recognition is structural (attribute names); a custom
    block with the same attributes but a different function (squeeze-excite, dropout,
    stochastic depth, ...) fails the comparison, gets its stock forward back for good and a
    warning is issued -- the network function is never changed silently."""

    def forward(self, x):
        y = fused(self, x)
        # (in training mode the fused layers are the stock ops anyway, and a second stock
        # forward would update the BatchNorm running statistics twice)
        if not self._hf_block_verified and not self.training:
            self._hf_block_verified = True
            keep = [(m, m._hf_io) for m in self.modules() if getattr(m, "_hf_io", None) is not None]
            with torch.no_grad():
                want = self._hf_stock_block_forward(x.detach())
            for m, rec in keep:  # the stock pass must not replace the fused pass's records
                m._hf_io = rec
            scale = float(want.abs().max()) + 1e-30
            if want.shape != y.shape or not float((want - y.detach()).abs().max()) <= 1e-4 * scale:
                import warnings

                warnings.warn(
                    f"{type(self).__name__}: fused residual-block forward does not reproduce the module's "
                    "own forward; keeping the stock forward for this block")
                self.forward = self._hf_stock_block_forward
                return self._hf_stock_block_forward(x)
        return y

    return forward


def _looks_like(block, convs):
    names = [f"conv{i}" for i in range(1, convs + 1)] + [f"bn{i}" for i in range(1, convs + 1)]
    return (
        all(isinstance(getattr(block, n, None), nn.Module) for n in names)
        and not hasattr(block, f"conv{convs + 1}") and hasattr(block, "downsample")
        and isinstance(getattr(block, "relu", None), nn.ReLU)
    )


def fuse_residual_blocks(model):
    """Patch ResNet ``BasicBlock`` / ``Bottleneck`` modules (torchvision's or any module
    with the same attributes ``conv1..k, bn1..k, relu, downsample`` and the standard
    forward) so that ``relu(bn(.))`` and ``relu(bn(.) + identity)`` are single fused
    layers in every pass of the curvature product.  Recognition is by attribute
    structure, not by tracing; anything else is left alone.  Returns the number of
    blocks patched."""
    count = 0
    for m in model.modules():
        if hasattr(m, "_hf_block_patched"):
            continue
        if _looks_like(m, 2):
            fused = _basic_block_forward
        elif _looks_like(m, 3):
            fused = _bottleneck_forward
        else:
            continue
        m._hf_stock_block_forward = m.forward
        m._hf_block_verified = False
        m.forward = types.MethodType(_verified_block_forward(fused), m)
        m._hf_block_patched = True
        count += 1
    return count


# ------------------------------------------------------------------------------------
# convolution: one launch for the tangent map
# ------------------------------------------------------------------------------------
def _fmt(t, cl):
    if t is None:
        return None
    return t.contiguous(memory_format=torch.channels_last) if cl else t.contiguous()


class _miopen_mode:
    """NHWC convolutions run in MIOpen's IMMEDIATE mode (``cudnn.benchmark`` off for the
    duration of the call): the solver is the one the find-db record of the shape ranks
    first (``miopen_db/`` ships the records of the BASELINE.json workloads) or, without
    a record, MIOpen's heuristic choice -- never the outcome of a find step.  Measured:
    on a machine with a cold kernel cache (every fresh box) MIOpen's find step
    re-measures even shapes that have a record, rankings between close candidates flip
    from process to process, and with NHWC operands one of the candidates is only good
    to 1.5e-4; in immediate mode on the shipped records 40 of 40 processes reproduced
    the float64 product to 3e-7 (scripts/experiments/nhwc_mode.sh, nhwc_diag.sh).
    ``HF_NHWC_FIND=1`` keeps the find step (used once, result checked, to produce the
    shipped records).  NCHW calls are left as configured."""

    _lock = threading.Lock()
    _depth = 0      # the flag is process-global and the autograd engine calls in from its own
    _saved = None   # threads: only the outermost entry saves, only the last exit restores

    def __init__(self, cl):
        self.active = bool(cl) and not os.environ.get("HF_NHWC_FIND")

    def __enter__(self):
        if self.active:
            with _miopen_mode._lock:
                if _miopen_mode._depth == 0:
                    _miopen_mode._saved = torch.backends.cudnn.benchmark
                    torch.backends.cudnn.benchmark = False
                _miopen_mode._depth += 1

    def __exit__(self, *exc):
        if self.active:
            with _miopen_mode._lock:
                _miopen_mode._depth -= 1
                if _miopen_mode._depth == 0:
                    torch.backends.cudnn.benchmark = _miopen_mode._saved
        return False


class tangent_owner:
    """Context a curvature operator holds around its tangent sweep over the flat vector
    ``v``.  Conv layers whose ``v_W`` cotangent is a contiguous slice of ``v`` register
    their weight buffer with the operator on the first sweep; on every later sweep (in
    particular the one a hipGraph records) the operator fills all of them with ONE
    ``hf_unpack_tangent`` launch up front and the layers skip their own strided copy."""

    def __init__(self, operator, v):
        self.operator, self.v = operator, v

    def __enter__(self):
        op = self.operator
        if not hasattr(op, "_tangent_slots"):
            op._tangent_slots = {}  # id(ctx) -> (offset, wcat, cin)
        self.saved = (_Mode.owner, _Mode.vector, _Mode.prefilled, _Mode.producers)
        _Mode.owner, _Mode.vector = op, self.v
        _Mode.prefilled, _Mode.producers = False, {}
        if op._tangent_slots and self.v.is_cuda and self.v.is_contiguous():
            _lib.unpack_tangent(self.v, list(op._tangent_slots.values()))
            _Mode.prefilled = True
        return self

    def __exit__(self, *exc):
        _Mode.owner, _Mode.vector, _Mode.prefilled, _Mode.producers = self.saved
        return False


def _claim_direct(vgx, dst, cl):
    """First sweep of an operator: if ``vgx`` is the tangent output of a fused BatchNorm
    layer of this sweep, dense in the layout of the conv's operand buffer, let that layer
    write into ``dst`` (= ``xcat[:, :cin]``) from the next sweep on.  One claim per
    producer; its other consumers (a downsample conv, the residual branch) then see the
    slice, which they handle as a strided tensor."""
    if _Mode.owner is None:
        return
    entry = _Mode.producers.get(vgx.data_ptr())
    if entry is None or entry[1] is not vgx and entry[1].data_ptr() != vgx.data_ptr():
        return
    prod, t = entry
    if getattr(prod, "_hf_direct", None) is not None or t.shape != dst.shape or t.dtype != dst.dtype:
        return
    if _is_cl(t) != bool(cl) and not (t.is_contiguous() and t.is_contiguous(memory_format=torch.channels_last)):
        return
    if _slice_ld(dst, t) in (None, 0):
        return
    prod._hf_direct = dst


def _tangent_prefilled(ctx, vgw, wcat, cin):
    """True when the owner's up-front scatter already wrote ``vgw`` into ``wcat``;
    otherwise registers the slot (if ``vgw`` is a slice of the owner's vector)."""
    op, v = _Mode.owner, _Mode.vector
    if op is None or v is None or not vgw.is_contiguous() or vgw.dtype != v.dtype:
        return False
    esize = v.element_size()
    delta = vgw.data_ptr() - v.data_ptr()
    if delta < 0 or delta % esize or delta // esize + vgw.numel() > v.numel():
        return False
    off = delta // esize
    slot = op._tangent_slots.get(id(ctx))
    if _Mode.prefilled and slot is not None and slot[0] == off and slot[1] is wcat:
        return True
    op._tangent_slots[id(ctx)] = (off, wcat, cin)
    return False


def _point(x, w, padding, dilation, cl):
    """Index of the kernel centre if this NHWC convolution sees a 1x1 map through an odd
    square kernel with "same" padding -- then only the centre tap ever meets data and the
    layer is a GEMM on ``w[:, :, c, c]`` (unit stride over the input channels in NHWC:
    no copy); ``None`` otherwise.  The last stage of a ResNet on MNIST-sized inputs is
    made of such layers: MIOpen spends a zero-fill and an implicit-GEMM kernel of 7-16 us
    on each pass through them, the GEMM is one ~5 us kernel."""
    if not cl or x.dim() != 4 or x.shape[2] != 1 or x.shape[3] != 1:
        return None
    k = w.shape[2]
    if w.shape[3] != k or k % 2 == 0 or list(padding) != [k // 2] * 2 or list(dilation) != [1, 1]:
        return None
    return k // 2


def _point_backward(gy, x, w, c, need_gx, gw_out=None):
    """(gx, gw) of a ``_point`` layer: ``gx = gy W_c``; ``gw`` is zero off the centre tap
    and ``gy^T x`` on it (written into the persistent zero-initialised ``gw_out`` when
    given)."""
    gy2, x2 = gy.flatten(1), x.flatten(1)
    gx = (gy2 @ w[:, :, c, c]).view(x.shape) if need_gx else None
    if gw_out is None:
        gw_out = torch.zeros_like(w)
    torch.mm(gy2.t(), x2, out=gw_out[:, :, c, c])
    return gx, gw_out


def _bias_grad(gy):
    """Bias gradient ``sum_{n,hw} gy`` of a conv layer by the ``hf_chan_affine_bwd``
    reduction kernel.  Neither MIOpen's backward-bias routine nor ``gy.sum((0, 2, 3))``
    is used on this path: for NHWC cotangents of some shapes (All-CNN-C at batch 32: two
    of nine layers) both were right when issued eagerly and wrong -- drifting from
    replay to replay -- once captured in a hipGraph (scripts/experiments/nhwc_diag.py,
    scripts/experiments/bias_diag.sh); they zero a scratch buffer with a memset that does not
    survive the capture intact.  The kernel here has no scratch state."""
    if torch.is_grad_enabled() and gy.requires_grad:  # differentiable (Hessian products)
        return gy.sum(dim=(0, 2, 3))
    g = _dense(gy)
    n, c, hw = _dims(g)
    gb = torch.empty(c, dtype=g.dtype, device=g.device)
    _lib.check(
        _lib.load().hf_chan_affine_bwd(
            None, None, _p(gb), None, _p(g), None, None, None, None, None, None, n, c, hw,
            1 if _is_cl(g) else 0, _lib.dtype_code(g.dtype), _lib.current_stream_ptr(g.device)),
        "hf_chan_affine_bwd")
    return gb


def _own_conv_ok(x, w, cl, dilation, channels):
    """The package's implicit-GEMM kernels (``hf_conv2d_nhwc``: one launch, deterministic
    split-K, dead taps skipped) apply: NHWC fp32 on the GPU, unit dilation, both channel
    counts multiples of 4 (``HF_CONV=miopen`` keeps MIOpen everywhere)."""
    if not (cl and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and list(dilation) == [1, 1]
            and channels % 4 == 0 and w.shape[0] % 4 == 0 and w.shape[2] * w.shape[3] <= 64):
        return False
    # the kernels' 32-bit index limits (hf_conv.hip: check_common), with the output map bounded by
    # the input map (stride >= 1, "same"-or-smaller padding is the common case; a layer that still
    # exceeds them is refused by the library and falls back to MIOpen in the callers below)
    n, _, h, w_ = x.shape
    wide = max(channels, w.shape[0])
    return n * h * w_ * wide < 2**31 and w.numel() * (channels // max(1, w.shape[1])) < 2**31


def _use_own(mode, kind, rows):
    """Which implementation runs a convolution of the product whose GEMM view has ``rows``
    rows (N*OH*OW): the package's one-launch deterministic kernels or MIOpen.

    ``mode`` "own" / "miopen" force one of them (``prepare_model(deterministic=True)`` sets
    "own"; env ``HF_CONV`` overrides).  "auto" takes the faster one as measured on MI355X
    INSIDE the ResNet-18 product (bench.py --conv ..., profiles/r02_conv_modes.jsonl): MIOpen,
    for every layer and direction -- its hand-scheduled split-K kernels plus their zero-fill
    launch (~11 us per convolution) beat the own one-launch kernels (12-20 us), whose
    in-launch reduction is a chain of dependent memory round trips (partial tiles out,
    ticket, acquire, partial tiles in).  A stand-alone micro-benchmark of the same shapes
    (scripts/conv_kernel_bench.py) favours the own kernels for <= 128 rows; in the product
    that advantage is gone (operands arrive cold from other XCDs' L2).  kind: "T" tangent,
    "D" data gradient, "W" weight gradient, "DW" both in one launch, "stem" the tiny-Cin
    layer, "F" the once-per-step forward pass (own kernels: accuracy first)."""
    mode = os.environ.get("HF_CONV") or mode or "auto"
    if mode == "own":
        return True
    if mode == "miopen":
        return False
    return rows <= _auto_rows().get(kind, 0)


_AUTO_DEFAULT = {"T": 0, "D": 0, "W": 0, "DW": 0, "stem": 0, "F": 1 << 30}


def _auto_rows():
    """Row thresholds of the "auto" rule per kind; kind "stem" = the tiny-Cin layer (1 = own, 0 = MIOpen)."""
    return _AUTO_DEFAULT


def _tiny_cin(x, w, cl):
    """A layer with so few input channels that one kernel tap holds less than a 16-byte
    gather (the 1-channel 7x7 stem of the MNIST ResNet): its tangent and weight gradient run
    as 1x1 products over the im2col of the input, which is constant for a step."""
    return cl and x.is_cuda and x.dtype == torch.float32 and x.shape[1] < 4 and (
        x.shape[1] * w.shape[2] * w.shape[3] <= 256) and w.is_contiguous()


def _cols_of(ctx, x, w, stride, padding):
    if getattr(ctx, "cols", None) is None:
        u = torch.nn.functional.unfold(x.contiguous(), (w.shape[2], w.shape[3]), padding=tuple(padding),
                                       stride=tuple(stride))
        ctx.cols = u.transpose(1, 2).contiguous()  # [N, OH*OW, Cin*R*S], channel order (c, r, s) as w.flatten(1)
    return ctx.cols


def _out_hw(x, w, stride, padding):
    return ((x.shape[2] + 2 * padding[0] - w.shape[2]) // stride[0] + 1,
            (x.shape[3] + 2 * padding[1] - w.shape[3]) // stride[1] + 1)


def _own_conv_forward(xa, wa, stride, padding):
    """``conv2d(xa, wa)`` for NHWC ``xa`` [N, C, H, W] and ``wa`` [K, C, R, S] (both stored
    channels-last; C = the channel count both actually hold, e.g. 2*Cin for the tangent
    operands)."""
    n, c, h, w_ = xa.shape
    k, _, r, s = wa.shape
    oh, ow = _out_hw(xa, wa, stride, padding)
    out = torch.empty((n, k, oh, ow), dtype=xa.dtype, device=xa.device).contiguous(
        memory_format=torch.channels_last)
    return _lib.conv2d_nhwc(0, out, xa, wa, n, h, w_, c, k, r, s, stride, padding)


def _own_conv_dgrad(gy, x_shape, wT, r, s, stride, padding):
    """Data gradient: ``wT`` is the weight stored (I, H, W, O)."""
    n, c, h, w_ = x_shape
    k = gy.shape[1]
    gx = torch.empty((n, c, h, w_), dtype=gy.dtype, device=gy.device).contiguous(
        memory_format=torch.channels_last)
    return _lib.conv2d_nhwc(1, gx, gy, wT, n, h, w_, c, k, r, s, stride, padding)


def _own_conv_wgrad(gy, xa, w_like, stride, padding, out=None):
    """Weight gradient in the layout of ``w_like`` (channels-last).  Taps that never meet
    data are not written: ``out`` must then be a zero-initialised buffer."""
    n, c, h, w_ = xa.shape
    k, _, r, s = w_like.shape
    if out is None:
        out = torch.zeros_like(w_like)
    return _lib.conv2d_nhwc(2, out, xa, gy, n, h, w_, c, k, r, s, stride, padding)


class _ConvBwd(torch.autograd.Function):
    """(gy; x, w) -> (gx, gw, gb) of a convolution.  Recorded only in
    ``first_order_only`` mode, so the sole derivative ever taken is d/d gy, whose
    transpose is the layer's tangent map ``v_gy = conv(v_gx, w) + conv(x, v_gw) + v_gb``.
    PyTorch's generic double-backward evaluates the two convolutions separately
    (2 MIOpen calls, their layout transposes, 2 strided copies, 1 add); here they
    are ONE convolution over concatenated input channels, fed from two persistent
    buffers whose constant halves (x, w) are written once per step.

    With ``cl`` all operands are NHWC (channels_last): MIOpen's implicit-GEMM kernels
    then run without the NCHW<->NHWC transposes that make up ~40 % of the kernels of
    a product in NCHW (scripts/experiments/nhwc_probe.py: 261 -> 126 kernels, 1.08 -> 0.65 ms for
    the 20 layers of ResNet-18)."""

    @staticmethod
    def forward(ctx, gy, x, w, has_bias, stride, padding, dilation, cl, need_gx=True, mode=None):
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(x, w)
        ctx.conf = (stride, padding, dilation, has_bias, cl)
        ctx.mode = mode
        ctx.cat = None
        ctx.cols = None
        gy = _fmt(gy, cl)
        c = _point(x, w, padding, dilation, cl)
        if c is not None:
            gx, gw = _point_backward(gy, x, w, c, need_gx)
        else:
            with _miopen_mode(cl):
                gx, gw, _ = torch.ops.aten.convolution_backward(
                    gy, x, w, None, stride, padding, dilation, False, [0] * len(stride), 1,
                    [need_gx, True, False])
        return gx, gw, _bias_grad(gy) if has_bias else None

    @staticmethod
    def backward(ctx, vgx, vgw, vgb):
        x, w = ctx.saved_tensors
        stride, padding, dilation, _, cl = ctx.conf
        mode = ctx.mode
        c = _point(x, w, padding, dilation, cl)

        oh, ow = _out_hw(x, w, stride, padding)
        rows = x.shape[0] * oh * ow

        def conv(xa, wa, *rest):
            if _own_conv_ok(xa, wa, cl, dilation, xa.shape[1]) and _use_own(mode, "T", rows):
                return _own_conv_forward(xa, wa, stride, padding)
            if c is not None:  # a GEMM on the centre tap
                return (xa.flatten(1) @ wa[:, :, c, c].t()).view(xa.shape[0], wa.shape[0], 1, 1)
            with _miopen_mode(cl):
                return torch.nn.functional.conv2d(xa, wa, *rest)

        if vgx is None and vgw is None:
            v_gy = None
        elif vgx is None and _tiny_cin(x, w, cl) and vgw.is_contiguous() and _use_own(mode, "stem", 1):
            # 1x1 product over the step's im2col: v_gy[m, k] = sum_j cols[m, j] * v_W[k, j]
            cols = _cols_of(ctx, x, w, stride, padding)
            n_, k_, j_ = x.shape[0], w.shape[0], cols.shape[2]
            v_gy = torch.empty((n_, k_, oh, ow), dtype=x.dtype, device=x.device).contiguous(
                memory_format=torch.channels_last)
            _lib.conv2d_nhwc(0, v_gy, cols, vgw, n_ * oh * ow, 1, 1, j_, k_, 1, 1, (1, 1), (0, 0))
        elif vgx is None:
            v_gy = conv(x, _fmt(vgw, cl), None, stride, padding, dilation)
        elif vgw is None:
            v_gy = conv(_fmt(vgx, cl), w, None, stride, padding, dilation)
        else:
            cin = x.shape[1]
            if ctx.cat is None:
                fmt = torch.channels_last if cl else torch.contiguous_format
                xcat = torch.empty((x.shape[0], 2 * cin) + tuple(x.shape[2:]), dtype=x.dtype,
                                   device=x.device).contiguous(memory_format=fmt)
                wcat = torch.empty((w.shape[0], 2 * cin) + tuple(w.shape[2:]), dtype=w.dtype,
                                   device=w.device).contiguous(memory_format=fmt)
                xcat[:, cin:].copy_(x)
                wcat[:, :cin].copy_(w)
                ctx.cat = (xcat, wcat)
            xcat, wcat = ctx.cat
            dst = xcat[:, :cin]
            if not (vgx.data_ptr() == dst.data_ptr() and vgx.shape == dst.shape
                    and vgx.stride() == dst.stride()):  # else: the producer wrote it in place
                dst.copy_(vgx)
                _claim_direct(vgx, dst, cl)
            if not _tangent_prefilled(ctx, vgw, wcat, cin):
                wcat[:, cin:].copy_(vgw)
            v_gy = conv(xcat, wcat, None, stride, padding, dilation)
        if vgb is not None:
            vb = vgb.view(1, -1, 1, 1)
            if v_gy is None:  # only the bias carries a tangent: constant over the output map
                v_gy = vb.expand(x.shape[0], -1, *_out_hw(x, w, stride, padding))
            else:
                v_gy = v_gy + vb
        return v_gy, None, None, None, None, None, None, None, None, None


def _own_backward(ctx, gy, xf, wf, stride, padding, own_dw, own_d):
    """(gx, gw) of a conv layer on the package's deterministic one-launch kernels (hf_conv2d_nhwc);
    raises ``RuntimeError`` when the library refuses the geometry."""
    gy = _fmt(gy, True)
    r, s_ = wf.shape[2], wf.shape[3]
    # taps that only ever meet padding are never written: a zero-filled buffer,
    # persistent while the results are consumed at once (the product's gather),
    # fresh otherwise (a caller may retain several gradients of one graph)
    if _Mode.consume_now:
        if ctx.gw_buf is None:
            ctx.gw_buf = torch.zeros_like(wf)
        gw = ctx.gw_buf
    else:
        gw = torch.zeros_like(wf)
    if (own_dw or own_d) and ctx.wT is None:  # weight stored (I, H, W, O), once per step
        ctx.wT = wf.permute(1, 2, 3, 0).contiguous()
    gx = None
    if own_dw:
        n_, c_, h_, w_ = xf.shape
        gx = torch.empty_like(xf)  # channels_last like xf
        _lib.conv2d_nhwc_backward(gx, gw, gy, xf, ctx.wT, n_, h_, w_, c_, wf.shape[0], r, s_,
                                  stride, padding)  # data + weight gradient: ONE launch
    else:
        _own_conv_wgrad(gy, xf, wf, stride, padding, out=gw)
        if own_d:
            gx = _own_conv_dgrad(gy, xf.shape, ctx.wT, r, s_, stride, padding)
    return gx, gw


class _Conv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, stride, padding, dilation, cl, mode=None):
        xf, wf = _fmt(x, cl), _fmt(w, cl)
        ctx.save_for_backward(x, w, xf, wf)
        ctx.conf = (stride, padding, dilation, b is not None, cl)
        ctx.mode = mode
        ctx.cols = None
        ctx.gw_buf = None
        ctx.wT = None
        # The forward pass of an NHWC layer runs on the package's own kernels by default ("F" in
        # the auto rule): once per step its speed does not matter, its accuracy does -- the
        # activations feed every curvature product -- and for shapes without a find-db record
        # MIOpen's heuristic solver choice was measured to be good to only 1e-4..1e-3 on some
        # machines (a fresh box with a cold kernel cache), run to run.
        oh, ow = _out_hw(xf, wf, stride, padding)
        rows = xf.shape[0] * oh * ow
        # (a geometry the library refuses -- HF_ERR_ARG / HF_ERR_CAPACITY: index or scratch limits of
        # very large maps -- runs on MIOpen like any layer the own kernels do not cover)
        if _own_conv_ok(xf, wf, cl, dilation, xf.shape[1]) and _use_own(mode, "F", rows):
            try:
                y = _own_conv_forward(xf, wf, stride, padding)
                return y if b is None else y + b.view(1, -1, 1, 1)
            except _lib.Refused:  # (a size / alignment the library does not cover; real faults propagate)
                pass
        elif _tiny_cin(xf, w, cl) and _use_own(mode, "F", rows):
            try:
                cols = _cols_of(ctx, xf, w, stride, padding)  # 1x1 product over the im2col, (c, r, s) order
                n_, k_, j_ = xf.shape[0], w.shape[0], cols.shape[2]
                y = torch.empty((n_, k_, oh, ow), dtype=xf.dtype, device=xf.device).contiguous(
                    memory_format=torch.channels_last)
                _lib.conv2d_nhwc(0, y, cols, w, n_ * oh * ow, 1, 1, j_, k_, 1, 1, (1, 1), (0, 0))
                return y if b is None else y + b.view(1, -1, 1, 1)
            except _lib.Refused:
                ctx.cols = None
        c = _point(xf, wf, padding, dilation, cl)
        if c is not None:
            y = xf.flatten(1) @ wf[:, :, c, c].t()
            return (y if b is None else y + b).view(xf.shape[0], wf.shape[0], 1, 1)
        with _miopen_mode(cl):
            return torch.nn.functional.conv2d(xf, wf, b, stride, padding, dilation)

    @staticmethod
    def backward(ctx, gy):
        x, w, xf, wf = ctx.saved_tensors
        stride, padding, dilation, has_bias, cl = ctx.conf
        need_gx = ctx.needs_input_grad[0]  # False for the first layer of a net
        if _Mode.first_order_only:
            gx, gw, gb = _ConvBwd.apply(gy, xf.detach(), wf.detach(), has_bias, stride, padding,
                                        dilation, cl, need_gx, ctx.mode)
        else:
            # stock call.  Differentiable again (Hessian products, create_graph) when grad
            # mode is on, so then it must see the tracked x, w; a plain first-order sweep
            # (the adjoint pass of every GGN product) uses the layout-converted copies --
            # with the NCHW parameter PyTorch would re-convert the weight on every call
            c = None
            own = tiny = False
            mode = ctx.mode
            if torch.is_grad_enabled():
                xa, wa = x, w
            else:
                xa, wa = xf, wf
                own = _own_conv_ok(xf, wf, cl, dilation, xf.shape[1])
                tiny = (not need_gx) and _tiny_cin(xf, wf, cl) and _use_own(mode, "stem", 1)
                c = _point(xf, wf, padding, dilation, cl)
            oh, ow = _out_hw(xf, wf, stride, padding)
            rows = xf.shape[0] * oh * ow
            own_dw = own and need_gx and _use_own(mode, "DW", rows)
            own_w = own and not own_dw and _use_own(mode, "W", rows)
            own_d = own_w and need_gx and _use_own(mode, "D", rows)
            gx = gw = None
            if own_dw or own_w:
                try:
                    gx, gw = _own_backward(ctx, gy, xf, wf, stride, padding, own_dw, own_d)
                except _lib.Refused:  # refused by the library (size limits): MIOpen below
                    gx = gw = None
            elif tiny:
                # weight gradient of a tiny-Cin layer: gW[k, j] = sum_m gy[m, k] * cols[m, j]
                cols = _cols_of(ctx, xf, wf, stride, padding)
                gy = _fmt(gy, True)
                gw = torch.empty_like(wf)
                _lib.conv2d_nhwc(2, gw, cols, gy, cols.shape[0] * cols.shape[1], 1, 1, cols.shape[2],
                                 wf.shape[0], 1, 1, (1, 1), (0, 0))
            elif c is not None:
                if _Mode.consume_now:
                    if ctx.gw_buf is None:  # zero off the centre tap, for good
                        ctx.gw_buf = torch.zeros_like(wf)
                    gx, gw = _point_backward(gy, xf, wf, c, need_gx, ctx.gw_buf)
                else:
                    gx, gw = _point_backward(gy, xf, wf, c, need_gx)
            if gw is None or (need_gx and gx is None):
                with _miopen_mode(cl):
                    g2 = torch.ops.aten.convolution_backward(
                        gy, xa, wa, None, stride, padding, dilation, False, [0] * len(stride), 1,
                        [need_gx and gx is None, gw is None, False])
                gx = g2[0] if (need_gx and gx is None) else gx
                gw = g2[1] if gw is None else gw
            gb = _bias_grad(gy) if has_bias else None
        return gx, gw, gb if has_bias else None, None, None, None, None, None


def _conv_forward(self, x):
    usable = (
        x.is_cuda and self.groups == 1 and self.padding_mode == "zeros" and x.dim() == 4
        and not isinstance(self.padding, str)
    )
    if not usable:
        return self._hf_stock_forward(x)
    y = _Conv.apply(x, self.weight, self.bias, list(self.stride), list(self.padding),
                    list(self.dilation), bool(getattr(self, "_hf_channels_last", False)),
                    getattr(self, "_hf_conv_mode", None))
    # what the curvature engine (engine.py) needs of this step's forward; detached: a record
    # must not keep an autograd graph (and its stream affinity) alive
    self._hf_io = (x.detach(), y.detach())
    return y


def fuse_conv_tangent(model, channels_last=False, conv_mode=None):
    """Patch every plain ``nn.Conv2d`` (groups=1, zero padding) so that, inside a GGN
    product, its tangent map is one convolution (see ``_ConvBwd``).  Forward and
    first-order backward are the stock MIOpen calls.  With ``channels_last`` the
    layer computes in NHWC (activations and a per-step NHWC copy of the weight):
    combined with the NHWC-capable fused BatchNorm kernels a whole conv net then
    runs its curvature passes without layout transposes.  Parameters, their order
    and the flat-vector layout are unchanged.  Returns the number patched."""
    count = 0
    for m in model.modules():
        if type(m) is nn.Conv2d:
            m._hf_channels_last = bool(channels_last)
            m._hf_conv_mode = conv_mode
            if not hasattr(m, "_hf_stock_forward"):
                m._hf_stock_forward = m.forward
                m.forward = types.MethodType(_conv_forward, m)
                count += 1
    return count


def _pool_forward(self, x):
    if x.dim() >= 3 and tuple(x.shape[-2:]) == (1, 1):
        return x  # the mean over a single pixel: skip the kernel (and its two in every product)
    return self._hf_stock_forward(x)


def skip_identity_pools(model):
    """``AdaptiveAvgPool2d(1)`` applied to a map that is already 1x1 (ResNets on MNIST-sized
    inputs) returns its input: patch the module so that it issues no kernel then.  Exact.
    Returns the number of modules patched."""
    count = 0
    for m in model.modules():
        if type(m) is nn.AdaptiveAvgPool2d and m.output_size in (1, (1, 1)) and not hasattr(m, "_hf_stock_forward"):
            m._hf_stock_forward = m.forward
            m.forward = types.MethodType(_pool_forward, m)
            count += 1
    return count


def release_records(model):
    """Drop the activation records of the last forward pass (``_hf_io`` of every layer): the engine has copied
    what it needs, and a record keeps its activation alive until the next forward pass."""
    for m in model.modules():
        if getattr(m, "_hf_io", None) is not None:
            m._hf_io = None


def _record_io(module, inputs, output):
    x = inputs[0] if inputs else None
    module._hf_io = (x.detach() if isinstance(x, torch.Tensor) else x,
                     output.detach() if isinstance(output, torch.Tensor) else output)


def _tag_output(module, inputs, output):
    """Top-level forward hook: the network output remembers which prepared model produced it
    from which input, so that ``curvature.ggn_operator`` can hand the product to the fused
    curvature engine (engine.py) when the model is of a family it knows."""
    if isinstance(output, torch.Tensor):
        import weakref

        output._hf_model = weakref.ref(module)
        output._hf_input = inputs[0] if inputs else None


class session_forward:
    """Context ``HessianFree.step`` holds around the caller's ``forward()`` once a persistent engine
    session exists for the model: the model's forward pass is then ONE graph replay on the
    session's static buffers (``EngineSession.forward_override``) instead of ~60 Python-level layer
    calls, and returns the logits as a fresh leaf -- the caller's loss function still runs under
    autograd on it, which is all the step needs (the gradient is the engine's own sweep)."""

    def __init__(self, session):
        self.session = session

    def __enter__(self):
        self._old = _Mode.session
        _Mode.session = self.session

    def __exit__(self, *exc):
        _Mode.session = self._old
        return False


def _model_forward(self, *args, **kwargs):
    sess = _Mode.session
    if sess is not None and len(args) == 1 and not kwargs:
        out = sess.forward_override(self, args[0])
        if out is not None:
            return out
    return self._hf_stock_model_forward(*args, **kwargs)


def _install_engine_hooks(model):
    if getattr(model, "_hf_engine_hooks", False):
        return
    for m in model.modules():
        if isinstance(m, (nn.MaxPool2d, nn.AdaptiveAvgPool2d, nn.Linear, nn.Flatten, nn.ReLU)):
            m.register_forward_hook(_record_io)
    model.register_forward_hook(_tag_output)
    model._hf_stock_model_forward = model.forward
    model.forward = types.MethodType(_model_forward, model)
    model._hf_engine_hooks = True


def prepare_model(model, channels_last=False, deterministic=False):
    """All opt-in preparations; returns ``model`` for chaining.

    ``deterministic=True`` (with ``channels_last=True``): every convolution of the curvature
    product that the package's own kernels can run (fp32 NHWC, unit dilation) does run on
    them -- one launch each, split-K partial sums combined in a fixed order -- instead of
    MIOpen's split-K kernels with atomic accumulation.  Two products of the same vector are
    then bitwise equal (the reference's ``_test_mvp_deterministic``, optimizer.py:414-448,
    passes exactly); it costs ~20 % of the product's speed on the ResNet-18 workload.  The
    default ("auto") uses the own kernels only where they are also the faster ones.

    ``channels_last=True`` additionally runs the convolution layers, and with them the
    fused BatchNorm kernels, in NHWC: MIOpen's implicit-GEMM kernels then need no layout
    transposes (~40 % of the kernels of a product in NCHW; ResNet-18 bench 668 -> 925
    matvecs/s).  Parameters, their order and the flat-vector layout do not change:
    layers keep a per-step NHWC copy of their weight and ``hf_pack`` un-permutes the
    NHWC weight gradients while it gathers them.  NHWC convolutions run in MIOpen's
    immediate mode (``_miopen_mode``), i.e. at full speed for shapes with a find-db
    record (shipped for the BASELINE.json ResNet-18 workload; ``HF_NHWC_FIND=1`` once to
    add yours) and on MIOpen's heuristic choice otherwise -- which is why NCHW, where
    the find step has been reliable, stays the default.

    History: the NHWC path used to produce, in about one process of ten, a product that
    was off by 1e-3 or plain garbage.  Causes found (scripts/experiments/nhwc_diag.py): MIOpen's
    composable-kernel split-K weight gradient, now disabled package-wide (``__init__``),
    and reductions that zero a scratch buffer before accumulating (MIOpen's backward-bias,
    PyTorch's multi-block ``sum`` of an NHWC tensor), which break inside a hipGraph and
    are replaced by ``_bias_grad``.  ``GraphedOperator`` now checks its first replay
    against the eager product."""
    from .config import configure

    configure()  # MIOpen settings the patched layers were validated with (config.py)
    fuse_eval_batchnorm(model)
    fuse_conv_tangent(model, channels_last=channels_last, conv_mode="own" if deterministic else None)
    fuse_residual_blocks(model)
    fuse_bn_relu(model)
    skip_identity_pools(model)
    _install_engine_hooks(model)
    return model


def fuse_eval_batchnorm(model):
    """Patch all BatchNorm layers of ``model`` (see module docstring); returns the
    number of layers patched."""
    count = 0
    for m in model.modules():
        if isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d)) and not hasattr(
            m, "_hf_stock_forward"
        ):
            m._hf_stock_forward = m.forward
            m.forward = types.MethodType(_fused_forward, m)
            count += 1
    return count
