"""Fused curvature engine: the GGN product ``v -> J^T H_L J v`` of a prepared ResNet-family
model (conv - eval-BatchNorm - ReLU units with residual connections, NHWC fp32) by EXPLICIT
tangent and adjoint sweeps over its layers, issued as direct kernel launches.

Why.  The reference obtains the product from BackPACK's R-op / L-op (optimizer.py:457-462), i.e.
from autograd; so does ``curvature.GGNOperator``.  On an MI355X that product is bound by the
NUMBER of dependent launches (~170 x ~6 us for ResNet-18 on 28x28 inputs), and the convolutions
inside it are MIOpen split-K kernels: a zero-fill launch + a kernel that accumulates with
atomics (not repeatable).  Here every convolution is ONE launch of the package's implicit-GEMM
kernels (hf_conv.hip) whose split-K partial results ("slabs") are summed by the kernel that
CONSUMES them, in its prologue -- the launch boundary publishes them, there is no zero-fill, no
atomic, no in-launch reduction:

    tangent sweep, per unit :  T-conv([t_x | x], [W | v_W]) -> slabs
                               BatchNorm tangent (+ residual tangent, ReLU mask) sums the slabs and
                               writes straight into the next unit's [t_x | x] operand
    adjoint sweep, per unit :  BatchNorm adjoint: g = mask * (sum of the consumers' cotangent slabs),
                               g_a = g * w * rstd, per-channel sums      (hf_chan_affine_bwd_ex)
                               data + weight gradient of the convolution in ONE launch -> slabs
    once per product        :  hf_unpack_tangent_ex (v_W of all layers), hf_maxpool_tangent_nhwc,
                               hf_linear_ce_head (logits' tangent, loss Hessian, the head's three
                               gradients), hf_maxpool_adjoint_nhwc, hf_pack_ex (all parameter
                               gradients, summing the weight-gradient slabs while it gathers);
                               slices of kernel taps that never meet data are skipped throughout

4 launches per conv-BN unit instead of 8 -- 2 where a block's first convolution and its downsample
branch share their launches (grouped kernels) --, 74 per product of ResNet-18; bitwise repeatable.  Under
data parallelism only the entries of the product that can be non-zero are all-reduced (``reduce``).
The layer topology is taken from the prepared model's module tree and from the activations its
patched layers recorded during the step's forward pass (``modelprep`` stores them detached);
anything the engine does not recognise makes ``try_build`` return ``None`` and the caller uses
the autograd operator.  The first product of every model signature is compared with that
operator's product on a random vector.
"""

import os
import warnings

import torch

from .. import _lib
from ..curvature import GGNOperator, _Operator
from .common import _Node, _Unsupported, _ce_node, ce_loss_spec, loss_spec_of  # noqa: F401
from .adjoint import _AdjointSweep
from .buffers import _Buffers
from .dataparallel import _DataParallel
from .diag_ef import _DiagEF
from .forward import _Forward
from .hessian import _HessianExtras
from .tangent import _TangentSweep
from .topology import _Topology


class FusedGGNEngine(_Topology, _Buffers, _Forward, _TangentSweep, _AdjointSweep, _HessianExtras, _DiagEF, _DataParallel,
                     _Operator):
    mode = ("fused curvature engine: own deterministic convolutions (split-K slabs summed by the consumer "
            "kernel), BatchNorm tangents/adjoints fused, 4 launches per conv-BN unit, downsample branches grouped "
            "with their block's first convolution")

    # ------------------------------------------------------------------------------------
    @classmethod
    def try_build(cls, loss, outputs, params, weight=1.0, group=None, hessian=False, why=None):
        """The engine for the model that produced ``outputs``, or ``None``; ``why`` (a list) receives one line per
        reason it was not taken -- what ``HessianFree.path_report()`` and its one-time warning quote."""
        why = [] if why is None else why
        if os.environ.get("HF_ENGINE", "1") == "0":
            why.append("the fused engine is switched off (HF_ENGINE=0)")
            return None
        ref = getattr(outputs, "_hf_model", None)
        model = ref() if ref is not None else None
        if model is None:
            why.append("the model is not a prepared one (modelprep.prepare_model(model, channels_last=True) installs "
                       "the layers the fused engine reads)")
            return None
        if not outputs.is_cuda or outputs.dtype != torch.float32 or outputs.dim() != 2:
            why.append(f"the model output is not a CUDA float32 [batch, classes] tensor (got {outputs.dtype}, "
                       f"{tuple(outputs.shape)}, {outputs.device})")
            return None
        from .plain import PlainStackEngine

        kinds = [cls] if cls is not FusedGGNEngine else [FusedGGNEngine, PlainStackEngine]
        if hessian:
            kinds = [k for k in kinds if k.supports_hessian]
        for kind in kinds:
            try:
                return kind(model, loss, outputs, params, weight, group, hessian=hessian)
            except _Unsupported as exc:
                # (a model the engine does not cover is the normal case: quiet unless asked;
                # a product that FAILED its check is always reported)
                why.append(f"{kind.__name__}: {exc}")
                if os.environ.get("HF_ENGINE_DEBUG") or getattr(exc, "loud", False):
                    warnings.warn(f"fused curvature engine ({kind.__name__}) not used: {exc}")
                if getattr(exc, "loud", False):
                    return None
            except _lib.Refused as exc:  # a kernel refused its arguments (alignment, size limits ...)
                why.append(f"{kind.__name__}: a kernel refused its arguments: {exc}")
                warnings.warn(f"fused curvature engine ({kind.__name__}) not used: {exc}")
                return None
        return None

    # Hessian products (optimizer.py:450-455) by forward-over-reverse on the same kernels; residual nets with
    # EVAL-mode BatchNorm (the layer is a per-channel affine map; ReLU masks and max-pool positions are
    # piecewise constant): per unit, besides the GGN's terms, conv_D(g, V) and conv_W(t_x, g) (g: the step's
    # first-order cotangent, kept by the gradient sweep) as MORE SLABS of the same buffers, and the BatchNorm
    # scale's own second-order terms  g_a' += g_z * rstd * v_gamma ,  g_gamma' += sum g_z * rstd * t_a .
    # TRAIN-mode BatchNorm (round 5): the batch statistics' share of the adjoint's tangent in closed form per channel
    # (hf_bn_train_hessian_coeffs / _apply; the formulas stand in csrc/hf_bn.hip and DESIGN.md section 4.3).
    supports_hessian = True

    def __init__(self, model, loss, outputs, params, weight, group, hessian=False):
        super().__init__(params, weight, group)
        self.hessian = bool(hessian)
        self._second = False  # (inside the adjoint sweep of a Hessian product)
        self._l2 = None
        self.outputs = outputs
        self.dev = outputs.device
        self._index = {id(p): i for i, p in enumerate(self.params)}
        offs, o = [], 0
        for p in self.params:
            offs.append(o)
            o += p.numel()
        self._offs = offs
        self.train_bn = False
        self.frozen_any, self.dead_blocks = False, 0
        self._layout(model)
        if self.hessian:
            for u in self.units:
                u.needs_g = True  # (the first-order masked cotangent of every unit is kept)
        self._allocate()
        self.set_batch(getattr(outputs, "_hf_input").detach(), None)
        self.refresh_weights(transposed=True)
        self.train_own = False
        if self.train_bn:
            # train-mode BatchNorm: the engine linearises at the activations and batch statistics the MODEL's
            # forward pass recorded.  Its own forward pass (batch statistics by own kernels, running statistics
            # moved as the layers' forward moves them) lets a persistent session serve such a model too -- if it
            # reproduces the model's output here (with the running statistics left alone)
            if (all(u.bn.momentum is not None and u.bn.track_running_stats for u in self.units if u.train)):
                self.forward_own(update_running=False)
                want = outputs.detach()
                err = float((self.logits - want).abs().max() / want.abs().max().clamp_min(1e-30))
                self.train_own = err < 1e-4
            self._load_recorded(outputs)
        else:
            # own forward pass on the engine's static buffers; it must reproduce the model's output
            self.forward_own()
            want = outputs.detach()
            err = float((self.logits - want).abs().max() / want.abs().max().clamp_min(1e-30))
            if not err < 1e-4:
                raise _Unsupported(f"the engine's forward pass differs from the model's output by {err:.2e}")
        self._loss_setup(loss, outputs)
        self._verify(loss)
        # ONE linearisation point after construction, whether or not the first-use check ran (it is skipped for a
        # model signature that has passed before): the engine's OWN forward pass where it has one, else -- a
        # train-mode model whose layers the own pass does not reproduce -- everything the model recorded, batch
        # statistics included.  (Round 4 left a train-mode engine whose check was skipped at the recorded
        # activations with its own statistics: 3.65e-6 / one ReLU decision away from the checked one, GPUTEST_r04.)
        if self._at != "own" and (not self.train_bn or self.train_own):
            self.forward_own(update_running=False)  # (eval mode: nothing to move)
            if self.hessian:
                self.gradient()  # the first-order cotangents the Hessian products read, at the same point
        for u in self.units:  # the model's own activations were only needed up to here
            u.rx = u.ry = u.ra = None
            u.rec_stats = None
        self._rec_pool = None
        if self.loss_spec is not None:
            self.outputs = None  # nothing of the step's autograd graph stays alive in the engine
            from ..modelprep import release_records

            release_records(model)  # (the layers' records pinned this pass's activations until the next one)

    # ---- loss Hessian (same contract as GGNOperator) -------------------------------------
    def _loss_setup(self, loss, outputs):
        (self._dl,) = torch.autograd.grad(loss, outputs, create_graph=True, retain_graph=True)
        self._ce = GGNOperator._closed_form_loss_hessian(self, _Node(_ce_node(loss)), outputs)
        # a plain softmax cross-entropy (checked numerically above): the engine can then evaluate
        # loss, probabilities and d loss / d logits itself, on its own forward pass -- which is what
        # lets ONE engine serve many steps and trial points (``session.EngineSession``)
        # ... or a mean-squared error (round 6; the loss of the reference's own examples / tests): gradient
        # 2c (out - t), Hessian 2c I
        self.loss_spec = None
        if not self.train_bn or self.train_own:
            spec = loss_spec_of(loss, outputs)
            if spec is not None and (spec["kind"] == "mse" or self._ce is not None):
                self.loss_spec = spec
                self._set_quadratic(spec.get("quadratic"))
                self.set_targets(spec["targets"])
                if spec["kind"] == "mse":
                    self._ce = None
                    self._mse2 = 2.0 / float(outputs.numel()) if spec["reduction"] == "mean" else 2.0
                self._loss_head()
                if spec["kind"] == "ce":
                    self._ce = (self._p, self._ce[1])  # the static buffer the own forward pass refreshes
                self._dl = None                        # (nothing of this step's autograd graph is kept)
        if self.hessian:
            if self.loss_spec is None:
                raise _Unsupported("Hessian products on the engine need a plain softmax cross-entropy or mean-squared-"
                                   "error loss")
            self.gradient()  # fills the first-order cotangents the Hessian products read

    def _set_quadratic(self, terms):
        """``loss = cross-entropy + sum_j 0.5 * coef_j * ||w_j||^2`` (the L2 term of the reference's
        All-CNN-C example, examples/example_utils.py:77-81): per-entry coefficients of the flat vector.
        Loss value, gradient and Hessian product of that term are ``0.5 <d*theta, theta>``,
        ``d*theta`` and ``d*v``."""
        self._l2 = None
        if not terms:
            return
        d = torch.zeros(self.n, dtype=torch.float32, device=self.dev)
        for coef, tensors in terms:
            for w in tensors:
                i = self._index.get(id(w))
                if i is None:
                    # (a frozen tensor would add a constant to the loss value the engine's own loss head reports)
                    raise _Unsupported("a regularised tensor is not among the optimizer's parameters")
                d[self._offs[i]: self._offs[i] + w.numel()] += float(coef)
        self._l2 = d

    def _theta(self):
        flat = self._flat_params
        if flat is not None and flat.data_ptr() == self.params[0].data_ptr():
            return flat
        return torch.cat([p.detach().reshape(-1) for p in self.params])

    def gradient(self, out=None):
        """``weight * d loss / d params`` of the softmax cross-entropy by ONE adjoint sweep of the
        engine (the gradient the reference takes with ``torch.autograd.grad``, optimizer.py:231-234),
        on the activations of the last ``forward_own``."""
        if self.loss_spec is None:
            raise RuntimeError("engine.gradient needs a softmax cross-entropy loss")
        if out is None:
            out = torch.empty(self.n, dtype=torch.float32, device=self.dev)
        g = self._dlogits()  # d loss / d logits
        fw = self.fc.weight.detach()
        g_fw = g.t() @ self.feat
        g_fb = g.sum(0) if self.pfb is not None else None
        g_feat = g @ fw
        if self.hessian:
            # the first-order cotangents stay for the step's Hessian products: the sweep below writes the units'
            # g1 / ga1 instead of g / ga (buffers swapped for its duration)
            if getattr(self, "_gl1", None) is None:
                self._gl1 = torch.empty_like(g)
            self._gl1.copy_(g)
            self._swap_first_order()
        try:
            pool_srcs = self._adjoint_blocks(self._feature_cotangent(g_feat))
            self._adjoint_stem(pool_srcs)
        finally:
            if self.hessian:
                self._swap_first_order()
        self._gather(out, g_fw, g_fb, first_order=True)
        if self.hessian and self.train_bn:
            for u in self.units:  # the layer's own first-order parameter gradients: hf_bn_train_hessian_coeffs reads them
                if u.train:
                    torch.sum(u.gw[:u.rb], 0, out=u.gg1)
                    torch.sum(u.gb, 0, out=u.gb1)
        if self._l2 is not None:
            out.addcmul_(self._l2, self._theta(), value=self.weight)
        return out

    def _dlogits(self):
        """``d loss / d logits`` of the engine's own loss head at the current logits."""
        if self.loss_spec["kind"] == "mse":
            return (self.logits - self._targets) * self._mse2
        return (self._p - self._onehot) * self._ce[1]

    def _loss_hessian(self, Jv):
        if self.loss_spec is not None and self.loss_spec["kind"] == "mse":
            return Jv * self._mse2
        return GGNOperator._loss_hessian(self, Jv)

    # ---- the product -------------------------------------------------------------------------
    def local(self, v, out=None):
        if out is None:
            out = torch.empty(self.n, dtype=torch.float32, device=self.dev)
        v = v.detach()
        if not v.is_contiguous():
            v = v.contiguous()
        self._tangent_stem(v, carry_scatter=True)  # + the v_W halves of all [W | v_W] operands, same launch
        self._tangent_blocks(v)
        if self.hessian:
            if self._vt_slots:
                _lib.unpack_tangent(v, self._vt_slots, half=2)  # V as (I, H, W, O): the operand of conv_D(g, V)
            self._second, self._v = True, v
            self._extras_fork()
        try:
            g_last, g_fw, g_fb = self._head(v)
            pool_srcs = self._adjoint_blocks(g_last)
            self._adjoint_stem(pool_srcs)
            if self.hessian:
                self._extras_join()
        finally:
            self._second, self._v = False, None
        self._gather(out, g_fw, g_fb)
        if self.hessian and self._l2 is not None:  # the regulariser's Hessian: coef on its tensors' entries
            out.addcmul_(self._l2, v, value=self.weight)
        return out

    def __call__(self, v, out=None):
        self.calls += 1
        return self.reduce(self.local(v, out))


    # ---- safety net --------------------------------------------------------------------------
    # relative max-norm distance to the autograd product above which the engine is refused; fp32
    # products of the shipped workloads agree to ~1e-6.  Deep, badly conditioned nets (the random-init
    # ResNet-50) scatter more for EVERY fp32 implementation: callers that have measured what stock
    # fp32 autograd achieves against float64 (bench.py) raise it to max(1e-5, 5 x that error)
    verify_tol = 1e-5

    def _verify(self, loss):
        """First product of every (model, shape) signature against the autograd operator, both on the
        activations the model's own forward pass recorded (``_load_recorded``)."""
        policy = os.environ.get("HF_ENGINE_VERIFY", "first")
        key = ("hessian" if self.hessian else "ggn", self.train_bn,
               tuple(type(m).__name__ for m in self.model_ref.modules()), self.n,
               tuple(tuple(p.shape) for p in self.params), tuple(self.logits.shape), tuple(self.x_in.shape),
               str(self.dev))
        # (kept ON the model: a registry keyed by id(model) outlives the model, and CPython hands the address of a
        # collected model to the next one)
        verified = self.model_ref.__dict__.setdefault("_hf_engine_verified", set())
        if policy == "never" or (policy != "always" and key in verified):
            return
        gen = torch.Generator(device=self.dev).manual_seed(4321)
        v = torch.randn(self.n, device=self.dev, generator=gen)
        weight, self.weight = self.weight, 1.0
        self._load_recorded(self.outputs)
        try:
            if self.hessian:
                self.gradient()  # first-order cotangents at the recorded activations
            got = self.local(v).clone()
        finally:
            self.weight = weight  # (the constructor moves the engine to its final linearisation point afterwards)
        if self.hessian:
            from ..curvature import HessianOperator

            want = HessianOperator(loss, self.params).local(v)
        else:
            want = GGNOperator(loss, self.outputs, self.params).local(v)
        err = float((got - want).abs().max() / want.abs().max().clamp_min(1e-30))
        # (Hessian products through batch statistics: stock fp32 double backward is itself 6e-6 from float64 on the
        # train-mode ResNet-18, this engine 3e-6 ... 7e-6 -- two fp32 results may be the sum of both apart)
        tol = FusedGGNEngine.verify_tol * (3.0 if (self.hessian and self.train_bn) else 1.0)
        if not err < tol:
            exc = _Unsupported(f"engine {'Hessian ' if self.hessian else ''}product differs from the autograd product by {err:.2e} "
                               f"(tolerance {tol:.1e}); using the autograd operator")
            exc.loud = True
            raise exc
        verified.add(key)
