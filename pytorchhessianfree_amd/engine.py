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
from torch import nn

from . import _lib
from .curvature import GGNOperator, _Operator, _all_reduce_sum

_P = _lib.c_void_p


def _ptr(t):
    return _P(t.data_ptr()) if t is not None else None


def _same(a, b):
    """Two records refer to the same activation (records are detached: compare storage)."""
    return a is not None and b is not None and a.data_ptr() == b.data_ptr() and a.shape == b.shape


def _cl(t):
    return t.contiguous(memory_format=torch.channels_last)


def _cu_count(dev):
    return torch.cuda.get_device_properties(dev).multi_processor_count


class _Unit:
    """conv -> eval-BatchNorm (-> + residual) (-> ReLU), or conv + bias (-> ReLU) when ``bn`` is None."""

    def __init__(self, name, conv, bn):
        self.name, self.conv, self.bn = name, conv, bn
        self.rstd = None
        self.first = False        # first layer of a plain stack: no input tangent, no data gradient
        self.needs_g = True       # the masked cotangent itself is read by a residual branch
        self.im2col = False       # tiny-Cin layer: runs as a 1x1 product over the im2col of the input
        self.train = False        # train-mode BatchNorm: batch statistics (recorded), tangent / adjoint carry
        self.mean_t = None        # the statistics' dependence on the layer input
        self.epi = self.tsum = False  # train mode: the tangent's partial sums come from the convolution's epilogue
        self.res_unit = None      # downsample unit whose output is added before the activation
        self.res_identity = False  # ... or the block input itself
        self.consumers = 0
        self.src = None           # what the convolution reads: "input", "pool" or the producing unit

    # per-channel affine map behind the convolution: BatchNorm statistics / scale / shift, or a bias
    @property
    def mean(self):
        if self.train:
            return self.mean_t
        return self.bn.running_mean if self.bn is not None else None

    @property
    def scale(self):
        return self.bn.weight if self.bn is not None else None

    @property
    def shift(self):
        return self.bn.bias if self.bn is not None else self.conv.bias


class FusedGGNEngine(_Operator):
    mode = ("fused curvature engine: own deterministic convolutions (split-K slabs summed by the consumer "
            "kernel), BatchNorm tangents/adjoints fused, 4 launches per conv-BN unit, downsample branches grouped "
            "with their block's first convolution")

    # ------------------------------------------------------------------------------------
    @classmethod
    def try_build(cls, loss, outputs, params, weight=1.0, group=None, hessian=False):
        if os.environ.get("HF_ENGINE", "1") == "0":
            return None
        ref = getattr(outputs, "_hf_model", None)
        model = ref() if ref is not None else None
        if model is None or not outputs.is_cuda or outputs.dtype != torch.float32 or outputs.dim() != 2:
            return None
        kinds = [cls] if cls is not FusedGGNEngine else [FusedGGNEngine, PlainStackEngine]
        if hessian:
            kinds = [k for k in kinds if k.supports_hessian]
        for kind in kinds:
            try:
                return kind(model, loss, outputs, params, weight, group, hessian=hessian)
            except _Unsupported as exc:
                # (a model the engine does not cover is the normal case: quiet unless asked;
                # a product that FAILED its check is always reported)
                if os.environ.get("HF_ENGINE_DEBUG") or getattr(exc, "loud", False):
                    warnings.warn(f"fused curvature engine ({kind.__name__}) not used: {exc}")
                if getattr(exc, "loud", False):
                    return None
            except _lib.Refused as exc:  # a kernel refused its arguments (alignment, size limits ...)
                warnings.warn(f"fused curvature engine ({kind.__name__}) not used: {exc}")
                return None
        return None

    # Hessian products (optimizer.py:450-455) by forward-over-reverse on the same kernels; residual nets with
    # EVAL-mode BatchNorm (the layer is a per-channel affine map; ReLU masks and max-pool positions are
    # piecewise constant): per unit, besides the GGN's terms, conv_D(g, V) and conv_W(t_x, g) (g: the step's
    # first-order cotangent, kept by the gradient sweep) as MORE SLABS of the same buffers, and the BatchNorm
    # scale's own second-order terms  g_a' += g_z * rstd * v_gamma ,  g_gamma' += sum g_z * rstd * t_a .
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
        self._layout(model)
        if self.hessian and self.train_bn:
            raise _Unsupported("Hessian products with train-mode BatchNorm (batch statistics couple the samples: "
                               "cross terms the engine does not carry)")
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
            if (os.environ.get("HF_TRAIN_SESSION", "1") != "0"
                    and all(u.bn.momentum is not None and u.bn.track_running_stats for u in self.units if u.train)):
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
            from .modelprep import release_records

            release_records(model)  # (the layers' records pinned this pass's activations until the next one)

    # ---- topology ---------------------------------------------------------------------
    def _param(self, p):
        if p is None:
            return None
        i = self._index.get(id(p))
        if i is None:
            raise _Unsupported("a layer parameter is not among the optimizer's parameters")
        return i

    def _layout(self, model):
        need = ("conv1", "bn1", "maxpool", "avgpool", "fc")
        if not all(isinstance(getattr(model, a, None), nn.Module) for a in need):
            raise _Unsupported("not a ResNet-family module tree")
        if hasattr(model, "layers") and isinstance(model.layers, nn.Sequential):
            blocks = list(model.layers)
        elif all(hasattr(model, f"layer{i}") for i in range(1, 5)):
            blocks = [b for i in range(1, 5) for b in getattr(model, f"layer{i}")]
        else:
            raise _Unsupported("no block list")
        x_in = getattr(self.outputs, "_hf_input", None)
        if x_in is None:
            raise _Unsupported("no recorded input")

        def io(m, n):
            rec = getattr(m, "_hf_io", None)
            if rec is None or len(rec) != n:
                raise _Unsupported(f"{type(m).__name__} has no record of this forward pass")
            return rec

        units = []
        group_is_set = self.group is not None

        def make_unit(name, conv, bn, relu_expected):
            if type(conv) is not nn.Conv2d or conv.bias is not None or conv.groups != 1:
                raise _Unsupported(f"{name}: unsupported convolution")
            if not getattr(conv, "_hf_channels_last", False) or tuple(conv.dilation) != (1, 1):
                raise _Unsupported(f"{name}: needs prepare_model(channels_last=True)")
            if not isinstance(bn, nn.BatchNorm2d):
                raise _Unsupported(f"{name}: not a BatchNorm2d")
            u = _Unit(name, conv, bn)
            cx, cy = io(conv, 2)
            rec = getattr(bn, "_hf_io", None)
            if bn.training:
                # train mode: the layer ran on stock ops and recorded its batch statistics (modelprep)
                if rec is None or len(rec) != 6 or group_is_set:
                    raise _Unsupported(f"{name}: train-mode BatchNorm without a record (or under data "
                                       "parallelism: batch statistics couple the samples of a shard)")
                bx, bres, by, brelu, rstd, mean_t = rec
                # (engine-owned static buffers: the own forward pass of a session rewrites them per batch)
                u.rec_stats = (mean_t, rstd)
                rstd, u.mean_t = rstd.clone(), mean_t.clone()
                u.train = True
            else:
                bx, bres, by, brelu, rstd = io(bn, 5)
            if not _same(cy, bx):
                raise _Unsupported(f"{name}: the BatchNorm does not consume the convolution's output")
            if brelu != relu_expected:
                raise _Unsupported(f"{name}: unexpected activation")
            # identity of activations: the RAW records (a 1-channel stem runs NCHW, its records are
            # converted below); the kernels get NHWC copies / views
            u.kx, u.ky, u.rx, u.ry, u.ra = cx.data_ptr(), by.data_ptr(), cx, by, cy
            u.x, u.a, u.y, u.relu, u.rstd, u.res = _cl(cx), _cl(cy), _cl(by), brelu, rstd, bres
            u.pw, u.pg, u.pb = self._param(conv.weight), self._param(bn.weight), self._param(bn.bias)
            units.append(u)
            return u

        # stem: conv1 -> bn1 (+ relu, fused by fuse_bn_relu or by a block-style forward) -> maxpool
        stem = make_unit("stem", model.conv1, model.bn1, True)
        if not _same(stem.rx, x_in.detach()) or stem.res is not None:
            raise _Unsupported("stem does not start at the network input")
        stem.src, stem.im2col, stem.first = "input", True, True
        self.model_ref, self._in_shape = model, tuple(x_in.shape)
        mp_x, mp_y = io(model.maxpool, 2)
        if not _same(mp_x, stem.ry):
            raise _Unsupported(f"maxpool does not follow the stem ({tuple(mp_x.shape)} {mp_x.stride()} "
                               f"{mp_x.data_ptr():x} vs {tuple(stem.y.shape)} {stem.y.stride()} {stem.y.data_ptr():x})")
        mp = model.maxpool
        self.stem, self.pool_args = stem, (mp.kernel_size, mp.stride, mp.padding, mp.dilation, mp.ceil_mode)
        cur = mp_y
        self.pool_out, self.pool_key, self._rec_pool = _cl(mp_y), mp_y.data_ptr(), mp_y
        self.blocks = []
        prev = "pool"  # producer of the current block input
        for bi, b in enumerate(blocks):
            convs = [n for n in ("conv1", "conv2", "conv3") if isinstance(getattr(b, n, None), nn.Conv2d)]
            if not getattr(b, "_hf_block_patched", False) or len(convs) < 2:
                raise _Unsupported(f"block {bi}: not a fused residual block")
            chain, inp = [], cur
            for k, cn in enumerate(convs):
                u = make_unit(f"block{bi}.{cn}", getattr(b, cn), getattr(b, "bn" + cn[-1]), True)
                if not _same(u.rx, inp):
                    raise _Unsupported(f"block {bi}.{cn}: input is not the previous activation")
                last = k == len(convs) - 1
                if (u.res is not None) != last:
                    raise _Unsupported(f"block {bi}.{cn}: unexpected residual")
                u.src = chain[-1] if chain else prev
                chain.append(u)
                inp = u.ry
            tail = chain[-1]
            ds = None
            if b.downsample is not None:
                d = b.downsample
                if not (isinstance(d, nn.Sequential) and len(d) == 2):
                    raise _Unsupported(f"block {bi}: unsupported downsample")
                ds = make_unit(f"block{bi}.downsample", d[0], d[1], False)
                if not _same(ds.rx, cur) or ds.res is not None or not _same(tail.res, ds.ry):
                    raise _Unsupported(f"block {bi}: downsample wiring")
                tail.res_unit = ds
                ds.src = prev
            else:
                if not _same(tail.res, cur):
                    raise _Unsupported(f"block {bi}: identity wiring")
                tail.res_identity = True
            self.blocks.append((chain, ds, cur))
            cur = tail.ry
            prev = tail
        ap_x, ap_y = io(model.avgpool, 2)
        if not _same(ap_x, cur):
            raise _Unsupported("avgpool does not follow the last block")
        fc = model.fc
        fc_x, fc_y = io(fc, 2)
        if fc_x.dim() != 2 or fc_x.shape[0] != cur.shape[0] or fc_x.shape[1] != cur.shape[1]:
            raise _Unsupported("the classifier does not take the pooled features")
        if not _same(fc_y, self.outputs.detach()):
            raise _Unsupported("the network output is not the classifier's output")
        self.fc, self.feat = fc, fc_x
        self.pfw, self.pfb = self._param(fc.weight), self._param(fc.bias)
        self.units = units
        self.tail = self.blocks[-1][0][-1]
        self.train_bn = any(u.train for u in units)
        used = {i for u in units for i in (u.pw, u.pg, u.pb)} | {self.pfw} | ({self.pfb} if self.pfb is not None else set())
        if used != set(range(len(self.params))):
            raise _Unsupported("the parameter list has entries the engine's layers do not cover")

    def layer_signature(self):
        """What the captured graphs of a session bake in about the model's layers besides shapes: module
        identities and every BatchNorm's mode / eps / momentum (kernel arguments).  Compared per step."""
        return tuple((id(u.conv), id(u.bn), None if u.bn is None else (u.bn.training, u.bn.eps, u.bn.momentum))
                     for u in self.units) + (bool(self.model_ref.training) if not self.units[0].bn is None else None,)

    # ---- loss Hessian (same contract as GGNOperator) -------------------------------------
    def _loss_setup(self, loss, outputs):
        (self._dl,) = torch.autograd.grad(loss, outputs, create_graph=True, retain_graph=True)
        self._ce = GGNOperator._closed_form_loss_hessian(self, _Node(_ce_node(loss)), outputs)
        # a plain softmax cross-entropy (checked numerically above): the engine can then evaluate
        # loss, probabilities and d loss / d logits itself, on its own forward pass -- which is what
        # lets ONE engine serve many steps and trial points (``session.EngineSession``)
        self.loss_spec = None
        if self._ce is not None and (not self.train_bn or self.train_own):
            spec = ce_loss_spec(loss, outputs)
            if spec is not None:
                self.loss_spec = spec
                self._set_quadratic(spec.get("quadratic"))
                self.set_targets(spec["targets"])
                self._loss_head()
                self._ce = (self._p, self._ce[1])  # the static buffer the own forward pass refreshes
                self._dl = None                     # (nothing of this step's autograd graph is kept)
        if self.hessian:
            if self.loss_spec is None:
                raise _Unsupported("Hessian products on the engine need a plain softmax cross-entropy loss")
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
                    raise _Unsupported("a regularised tensor is not among the optimizer's parameters")
                d[self._offs[i]: self._offs[i] + w.numel()] += float(coef)
        self._l2 = d

    def _theta(self):
        flat = self._flat_params
        if flat is not None and flat.data_ptr() == self.params[0].data_ptr():
            return flat
        return torch.cat([p.detach().reshape(-1) for p in self.params])

    # ---- own forward pass ------------------------------------------------------------------
    def set_batch(self, x, targets=None):
        """A new input batch of the same shape (and its targets): static input, the stem's im2col."""
        if tuple(x.shape) != tuple(self.x_in.shape):
            raise RuntimeError("engine: input shape changed")
        self.x_in.copy_(x)
        s = self.units[0]
        if s.im2col:
            cols = torch.nn.functional.unfold(self.x_in.contiguous(), tuple(s.conv.kernel_size),
                                              padding=tuple(s.conv.padding), stride=tuple(s.conv.stride))
            s.cols.copy_(cols.transpose(1, 2))
            s.cols_pad[:, :, :s.jcols].copy_(s.cols)
        for u in self.units:  # eval-mode statistics are constants -- unless somebody retrained them
            if u.bn is not None and not u.train and u.bn.running_var._version != u.rstd_version:
                torch.rsqrt(u.bn.running_var + u.bn.eps, out=u.rstd)
                u.rstd_version = u.bn.running_var._version
        if targets is not None:
            self.set_targets(targets)

    def set_targets(self, targets):
        if getattr(self, "_targets", None) is None:
            self._targets = torch.empty_like(targets)
            self._onehot = torch.zeros_like(self.logits)
        self._targets.copy_(targets)
        k = self.logits.shape[1]
        # class indices outside [0, K) (an ``ignore_index``) are not covered by the closed forms: the
        # flag is read back by the caller together with the loss value (no extra sync)
        self.bad_targets = ((self._targets < 0) | (self._targets >= k)).any()
        self._onehot.zero_()
        self._onehot.scatter_(1, self._targets.clamp(0, k - 1).view(-1, 1), 1.0)

    def refresh_weights(self, transposed=False):
        """The W halves of all [W | v_W] operands from the CURRENT parameters (one scatter launch
        when the parameters are views of one flat vector); ``transposed``: also the (I, H, W, O)
        copies the data-gradient convolutions read (once per step; trial points need only W)."""
        flat = self._flat_params
        if flat is not None and flat.data_ptr() == self.params[0].data_ptr():
            _lib.unpack_tangent(flat, self._slot_list, half=0)
        else:
            for u in self.units:
                if not u.im2col:
                    c = u.x.shape[1]
                    u.wcat[:, :c].copy_(self.params[u.pw].detach())
        if transposed:
            if flat is not None and flat.data_ptr() == self.params[0].data_ptr() and self._wt_slots:
                _lib.unpack_tangent(flat, self._wt_slots, half=2)
            else:
                for u in self.units:
                    if not u.im2col and not u.first:
                        u.wT.copy_(self.params[u.pw].detach().permute(1, 2, 3, 0))

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
        torch.softmax(self.logits, 1, out=self._p)
        lsm = torch.log_softmax(self.logits, 1)
        val = torch.nn.functional.nll_loss(lsm, self._targets, reduction=self.loss_spec["reduction"])
        if getattr(self, "_l2", None) is not None:
            theta = self._theta()
            val = val + 0.5 * torch.dot(self._l2 * theta, theta)
        self.loss_buf.copy_(val)

    def gradient(self, out=None):
        """``weight * d loss / d params`` of the softmax cross-entropy by ONE adjoint sweep of the
        engine (the gradient the reference takes with ``torch.autograd.grad``, optimizer.py:231-234),
        on the activations of the last ``forward_own``."""
        if self.loss_spec is None:
            raise RuntimeError("engine.gradient needs a softmax cross-entropy loss")
        if out is None:
            out = torch.empty(self.n, dtype=torch.float32, device=self.dev)
        g = (self._p - self._onehot) * self._ce[1]  # d loss / d logits
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
        if self._l2 is not None:
            out.addcmul_(self._l2, self._theta(), value=self.weight)
        return out

    def _swap_first_order(self):
        for u in self.units:
            u.g, u.g1 = u.g1, u.g
            u.ga, u.ga1 = u.ga1, u.ga

    def _feature_cotangent(self, g_feat):
        tail = self.tail
        if self._head_hw == 1:
            return g_feat.view(tail.y.shape)
        return _cl((g_feat / self._head_hw).view(g_feat.shape[0], -1, 1, 1).expand(tail.y.shape))

    _loss_hessian = GGNOperator._loss_hessian

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
            raise RuntimeError("engine.diag_ef needs a softmax cross-entropy loss and eval-mode BatchNorm")
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
                ga = (u.ga1 if first_order else u.ga)[i:i + 1]
                if u.im2col:
                    self._conv_slabs(2, u.wps, u.cols_pad[i:i + 1], ga, u.geo_w1, u.sW1, out_c=u.jcols)
                else:
                    self._conv_slabs(2, u.wps, u.x[i:i + 1], ga, u.geo1, u.sW1)
                k, hw = u.a.shape[1], u.a.shape[2] * u.a.shape[3]
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
            g = (self._p - self._onehot) * (self._ce[1] * scale)
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

    # ---- buffers -------------------------------------------------------------------------
    def _plan(self, direction, u, forward=False):
        n, c, h, w = u.x.shape
        k, _, r, s = u.conv.weight.shape
        cin = 2 * c if (direction == 0 and not forward) else c
        sp = _lib.load().hf_conv2d_nhwc_plan(direction, n, h, w, cin, k, r, s, u.conv.stride[0], u.conv.stride[1],
                                             u.conv.padding[0], u.conv.padding[1],
                                             int(os.environ.get("HF_CONV_BLOCKS", "0")))
        if sp < 1:
            raise _Unsupported(f"{u.name}: convolution geometry refused ({sp})")
        return sp

    def _allocate(self):
        """Every buffer the sweeps touch is owned by the engine and STATIC: the activations are
        recomputed in place by ``forward_own`` (own kernels), so one engine -- and the hipGraphs
        captured over it -- can serve every Newton step and every trial point of a step."""
        dev, f32 = self.dev, torch.float32
        cl = torch.channels_last

        def nhwc(shape):
            return torch.empty(shape, dtype=f32, device=dev).contiguous(memory_format=cl)

        self._nhwc = nhwc
        self.x_in = nhwc(self._in_shape)
        if self.pool_args is not None:
            self.pool_out = nhwc(self.pool_out.shape)
        for u in self.units:
            u.a, u.y = nhwc(u.a.shape), nhwc(u.y.shape)
            if u.train:
                u.rstd_version = None  # (u.rstd: batch statistics -- recorded by the model's pass or the own one's)
            elif u.bn is not None:
                u.rstd = torch.rsqrt(u.bn.running_var + u.bn.eps)
                u.rstd_version = u.bn.running_var._version
        tails = {id(c[-1]): c[0] for c, _, _ in self.blocks}
        for u in self.units:  # the convolution's input IS its producer's output buffer
            u.x = self.x_in if u.src == "input" else self.pool_out if u.src == "pool" else u.src.y
            if u.res_unit is not None:
                u.res = u.res_unit.y
            elif u.res_identity:
                u.res = tails[id(u)].x  # the block input
            else:
                u.res = None
        xcats = {}
        self._tangent_slots = {}
        for u in self.units:
            n, c, h, w = u.x.shape
            k, _, r, s = u.conv.weight.shape
            if u.im2col:
                if c * r * s > 256:
                    raise _Unsupported(f"{u.name}: too many taps for the im2col formulation")
                oh, ow = u.a.shape[2], u.a.shape[3]
                j = c * r * s
                u.cols = torch.empty((n, oh * ow, j), dtype=f32, device=dev)  # [N, OH*OW, c*r*s]
                u.geo = (n * oh * ow, 1, 1, j, k, 1, 1, (1, 1), (0, 0))
                # the weight gradient reads the im2col with rows padded to 16-byte multiples (zero
                # channels): 16-byte gathers instead of element-wise ones (28 -> 11 us for the 49-tap stem)
                jp = -(-j // 4) * 4
                u.cols_pad = torch.zeros((n, oh * ow, jp), dtype=f32, device=dev)
                u.geo_w = (n * oh * ow, 1, 1, jp, k, 1, 1, (1, 1), (0, 0))
                u.jcols = j
                if not u.conv.weight.is_contiguous():
                    raise _Unsupported(f"{u.name}: weight layout")
            else:
                if c % 4 or k % 4:
                    raise _Unsupported(f"{u.name}: channel counts must be multiples of 4")
                u.geo = (n, h, w, c, k, r, s, tuple(u.conv.stride), tuple(u.conv.padding))
                key = id(u.x)
                if key not in xcats:
                    xcats[key] = torch.zeros((n, 2 * c, h, w), dtype=f32, device=dev).contiguous(memory_format=cl)
                u.xcat = xcats[key]
                # [W | v_W]; slices of taps that never meet data stay 0
                u.wcat = torch.zeros((k, 2 * c, r, s), dtype=f32, device=dev).contiguous(memory_format=cl)
                u.wT = torch.empty((c, r, s, k), dtype=f32, device=dev)  # (I, H, W, O)
                u.live = _live_taps(h, w, r, s, u.conv.stride, u.conv.padding)
                self._tangent_slots[id(u)] = (self._offs[u.pw], u.wcat, c, u.live)
            oh, ow = u.a.shape[2], u.a.shape[3]
            u.rows, u.cout = n * oh * ow, k
            # split-K slab buffers
            if u.im2col:
                u.sT = u.sF = self._plan_stem(0, u.geo)
                u.sW = self._plan_stem(2, u.geo_w)
                u.sD = 0
            else:
                u.sT, u.sW = self._plan(0, u), self._plan(2, u)
                u.sD = 0 if u.first else self._plan(1, u)
                u.sF = self._plan(0, u, forward=True)
            u.tbuf = torch.empty((max(u.sT, u.sF), u.rows * k), dtype=f32, device=dev)
            # Hessian products add, per layer, conv_W(t_x, g) to the weight gradient and conv_D(g, V) to the
            # data gradient (g: the step's first-order cotangent): as MORE SLABS of the same buffers, which
            # the consumers sum anyway
            u.nW = u.sW * (2 if (self.hessian and not u.first) else 1)
            u.nD = u.sD * (2 if self.hessian else 1)
            u.wbuf = torch.zeros((u.nW, u.conv.weight.numel()), dtype=f32, device=dev)  # dead taps stay 0
            if u.sD:
                u.dbuf = torch.empty((u.nD, u.x.numel()), dtype=f32, device=dev)
            if self.hessian:
                u.ga1 = torch.empty_like(u.a)  # first-order cotangent of the convolution output (per step)
                if u.bn is not None:
                    u.g1 = torch.empty_like(u.a)   # ... and of the BatchNorm output (masked), per step
                    u.gah = torch.empty_like(u.a)  # g_a' + g_z * rstd * v_gamma: what the convolutions' adjoints read
                if u.sD:
                    u.vT = torch.empty((c, r, s, k), dtype=f32, device=dev)  # V as (I, H, W, O), per product
            u.g = torch.empty_like(u.a) if u.needs_g else None  # masked cotangent of the unit's output
            u.ga = torch.empty_like(u.a)   # cotangent of the convolution output
            # the BatchNorm adjoint shares the rows among `rb` workgroups per channel column; the
            # per-channel sums arrive as rb partial rows that hf_pack_ex adds up
            u.rb = 1
            if k % 4 == 0 and k // 4 <= 256 and u.rows >= 64:
                # row-major adjoint kernel: ~64 workgroups, each reading whole contiguous rows, one
                # pass of the row loop where the map is small enough (measured on the ResNet-18
                # bench: 32 workgroups x 2 passes 1124, 64 x 1 1150, 128 x 1 the same, 256 x 1 1138)
                rp = 256 // (k // 4)
                # (64 workgroups suit the <= 1.6 MB maps of ResNet-18; a 12.6 MB map of All-CNN-C needs the
                # whole chip: one workgroup per 32 KB of the map, 64 ... 1024)
                tgt = int(os.environ.get("HF_BN_ROW_BLOCKS", "0")) or min(1024, max(64, u.a.numel() * 4 // 32768))
                per = max(int(os.environ.get("HF_BN_ROW_PASSES", "1")) * rp, -(-u.rows // tgt))
                u.rb = -(-u.rows // per)
                if u.rb < 2:
                    u.rb = 1
            # (Hessian: the scale's second-order term arrives as `rb` more partial rows for hf_pack_ex to add)
            u.gw_rows = u.rb * (2 if (self.hessian and u.bn is not None) else 1)
            u.gw = torch.empty((u.gw_rows, k), dtype=f32, device=dev)
            u.gb = torch.empty((u.rb, k), dtype=f32, device=dev)
            if u.train:
                # The per-channel finalisation of a train-mode tangent / adjoint runs in the PROLOGUE of the elementwise
                # pass: its workgroups add the reduction's partial rows up themselves (hf_chan_affine_train).  The
                # forms of round 4 that handed over inside a launch or took a launch of their own were measured slower
                # (profiles/r04_train_bn_forms.jsonl) and are gone.
                if not (k % 4 == 0 and k // 4 <= 256):
                    raise _Unsupported(f"{u.name}: train-mode BatchNorm over {k} channels (the prologue form takes "
                                       "multiples of 4 up to 1024)")
                u.stat_part = torch.empty((u.rb, 2, k), dtype=torch.float64, device=dev)  # one-pass statistics
                # ... and the tangent's partial sums by the convolution's own epilogue (64x64-tile launches; one row per
                # (row tile, split): beyond HF_BN_EPILOGUE_ROWS rows the separate reduction's `rb` rows are cheaper
                # for the elementwise pass to add up)
                tp_rows = -(-u.rows // 64) * u.sT
                u.epi = (not u.im2col and not u.first and hasattr(u, "xcat")
                         and tp_rows <= int(os.environ.get("HF_BN_EPILOGUE_ROWS", "256"))
                         and os.environ.get("HF_BN_EPILOGUE", "1") != "0")
                if u.epi:
                    u.tp1 = torch.empty((tp_rows, k), dtype=f32, device=dev)
                    u.tpx = torch.empty((tp_rows, k), dtype=f32, device=dev)
        # where each unit's output goes besides its own dense buffer: the [t_x | x] operand of its
        # consumer -- the tangent into the first half, the value (forward pass) into the second
        for u in self.units:
            xc = xcats.get(id(u.y))
            c = u.y.shape[1]
            if xc is not None:
                u.tout, u.tout_ld, u.yout2 = xc[:, :c], 2 * c, xc[:, c:]
            else:
                u.tout, u.tout_ld, u.yout2 = torch.empty_like(u.y), 0, None
        self._xcats = xcats
        self._slot_list = list(self._tangent_slots.values())
        self._carry_ok = os.environ.get("HF_CARRY_SCATTER", "1") != "0"  # (the stem's launch carries the v_W scatter)
        # (I, H, W, O) copies: the weights (once per step) and, for Hessian products, V (per product)
        self._wt_slots = [(self._offs[u.pw], u.wT, u.x.shape[1]) for u in self.units if not u.im2col and not u.first]
        self._vt_slots = [(self._offs[u.pw], u.vT, u.x.shape[1]) for u in self.units
                          if self.hessian and not u.im2col and u.sD]
        self._allocate_pool()
        self._allocate_head()
        # the parameters as ONE flat vector, when they are consecutive views of one (the optimizer's
        # arena): every W half is then refreshed by a single scatter launch
        self._flat_params = _flat_view(self.params, self.n)

    def _allocate_pool(self):
        self.pool_t = self._xcats.get(id(self.pool_out))
        if self.pool_t is None:
            raise _Unsupported("nothing consumes the pooled stem output")
        if _pair(self.pool_args[3]) != [1, 1] or self.pool_args[4]:
            raise _Unsupported("max-pool with dilation / ceil_mode")
        self.pool_idx32 = torch.empty(tuple(self.pool_out.permute(0, 2, 3, 1).shape), dtype=torch.int32,
                                      device=self.dev)
        self._g_stem = torch.empty_like(self.stem.y)

    def _allocate_head(self):
        """Classifier head: (global average pool ->) linear layer."""
        f32, dev = torch.float32, self.dev
        tail = self.tail
        n, k = tail.y.shape[0], tail.y.shape[1]
        self._head_hw = tail.y.shape[2] * tail.y.shape[3]
        if self._head_hw == 1:
            self.feat = tail.y.permute(0, 2, 3, 1).reshape(n, k)  # a view: NHWC with a 1x1 map is [n, k]
            if self.feat.data_ptr() != tail.y.data_ptr():
                raise _Unsupported("feature view")
        else:
            self.feat = torch.empty((n, k), dtype=f32, device=dev)
        self.logits = torch.empty((n, self.fc.weight.shape[0]), dtype=f32, device=dev)
        self._p = torch.empty_like(self.logits)
        self.loss_buf = torch.zeros((), dtype=f32, device=dev)

    def _plan_stem(self, direction, geo):
        n, h, w, c, k, r, s, st, pd = geo
        sp = _lib.load().hf_conv2d_nhwc_plan(direction, n, h, w, c, k, r, s, 1, 1, 0, 0,
                                             int(os.environ.get("HF_CONV_BLOCKS", "0")))
        if sp < 1:
            raise _Unsupported(f"stem geometry refused ({sp})")
        return sp

    # ---- kernels ---------------------------------------------------------------------------
    def _conv_slabs(self, direction, out, act, mat, geo, splits, act_ld=0, out_c=0, mat_ld=0):
        n, h, w, c, k, r, s, st, pd = geo
        _lib.check(_lib.load().hf_conv2d_nhwc_slabs(
            direction, _ptr(out), _ptr(act), _ptr(mat), n, h, w, c, k, r, s, st[0], st[1], pd[0], pd[1], act_ld,
            mat_ld, out_c, splits, out.shape[1] if out.dim() == 2 else 0, _lib.HF_F32,
            _lib.current_stream_ptr(self.dev)), "hf_conv2d_nhwc_slabs")

    def _conv_carrying_scatter(self, u, mat, splits, src, half):
        """The im2col'd first layer's convolution (its weight operand ``mat`` is a slice of a flat vector, no
        scattered operand is read) in ONE launch with the scatter of ``src`` into the v_W (``half=1``) / W
        (``half=0``) halves of every other layer's operand (``hf_conv2d_nhwc_slabs_unpack``): the scatter hides
        behind the latency-bound convolution.  False: not taken (switched off, no scatter to carry, or a
        geometry / tensor count the merged launch refuses) -- the caller issues the two launches."""
        if not (self._carry_ok and u.im2col and self._slot_list):
            return False
        n, h, w, c, k, r, s_, st, pd = u.geo
        rc = _lib.load().hf_conv2d_nhwc_slabs_unpack(
            _ptr(u.tbuf), _ptr(u.cols), _ptr(mat), n, h, w, c, k, r, s_, st[0], st[1], pd[0], pd[1], 0, 0, splits,
            u.tbuf.shape[1], _ptr(src), *_lib.unpack_table(src, self._slot_list, half=half), _lib.HF_F32,
            _lib.current_stream_ptr(self.dev))
        if rc == _lib.HF_ERR_ARG:
            self._carry_ok = False
            return False
        _lib.check(rc, "hf_conv2d_nhwc_slabs_unpack")
        return True

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
            if u.tsum:
                px, p1, nparts = u.tpx, u.tp1, u.tp1.shape[0]
            else:
                _lib.check(lib.hf_chan_affine_bwd_ex(
                    None, _ptr(u.gw), _ptr(u.gb), None, _ptr(u.tbuf), u.sT, u.tbuf.shape[1], None, 1, 0, _ptr(u.a),
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

    def _bn_adjoint_pair_train(self, u1, srcs1, u2, srcs2):
        """Train mode, prologue form: both units' reduction passes in one launch, both elementwise passes in one."""
        self._bn_adjoint_pair(u1, srcs1, u2, srcs2, train=True)
        arr = (_lib.AffineTrainProblem * 2)()
        for q, u in zip(arr, (u1, u2)):
            self._affine_train_problem(q, u, u.ga, 0, u.g, 1, 0, u.gw, u.gb, u.rb, None, None, None)
        _lib.check(_lib.load().hf_chan_affine_train_pair(_lib.ctypes.cast(arr, _lib.c_void_p), _lib.HF_F32,
                                                         _lib.current_stream_ptr(self.dev)), "hf_chan_affine_train_pair")

    def _bn_tangent_pair(self, u1, u2, v):
        """The BatchNorm tangents of two units without residual input in ONE launch."""
        arr = (_lib.AffineProblem * 2)()
        for q, u in zip(arr, (u1, u2)):
            n, k, oh, ow = u.a.shape
            q.out, q.a, q.x = u.tout.data_ptr(), u.tbuf.data_ptr(), u.a.data_ptr()
            q.mean, q.rstd, q.w = u.bn.running_mean.data_ptr(), u.rstd.data_ptr(), u.bn.weight.data_ptr()
            q.q = v.data_ptr() + 4 * self._offs[u.pg]
            q.r = v.data_ptr() + 4 * self._offs[u.pb]
            q.add, q.mask_src, q.relu_self = None, (u.y.data_ptr() if u.relu else None), 0
            q.n, q.c, q.hw, q.out_ld, q.add_ld = n, k, oh * ow, u.tout_ld, 0
            q.a_splits, q.a_slab = u.sT, u.tbuf.shape[1]
        _lib.check(_lib.load().hf_chan_affine_pair(_lib.ctypes.cast(arr, _lib.c_void_p), _lib.HF_F32,
                                                   _lib.current_stream_ptr(self.dev)), "hf_chan_affine_pair")

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
        self._bn_adjoint(u, srcs)
        if not (self._second and u.bn is not None):
            self._conv_adjoint(u)
            return
        # ---- Hessian product: the tangent of the backward sweep through this unit --------------------------
        lib, st = _lib.load(), _lib.current_stream_ptr(self.dev)
        n, k, oh, ow = u.a.shape
        v = self._v
        v_gamma = v[self._offs[u.pg]: self._offs[u.pg] + k]
        if not self._extras_parallel:
            self._hessian_extras(u)
        # the convolution's cotangent tangent:  g_a' + g_z * rstd * v_gamma
        _lib.check(lib.hf_chan_affine_ex(
            _ptr(u.gah), _ptr(u.g1), None, None, _ptr(u.rstd), _ptr(v_gamma), None, None, _ptr(u.ga), None, 0, n, k,
            oh * ow, 1, 0, 0, 1, 0, _lib.HF_F32, st), "hf_chan_affine_ex")
        if self._extras_mode == 2 and not u.im2col and not u.first:
            # the chain's launch also computes conv_D(g_a, V) -- the one extra term the chain itself needs next
            _lib.conv_group_slabs([(1, u.dbuf, u.gah, u.wT, u.geo, u.sD, 0, 0), (2, u.wbuf, u.x, u.gah, u.geo, u.sW, 0, 0),
                                   (1, u.dbuf[u.sD:], u.ga1, u.vT, u.geo, u.sD, 0, 0)], self.dev)
        else:
            self._conv_adjoint(u, u.gah)

    def _hessian_extras(self, u):
        """The terms of a Hessian product that do NOT depend on the adjoint chain -- only on the tangent sweep's
        results and the step's first-order cotangents: the scale's  sum_rows g_z * rstd * t_a  (t_a = the sum of the
        tangent convolution's slabs, still in place) as `rb` more partial rows of the gw buffer, and conv_D(g_a, V) /
        conv_W(t_x, g_a) as MORE SLABS of the same buffers (the consumers sum them anyway)."""
        if u.bn is not None:
            n, k, oh, ow = u.a.shape
            _lib.check(_lib.load().hf_chan_affine_bwd_ex(
                None, _ptr(u.gw[u.rb:]), None, None, _ptr(u.tbuf), u.sT, u.tbuf.shape[1], None, 1, 0, _ptr(u.g1),
                _ptr(self._zeros(k)), _ptr(u.rstd), None, None, n, k, oh * ow, 1, u.rb, _lib.HF_F32,
                _lib.current_stream_ptr(self.dev)), "hf_chan_affine_bwd_ex")
        if not u.im2col and not u.first:
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
    # HF_HESSIAN_PARALLEL: 0 = everything in sequence on the chain; 1 = all extras on the side branch, the chain waits
    # per unit for the data-gradient slabs it needs (one cross-branch dependency per unit); 2 = conv_D(g, V) inside
    # the chain's own grouped launch, ONLY results nobody on the chain reads on the side branch: one fork, one join.
    _extras_parallel = False
    _extras_mode = 0
    _extras_default = 2

    def _extras_fork(self):
        # (``_extras_allowed = False``: the caller already runs this engine on one of several parallel branches --
        # session.AccumulatedSession -- and a fork inside a forked capture branch crashes hipStreamEndCapture
        # on this stack: segfault in capture_end, round-4 batch r4f)
        # (measured, profiles/r04_hessian_parallel_branch.jsonl: ResNet-18 form 1 922-930, form 2 947-959 matvecs/s;
        # All-CNN-C form 1 521, form 2 283 -- its 128-wide tile configurations spill a three-problem argument block)
        mode = int(os.environ.get("HF_HESSIAN_PARALLEL", str(self._extras_default)))
        self._extras_parallel = mode != 0 and getattr(self, "_extras_allowed", True)
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

    def _zeros(self, k):
        cache = self.__dict__.setdefault("_zeros_cache", {})
        if k not in cache:
            cache[k] = torch.zeros(k, dtype=torch.float32, device=self.dev)
        return cache[k]

    def _dslabs(self, u):
        """How many data-gradient slabs the consumers of ``u``'s input cotangent must sum in the current sweep."""
        return u.nD if self._second else u.sD

    def _bn_adjoint(self, u, srcs, ga=None):
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
        if u.first:  # the network input needs no gradient
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

    # ---- side branch for launches that are off the adjoint sweep's dependency chain -------------------------
    # (inside a hipGraph capture: a parallel branch of the graph; eager: a second stream)
    def _side_setup(self):
        if getattr(self, "_xside", None) is None:
            self._xside = torch.cuda.Stream(device=self.dev)
            self._xfork = torch.cuda.Event()
            self._xev = {id(u): torch.cuda.Event() for u in self.units}

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

    # ---- the product in two phases, for overlapping the all-reduce with the rest of the sweep ----
    def phase_split(self, tail_fraction=0.7):
        """Block index ``cut`` such that the parameters of blocks ``cut ...`` and of the classifier are a
        contiguous SUFFIX of the flat vector holding at least ``tail_fraction`` of the entries that travel
        (the adjoint sweep finishes them first: ResNet-18 on 28x28 inputs, layer3 + layer4 + fc = 14 of
        17 MB after ~60 % of the product), with the flat offset of that suffix; ``None`` if the layout does
        not allow it."""
        if not self.blocks or self.fc is None:
            return None
        live = self._live_counts()
        total = sum(live)
        acc = live[self.pfw] + (live[self.pfb] if self.pfb is not None else 0)
        cut = None
        for bi in range(len(self.blocks) - 1, 0, -1):
            chain, ds, _ = self.blocks[bi]
            acc += sum(live[i] for u in chain + ([ds] if ds is not None else []) for i in (u.pw, u.pg, u.pb))
            if acc >= tail_fraction * total:
                cut = bi
                break
        if cut is None:
            return None
        late = {i for bi in range(cut, len(self.blocks)) for u in self.blocks[bi][0] + ([self.blocks[bi][1]] if self.blocks[bi][1] is not None else [])
                for i in (u.pw, u.pg, u.pb)} | {self.pfw} | ({self.pfb} if self.pfb is not None else set())
        first = min(late)
        if late != set(range(first, len(self.params))):
            return None  # the late layers' parameters are not a suffix of the vector
        return cut, first, self._offs[first]

    def _live_counts(self):
        counts = [p.numel() for p in self.params]
        for u in self.units:
            if not u.im2col and getattr(u, "live", 0):
                rs = u.conv.weight.shape[2] * u.conv.weight.shape[3]
                counts[u.pw] = u.conv.weight.numel() // rs * bin(u.live).count("1")
        return counts

    def local_phase_a(self, v, out, split):
        """Tangent sweep, head, adjoint sweep of blocks ``cut ...``, and the suffix of the product they
        determine (``out[offset:]``)."""
        cut, first, offset = split
        v = v.detach()
        self._tangent_stem(v, carry_scatter=True)
        self._tangent_blocks(v)
        g_last, g_fw, g_fb = self._head(v)
        self._phase_state = self._adjoint_blocks(g_last, last_block=cut)
        self._gather_range(out, g_fw, g_fb, first, len(self.params))

    def local_phase_b(self, out, split):
        """The rest of the adjoint sweep and ``out[:offset]``."""
        cut, first, offset = split
        pool_srcs = self._adjoint_blocks(None, first=cut - 1, last_block=0, incoming=self._phase_state)
        self._adjoint_stem(pool_srcs)
        self._gather_range(out, None, None, 0, first)

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

    # ---- tangent sweep -------------------------------------------------------------------------
    def _pool_geometry(self):
        """(n, h, w, oh, ow, c) of the stem's max-pool (window maxima's positions: ``forward_own``)."""
        pn, _, ph, pw = self.stem.y.shape
        return pn, ph, pw, self.pool_out.shape[2], self.pool_out.shape[3], self.pool_out.shape[1]

    def _tangent_stem(self, v, carry_scatter=False):
        """Stem: conv(x, v_W) as a 1x1 convolution on the im2col'd input (the input has no tangent; v_W is a
        slice of ``v`` itself).  ``carry_scatter``: the launch also carries the scatter of every other layer's v_W
        into its ``[W | v_W]`` operand (``hf_conv2d_nhwc_slabs_unpack``) -- nothing in the stem reads those."""
        s = self.stem
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
        for chain, ds, _x in self.blocks:
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
        for bi in range(first, last_block - 1, -1):
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
                    _lib.conv_group_slabs(
                        [(1, u.dbuf, u.ga, u.wT, u.geo, u.sD, 0, 0), (2, u.wbuf, u.x, u.ga, u.geo, u.sW, 0, 0),
                         (1, ds.dbuf, ds.ga, ds.wT, ds.geo, ds.sD, 0, 0), (2, ds.wbuf, ds.x, ds.ga, ds.geo, ds.sW, 0, 0)],
                        self.dev)
                else:
                    self._adjoint_unit(u, incoming.pop(id(u)))
                if k > 0:
                    incoming.setdefault(id(chain[k - 1]), []).append((u.dbuf, self._dslabs(u), u.dbuf.shape[1]))
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

    def _grouping(self):
        # (Hessian products carry extra terms per unit: the plain one-unit launches)
        return os.environ.get("HF_ENGINE_GROUP", "1") != "0" and not self.hessian

    def _head_fused(self, hw, v_fw):
        """Whether ``hf_linear_ce_head`` applies: closed-form softmax-CE Hessian, a 1x1 final map
        (the pooling is then the identity), a small dense head, 16-byte aligned operands."""
        ok = getattr(self, "_head_ok", None)
        if ok is None:
            fw = self.fc.weight
            k, f = fw.shape
            ok = (
                os.environ.get("HF_ENGINE_HEAD", "1") != "0" and self._ce is not None and hw == 1
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

    def _pack_args(self, first_order=False):
        """(tensors, perms, splits) of ``hf_pack_ex``.  ``first_order``: a gradient sweep of a Hessian
        engine fills only the first ``sW`` weight-gradient slabs of each layer."""
        if first_order and self.hessian:
            tensors, perms, splits = self._pack_args()
            splits = dict(splits)
            for u in self.units:
                if u.nW != u.sW:
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

    def _tgeo(self, u):
        n, h, w, c, k, r, s, st, pd = u.geo
        return (n, h, w, 2 * c, k, r, s, st, pd)

    def __call__(self, v, out=None):
        self.calls += 1
        return self.reduce(self.local(v, out))

    # ---- data parallelism: only the entries that can be non-zero travel ----------------------
    def _live_segments(self):
        """Description of the product's entries that are not structurally zero -- the weight slices of
        kernel taps that never meet data are zero on every rank (``_live_taps``) --, or ``None`` when
        (almost) everything is live.  ResNet-18 on 28x28 inputs: 4.3 M of 11.2 M entries.

        Layout of what travels: the dense PREFIX of the vector (everything before the first tensor with dead
        taps: stem .. layer3, 2.8 M entries) is all-reduced IN PLACE in the full vector -- no copy at all;
        the rest (layer4's live taps, its BatchNorm vectors, the classifier: 1.5 M entries) is gathered into
        the compact staging vector by ``hf_live_copy``, all-reduced there and scattered back."""
        if not hasattr(self, "_live_segs"):
            self._live_segs = None
            masked = {u.pw: u for u in self.units if not u.im2col and getattr(u, "live", 0)}
            segs, dead, run_start = [], 0, None  # (full offset, count in the full vector, period, mask)
            brk = getattr(self, "_seg_break", None)  # parameter index at which a dense run must end
            self._seg_cut = None                     # (chunked all-reduce: the suffix starts a segment)
            for i, p in enumerate(self.params):
                off = self._offs[i]
                if i == brk:
                    if run_start is not None:
                        segs.append((run_start, off - run_start, 0, 0))
                        run_start = None
                    self._seg_cut = (len(segs), off - dead)  # (segment index, compact offset) of the suffix
                if i in masked:
                    if run_start is not None:
                        segs.append((run_start, off - run_start, 0, 0))
                        run_start = None
                    u = masked[i]
                    rs = p.shape[2] * p.shape[3]
                    segs.append((off, p.numel(), rs, u.live))
                    dead += p.numel() // rs * (rs - bin(u.live).count("1"))
                elif run_start is None:
                    run_start = off
            if run_start is not None:
                segs.append((run_start, self.n - run_start, 0, 0))
            # the in-place prefix: leading dense segments (at most two: a chunk break may cut the run), each
            # worth a collective of its own (>= 1 MB) and 16-byte aligned
            n_pre, prefix = 0, 0
            if os.environ.get("HF_INPLACE_PREFIX", "1") != "0":
                while (n_pre < min(2, len(segs) - 1) and segs[n_pre][2] == 0 and segs[n_pre][1] >= (1 << 18)
                       and (segs[n_pre][0] + segs[n_pre][1]) % 4 == 0):
                    prefix = segs[n_pre][0] + segs[n_pre][1]
                    n_pre += 1
            if (masked and dead >= 0.2 * self.n and len(segs) - n_pre <= 24
                    and os.environ.get("HF_COMPACT_ALLREDUCE", "1") != "0"):
                self._prefix_runs = [(sg[0], sg[0] + sg[1]) for sg in segs[:n_pre]]
                if self._seg_cut is not None:
                    k, coff = self._seg_cut
                    # (cut in segment units of the STAGED list and staged offsets; k < n_pre: inside the prefix)
                    self._seg_cut = (k - n_pre, coff - prefix) if k >= n_pre else (k - n_pre, 0)
                segs = segs[n_pre:]
                arr = lambda col: (_lib.c_int64 * len(segs))(*[sg[col] for sg in segs])
                self._live_segs = (arr(0), arr(1), arr(2), arr(3), len(segs))
                self._n_live = self.n - dead
                self._compact = torch.empty(self.n - dead - prefix, dtype=torch.float32, device=self.dev)
        return self._live_segs

    def _reduce_pieces(self, full, part=None):
        """The tensors one product's all-reduce consists of: in-place slices of ``full`` (the dense prefix) and
        the compact staging vector; ``part``: "head" / "tail" of the chunked layout (``_seg_cut``)."""
        runs = [full[a:b] for a, b in self._prefix_runs]
        if part is None:
            pieces = runs + [self._compact]
        else:
            k, coff = self._seg_cut
            if k < 0:  # the cut lies inside the prefix: runs[:cut] are the head, everything else the tail
                cut = len(runs) + k
                pieces = runs[:cut] if part == "head" else runs[cut:] + [self._compact]
            else:
                pieces = runs + [self._compact[:coff]] if part == "head" else [self._compact[coff:]]
        return [t for t in pieces if t.numel() > 0]

    def _live_copy(self, full, scatter, part=None):
        """Gather (``scatter=False``) the staged live entries of ``full`` into the compact vector, or scatter
        them back; ``part``: "head" / "tail" of the chunked layout (segments before / from ``_seg_cut``)."""
        offs, counts, periods, masks, ns = self._live_segs
        lo, hi, comp = 0, ns, self._compact
        if part is not None:
            k, coff = self._seg_cut
            k = max(k, 0)
            lo, hi, comp = (0, k, self._compact[:coff]) if part == "head" else (k, ns, self._compact[coff:])
        if hi <= lo:
            return
        sub = lambda a: (_lib.c_int64 * (hi - lo))(*a[lo:hi])  # noqa: E731
        _lib.check(_lib.load().hf_live_copy(_ptr(full), _ptr(comp), int(scatter), sub(offs), sub(counts), sub(periods),
                                            sub(masks), hi - lo, _lib.HF_F32, _lib.current_stream_ptr(self.dev)),
                   "hf_live_copy")

    @property
    def reduce_bytes(self):
        return 4 * (self.n if self._live_segments() is None else self._n_live)

    def reduce(self, t, group=None):
        """Sum of the local products over the ranks.  The structurally-zero entries are zero on
        every rank, so only the live ones travel: the dense prefix in place, the rest through the compact
        staging vector (17 MB instead of 44.7 MB per product on the ResNet-18 workload)."""
        group = self.group if group is None else group
        if group is None:
            return t
        if (self._live_segments() is None or t.dtype != torch.float32 or t.numel() != self.n
                or not t.is_contiguous() or not t.is_cuda):
            return _all_reduce_sum(t, group)
        from .distributed import all_reduce_sum_multi

        self._live_copy(t, False)
        all_reduce_sum_multi(self._reduce_pieces(t), group)  # (direct RCCL: one grouped launch)
        self._live_copy(t, True)
        return t


    # ---- safety net --------------------------------------------------------------------------
    # relative max-norm distance to the autograd product above which the engine is refused; fp32
    # products of the shipped workloads agree to ~1e-6.  Deep, badly conditioned nets (the random-init
    # ResNet-50) scatter more for EVERY fp32 implementation: callers that have measured what stock
    # fp32 autograd achieves against float64 (bench.py) raise it to max(1e-5, 5 x that error)
    verify_tol = float(os.environ.get("HF_ENGINE_VERIFY_TOL", "1e-5"))

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
            from .curvature import HessianOperator

            want = HessianOperator(loss, self.params).local(v)
        else:
            want = GGNOperator(loss, self.outputs, self.params).local(v)
        err = float((got - want).abs().max() / want.abs().max().clamp_min(1e-30))
        if not err < FusedGGNEngine.verify_tol:
            exc = _Unsupported(f"engine {'Hessian ' if self.hessian else ''}product differs from the autograd product by {err:.2e} "
                               f"(tolerance {FusedGGNEngine.verify_tol:.1e}); using the autograd operator")
            exc.loud = True
            raise exc
        verified.add(key)


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
        if first.im2col:  # (the first layer's launch carries the v_W scatter of all the others)
            vw = v[self._offs[first.pw]: self._offs[first.pw] + first.conv.weight.numel()]
            carried = self._conv_carrying_scatter(first, vw, first.sT, v, 1)
        if self._slot_list and not carried:
            _lib.unpack_tangent(v, self._slot_list)  # v_W halves of all [W | v_W] operands: one launch
        for u in self.units:
            if u.im2col:  # no input tangent: conv(x, v_W) as a 1x1 product over the im2col
                vw = v[self._offs[u.pw]: self._offs[u.pw] + u.conv.weight.numel()]
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
            ga = u.ga1 if (self.hessian and first_order) else u.ga
            self._bn_adjoint(u, srcs, ga)
            if second and not u.im2col and not u.first:
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
        if self.loss_spec is None:
            raise _Unsupported("the plain-stack engine needs a plain softmax cross-entropy loss")


class _Unsupported(Exception):
    pass


def ce_loss_spec(loss, outputs, check_values=True):
    """``{"reduction", "targets"}`` if ``loss`` is ``F.cross_entropy(outputs, targets)`` with class-index
    targets, no class weights, no label smoothing and no ignored target -- read off the autograd
    graph (``NllLossBackward0 <- LogSoftmaxBackward0 <- outputs``) --, else ``None``."""
    fn = _ce_node(loss)
    try:
        if fn is None or fn.name() != "NllLossBackward0" or outputs.dim() != 2:
            return None
        lsm = fn.next_functions[0][0]
        if lsm is None or lsm.name() != "LogSoftmaxBackward0" or lsm._saved_dim not in (1, -1):
            return None
        src = lsm.next_functions[0][0]
        if outputs.grad_fn is not None:
            if src is not outputs.grad_fn:
                return None
        elif src is None or getattr(src, "variable", None) is not outputs:
            return None  # (a leaf: the logits a persistent session handed out)
        if fn._saved_weight is not None:
            return None
        reduction = {1: "mean", 2: "sum"}.get(fn._saved_reduction)
        targets = fn._saved_target
        if reduction is None or targets.dim() != 1 or targets.dtype != torch.int64:
            return None
        # (two host syncs: at engine construction only; a later step with an ignored / negative
        # target shows up as a loss value the session does not reproduce)
        if check_values and (bool((targets == fn._saved_ignore_index).any()) or bool((targets < 0).any())):
            return None
    except AttributeError:
        return None
    spec = {"reduction": reduction, "targets": targets.detach()}
    if fn is not loss.grad_fn:
        spec["quadratic"] = loss._hf_quadratic
    return spec


class _Node:
    """Stands in for a loss tensor where only its ``grad_fn`` is looked at."""

    def __init__(self, fn):
        self.grad_fn = fn


def _ce_node(loss):
    """The autograd node of the cross-entropy inside ``loss``: ``loss.grad_fn`` itself, or -- for a loss
    tagged ``_hf_quadratic = ((coef, [tensors]), ...)`` by its constructor (``testproblems.l2_regularized``:
    ``loss = cross_entropy + sum 0.5 * coef * ||w||^2``) -- the addend that is an ``NllLossBackward0``."""
    fn = loss.grad_fn
    if fn is not None and fn.name() == "AddBackward0" and getattr(loss, "_hf_quadratic", None):
        for nxt, _ in fn.next_functions:
            if nxt is not None and nxt.name() == "NllLossBackward0":
                return nxt
        return None
    return fn


def _pair(v):
    return [v, v] if isinstance(v, int) else list(v)


def _flat_view(params, n):
    """The parameters as one flat fp32 vector if they are consecutive, contiguous views of one
    storage in list order (``utils.ParameterArena``), else ``None``."""
    p0 = params[0]
    if p0.dtype != torch.float32 or not p0.is_cuda:
        return None
    base, off = p0.data_ptr(), 0
    for p in params:
        if not p.is_contiguous() or p.data_ptr() != base + 4 * off or p.dtype != torch.float32:
            return None
        off += p.numel()
    try:
        if p0.untyped_storage().nbytes() < 4 * (p0.storage_offset() + n) or base % 16:
            return None
        return p0.detach().as_strided((n,), (1,), p0.storage_offset())
    except Exception:  # noqa: BLE001
        return None


def _live_taps(h, w, r, s, stride, padding):
    """Bit mask (bit ``i*s + j``) of the kernel taps that meet data at some output position; 0 when
    all do or the mask does not fit the kernels' 16 bits.  The rule ``hf_conv2d_nhwc`` drops taps
    by: a 3x3 kernel on a 1x1 map only ever uses its centre tap, the other 8/9 of the layer's
    weight tangent / weight gradient are never read / structurally zero."""
    if r * s > 16 or os.environ.get("HF_ENGINE_LIVE", "1") == "0":
        return 0
    oh = (h + 2 * padding[0] - r) // stride[0] + 1
    ow = (w + 2 * padding[1] - s) // stride[1] + 1
    rows = [any(0 <= o * stride[0] - padding[0] + i < h for o in range(oh)) for i in range(r)]
    cols = [any(0 <= o * stride[1] - padding[1] + j < w for o in range(ow)) for j in range(s)]
    mask = sum(1 << (i * s + j) for i in range(r) for j in range(s) if rows[i] and cols[j])
    return 0 if mask == (1 << (r * s)) - 1 else mask
