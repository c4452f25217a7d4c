"""What the engine's modules share: pointer helpers, the per-unit record, loss recognition, kernel-tap liveness."""

import os

import torch

from .. import _lib

_P = _lib.c_void_p


def _ptr(t):
    return _P(t.data_ptr()) if t is not None else None


def _same(a, b):
    """Two records refer to the same activation (records are detached: compare storage)."""
    return a is not None and b is not None and a.data_ptr() == b.data_ptr() and a.shape == b.shape


def _cl(t):
    return t.contiguous(memory_format=torch.channels_last)


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
        self.dead = False         # frozen, behind frozen layers only: no tangent, no cotangent (topology._mark_dead_prefix)
        self.no_dgrad = False     # a live unit whose input carries no tangent: its data gradient is not needed

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
    spec = {"kind": "ce", "reduction": reduction, "targets": targets.detach()}
    if fn is not loss.grad_fn:
        spec["quadratic"] = loss._hf_quadratic
    return spec


def mse_loss_spec(loss, outputs):
    """``{"kind": "mse", "reduction", "targets"}`` if ``loss`` is ``F.mse_loss(outputs, targets)`` (``nn.MSELoss``,
    mean or sum) with constant targets of the outputs' shape -- the loss of the reference's own examples and tests
    (examples/run_mwe.py:19, tests/test_utils.py:47) -- read off the autograd graph; else ``None``.  Its gradient
    w.r.t. the outputs is ``2 c (outputs - targets)`` and its Hessian ``2 c I`` with ``c = 1 / numel`` (mean) or 1
    (sum): the loss head of optimizer.py:457-462 in closed form."""
    fn = getattr(loss, "grad_fn", None)
    try:
        if fn is None or fn.name() != "MseLossBackward0" or outputs.dim() != 2:
            return None
        src = fn.next_functions[0][0]
        if outputs.grad_fn is not None:
            if src is not outputs.grad_fn:
                return None
        elif src is None or getattr(src, "variable", None) is not outputs:
            return None  # (a leaf: the logits a persistent session handed out)
        if len(fn.next_functions) > 1 and fn.next_functions[1][0] is not None:
            return None  # (targets that are themselves differentiated)
        reduction = {1: "mean", 2: "sum"}.get(fn._saved_reduction)
        targets = fn._saved_target
        if reduction is None or tuple(targets.shape) != tuple(outputs.shape) or targets.dtype != outputs.dtype:
            return None
    except AttributeError:
        return None
    return {"kind": "mse", "reduction": reduction, "targets": targets.detach()}


def loss_spec_of(loss, outputs, check_values=True):
    """The losses the engine evaluates itself -- loss value, gradient and loss Hessian w.r.t. the model output in closed
    form, on its own forward pass (what lets ONE engine serve many steps and trial points): a plain softmax
    cross-entropy (``ce_loss_spec``) or a mean-squared error (``mse_loss_spec``); else ``None``."""
    spec = ce_loss_spec(loss, outputs, check_values)
    return spec if spec is not None else mse_loss_spec(loss, outputs)


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
    if r * s > 16:
        return 0
    oh = (h + 2 * padding[0] - r) // stride[0] + 1
    ow = (w + 2 * padding[1] - s) // stride[1] + 1
    rows = [any(0 <= o * stride[0] - padding[0] + i < h for o in range(oh)) for i in range(r)]
    cols = [any(0 <= o * stride[1] - padding[1] + j < w for o in range(ow)) for j in range(s)]
    mask = sum(1 << (i * s + j) for i in range(r) for j in range(s) if rows[i] and cols[j])
    return 0 if mask == (1 << (r * s)) - 1 else mask
