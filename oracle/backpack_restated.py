"""TEST INFRASTRUCTURE ONLY -- restatement of the third-party arithmetic the
reference's curvature path delegates to.

The reference computes its curvature-vector products with BackPACK
(PyPI ``backpack-for-pytorch``, pinned ``>=1.5.0,<2.0.0`` in
``/root/reference/setup.py:16``), which is NOT vendored under
``/root/reference`` and not installed in this image (no network).  Call sites in
the reference:

  * ``ggn_vector_product_from_plist(loss, outputs, plist, v)`` optimizer.py:461
  * ``hessian_vector_product(loss, plist, v)`` ........................ optimizer.py:454
  * ``extend`` / ``backpack(SumGradSquared())`` / ``p.sum_grad_squared`` preconditioners.py:43-53

This file restates BackPACK 1.x's *published* algorithm for those entry points
(``backpack/hessianfree/{rop,lop,hvp,ggnvp}.py``): the R-operator is emulated
by double backward through a dummy cotangent, the L-operator is one reverse
pass, ``Hv`` is the R-op of the gradient, and ``GGN v = J^T (H_L (J v))``.
``SumGradSquared`` is restated from its definition (sum over samples of the
squared individual gradients of the *mean/sum-reduced* loss, i.e. with the
``1/N`` factor inside for ``mean``).

Parity status: the reference's own tests pin ``Hv`` (test_optimizer.py:97-155,
one Newton step on a quadratic) and ``SumGradSquared`` (test_preconditioners.py:
54-99) but NOT ``GGN v`` against an independent truth;
``tests/test_host_logic_cpu.py::test_curvature_products_match_reference_and_explicit_matrices``
therefore checks this restatement against an explicitly materialised ``J^T H_L J``.

``install_as_backpack()`` registers this module under the import names the
reference uses, so that ``tests/golden/make_golden.py`` can import the real
``hessianfree.optimizer`` in the build container.  It is never used on the GPU
box and never by the product.
"""

import sys
import types

import torch


def _densify(grads, like):
    return tuple(
        torch.zeros_like(t) if g is None else g for g, t in zip(grads, like)
    )


def L_op(ys, xs, ws, retain_graph=True, detach=True):
    """Vector-Jacobian product ``ws^T (d ys / d xs)`` (one reverse pass)."""
    out = torch.autograd.grad(
        ys,
        xs,
        grad_outputs=ws,
        create_graph=True,
        retain_graph=retain_graph,
        allow_unused=True,
    )
    out = _densify(out, xs if isinstance(xs, (list, tuple)) else [xs])
    return tuple(o.detach() for o in out) if detach else out


def R_op(ys, xs, vs, retain_graph=True, detach=True):
    """Jacobian-vector product ``(d ys / d xs) vs`` via the double-backward
    trick: differentiate ``u -> J^T u`` with respect to the dummy ``u``."""
    if isinstance(ys, (list, tuple)):
        dummies = [torch.zeros_like(y, requires_grad=True) for y in ys]
    else:
        dummies = torch.zeros_like(ys, requires_grad=True)
    JTu = torch.autograd.grad(
        ys,
        xs,
        grad_outputs=dummies,
        create_graph=True,
        retain_graph=retain_graph,
        allow_unused=True,
    )
    vs_list = vs if isinstance(vs, (list, tuple)) else [vs]
    pairs = [(g, v) for g, v in zip(JTu, vs_list) if g is not None]
    out = torch.autograd.grad(
        [g for g, _ in pairs],
        dummies,
        grad_outputs=[v for _, v in pairs],
        create_graph=True,
        retain_graph=True,
        allow_unused=True,
    )
    dl = dummies if isinstance(dummies, (list, tuple)) else [dummies]
    out = _densify(out, dl)
    return tuple(o.detach() for o in out) if detach else out


def hessian_vector_product(f, params, v, grad_params=None, detach=True):
    """``(d^2 f / d params^2) v`` as the R-op of the gradient."""
    if grad_params is not None:
        df = tuple(grad_params)
    else:
        df = torch.autograd.grad(f, params, create_graph=True, retain_graph=True)
    Hv = R_op(df, params, v)
    return tuple(j.detach() for j in Hv) if detach else Hv


def ggn_vector_product_from_plist(loss, output, plist, v):
    """``J^T H_L J v`` with ``J = d output / d plist``, ``H_L = d^2 loss / d output^2``."""
    Jv = R_op(output, plist, v)
    HJv = hessian_vector_product(loss, output, Jv)
    return L_op(output, plist, HJv)


def ggn_vector_product(loss, output, model, v):
    return ggn_vector_product_from_plist(
        loss, output, [p for p in model.parameters() if p.requires_grad], v
    )


# --- SumGradSquared stand-in (definition-level restatement) ------------------
class SumGradSquared:
    """Marker object; the work happens in ``backpack.__exit__`` below."""


class _ExtendedLoss(torch.nn.Module):
    def __init__(self, inner):
        super().__init__()
        self.inner = inner
        self.last = None

    def forward(self, outputs, targets):
        self.last = (outputs, targets)
        return self.inner(outputs, targets)


_REGISTRY = {"model": None, "loss": None, "inputs": None}


def extend(module):
    """BackPACK's ``extend`` returns the module itself (with hooks).  Here the
    model is remembered and a forward pre-hook records its input; the loss
    module is wrapped so that (outputs, targets) are remembered."""
    is_loss = isinstance(module, torch.nn.modules.loss._Loss)
    if is_loss:
        wrapped = _ExtendedLoss(module)
        _REGISTRY["loss"] = wrapped
        return wrapped
    _REGISTRY["model"] = module

    def _remember(mod, args):
        _REGISTRY["inputs"] = args[0]

    if not getattr(module, "_oracle_hooked", False):
        module.register_forward_pre_hook(_remember)
        module._oracle_hooked = True
    return module


class backpack:
    """Context manager: on exit (after ``loss.backward()`` ran inside it) attach
    ``sum_grad_squared`` to every trainable parameter.

    BackPACK's quantity for a ``mean``-reduced loss is
    ``sum_i (d (l_i / N) / d theta)^2`` -- i.e. ``1/N^2`` times the sum of the
    squared per-sample gradients (this is what preconditioners.py:56-58
    compensates with ``* N``)."""

    def __init__(self, *extensions):
        self.extensions = extensions

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        model, lossw, inputs = (
            _REGISTRY["model"],
            _REGISTRY["loss"],
            _REGISTRY["inputs"],
        )
        _, targets = lossw.last
        n = inputs.shape[0]
        reduction = getattr(lossw.inner, "reduction", "mean")
        params = [p for p in model.parameters() if p.requires_grad]
        acc = [torch.zeros_like(p) for p in params]
        for i in range(n):
            li = lossw.inner(model(inputs[i : i + 1]), targets[i : i + 1])
            gi = torch.autograd.grad(li, params)
            scale = (1.0 / n) if reduction == "mean" else 1.0
            for a, g in zip(acc, gi):
                a += (scale * g) ** 2
        for p, a in zip(params, acc):
            p.sum_grad_squared = a
        return False


def install_as_backpack():
    """Register this restatement under ``backpack``, ``backpack.hessianfree.*``
    and ``backpack.extensions`` (build container only; see module docstring)."""
    me = sys.modules[__name__]
    root = types.ModuleType("backpack")
    root.backpack = backpack
    root.extend = extend
    hf = types.ModuleType("backpack.hessianfree")
    ggnvp = types.ModuleType("backpack.hessianfree.ggnvp")
    ggnvp.ggn_vector_product_from_plist = ggn_vector_product_from_plist
    ggnvp.ggn_vector_product = ggn_vector_product
    hvp = types.ModuleType("backpack.hessianfree.hvp")
    hvp.hessian_vector_product = hessian_vector_product
    rop = types.ModuleType("backpack.hessianfree.rop")
    rop.R_op = R_op
    lop = types.ModuleType("backpack.hessianfree.lop")
    lop.L_op = L_op
    ext = types.ModuleType("backpack.extensions")
    ext.SumGradSquared = SumGradSquared
    root.hessianfree = hf
    root.extensions = ext
    hf.ggnvp, hf.hvp, hf.rop, hf.lop = ggnvp, hvp, rop, lop
    for name, mod in [
        ("backpack", root),
        ("backpack.hessianfree", hf),
        ("backpack.hessianfree.ggnvp", ggnvp),
        ("backpack.hessianfree.hvp", hvp),
        ("backpack.hessianfree.rop", rop),
        ("backpack.hessianfree.lop", lop),
        ("backpack.extensions", ext),
    ]:
        sys.modules[name] = mod
    return me
