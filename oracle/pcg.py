"""TEST INFRASTRUCTURE ONLY -- CPU oracle of the preconditioned-CG hot loop.

A restatement (not a copy) of the algorithm of the reference's
``hessianfree/cg.py`` written as a small state machine.  Every arithmetic step
issues the same ATen op, on the same operands, in the same order as the
reference, so that on CPU the iterates are *bit-identical* to
``hessianfree.cg.cg`` (asserted by ``tests/golden/make_golden.py`` in the build
container, and by ``tests/test_oracle_golden.py`` against the committed golden
vectors everywhere).

Parity status: PINNED (see ``oracle/__init__.py``).

Reference lines restated here (all in ``/root/reference/hessianfree/cg.py``):
  * tolerance bound ............................ :75-76
  * termination tests and their order .......... :80-118
  * non-positive curvature warning ............. :123-147
  * snapshot grid ``ceil(1.3**j) - 1`` ......... :152-170
  * initialisation ............................. :177-192
  * loop body .................................. :200-227
  * final iterate always stored ................ :229-230
"""

import math
import warnings

import torch

REASON_MARTENS = "Convergence (Martens)"
REASON_MAXITER = "Number of iterations"
REASON_DIVERGED = "Divergence"
REASON_TOL = "Convergence (tolerances)"


def snapshot_grid(max_iter, gamma=1.3):
    """Iterations at which iterates are kept for CG-backtracking.

    Restates cg.py:152-170.  The powers are evaluated by torch on an *int64*
    ``arange`` with a Python-float base (-> float32 ``pow``), which is what the
    reference does; recomputing them in double gives a different table.
    """
    if gamma < 1.0:
        raise ValueError(f"Invalid gamma = {gamma}")
    top = math.ceil(math.log(max_iter + 1) / math.log(gamma))
    exponents = torch.arange(top + 1)
    marks = (torch.ceil(gamma**exponents) - 1).int().tolist()
    return sorted(set(marks))


def _dot64(a, b):
    """Dot product accumulated in float64 from exact products, rounded once to
    the operand dtype -- the arithmetic of the HIP kernels' reductions."""
    return (a.double() * b.double()).sum().to(a.dtype)


def _norm64(a):
    return (a.double() * a.double()).sum().sqrt().to(a.dtype)


class _Trace:
    """Optional per-iteration scalar trace (for debugging GPU parity)."""

    def __init__(self):
        self.alpha, self.beta, self.pAp, self.res_norm = [], [], [], []


def pcg(
    A,
    b,
    x0=None,
    M=None,
    max_iter=None,
    tol=1e-5,
    atol=None,
    martens_conv_crit=False,
    store_x_at_iters=(),
    verbose=False,
    trace=None,
    accumulate="reference",
):
    """Oracle PCG.  Same signature and return value as the reference ``cg``:
    ``(x_iters, m_iters, reason)``.  ``trace`` may be a ``_Trace`` to collect the
    scalars of every iteration.

    ``accumulate="reference"`` (default) uses ``torch.dot`` / ``torch.linalg.norm``
    exactly as the reference does and is the mode pinned bit-for-bit against it.
    ``accumulate="fp64"`` changes ONLY the precision in which the five reductions
    are accumulated (exact products summed in float64, rounded once): that is the
    arithmetic of the HIP kernels, so GPU iterates can be compared (nearly) bit
    for bit even on ill-conditioned systems where any change of summation order
    moves fp32 CG iterates by percents.
    """
    if accumulate == "reference":
        dot, norm = torch.dot, torch.linalg.norm
    elif accumulate == "fp64":
        dot, norm = _dot64, _norm64
    else:
        raise ValueError(accumulate)
    # ---- tolerance bound (cg.py:75-76) --------------------------------------
    bound = tol * norm(b).item()
    if atol is not None:
        bound = max([bound, atol])

    # ---- defaults (cg.py:177-183) -------------------------------------------
    if max_iter is None:
        max_iter = b.numel()
    start = torch.zeros_like(b) if x0 is None else x0
    if store_x_at_iters is None:
        store_x_at_iters = snapshot_grid(max_iter)
    keep = set(store_x_at_iters)

    # ---- initial state (cg.py:186-192) --------------------------------------
    x = start
    xs = [x if 0 in keep else None]
    r = A(start) - b
    ms = [0.5 * dot(r - b, start)] if martens_conv_crit else None
    y = r if M is None else M(r)
    ry = dot(r, y)
    p = -y

    k = 0
    reason = ""
    kept_last = False
    while True:
        k += 1
        # ---- curvature along p (cg.py:205-207, :133-139) --------------------
        Ap = A(p).detach()
        pAp = dot(p, Ap)
        if not (pAp > 0):
            warnings.warn(
                f"Directional curvature pAp = {pAp:.3e} <= 0 detected in cg-"
                f"iteration {k}. This is a violation to the assumption "
                "of positive definiteness."
            )
        alpha = ry / pAp

        # ---- iterate and residual update (cg.py:208-211) --------------------
        x = x + alpha * p
        kept_last = k in keep
        xs.append(x if kept_last else None)
        r = r + alpha * Ap

        # ---- termination tests, in the reference's order (cg.py:93-115) -----
        res_norm = norm(r)
        if trace is not None:
            trace.alpha.append(float(alpha))
            trace.pAp.append(float(pAp))
            trace.res_norm.append(float(res_norm))
        stop = False
        if martens_conv_crit:
            ms.append(0.5 * dot(r - b, x))
            lag = max(10, int(k / 10))
            if lag < k:
                gain = ms[k] - ms[k - lag]
                total = ms[k] - ms[0]
                if gain / total < 5e-4:
                    stop, reason = True, REASON_MARTENS
        if not stop and k >= max_iter:
            stop, reason = True, REASON_MAXITER
        if not stop and torch.isnan(res_norm):
            stop, reason = True, REASON_DIVERGED
        if not stop and res_norm < bound:
            stop, reason = True, REASON_TOL
        if stop:
            break

        # ---- new search direction (cg.py:220-224) ---------------------------
        y = r if M is None else M(r)
        ry_next = dot(r, y)
        beta = ry_next / ry
        ry = ry_next
        p = -y + beta * p
        if trace is not None:
            trace.beta.append(float(beta))

    if not kept_last:
        xs[-1] = x  # the final iterate is always returned (cg.py:229-230)
    if verbose:
        print(f"oracle pcg: {k} iterations, {reason}")
    return xs, ms, reason
