"""Armijo back-off line search (reference ``hessianfree/linesearch.py:8-103``)."""

from warnings import warn

import torch


def simple_linesearch(f, f_grad_0, step, init_alpha=1.0, beta=0.8, c=1e-2, max_iter=20,
                      verbose=False):
    """Shrink ``alpha`` by ``beta`` until ``f(alpha*step) <= f(0) + alpha*c*g^T step``
    (at most ``max_iter`` trials).  Returns ``(alpha, f(alpha*step))`` or
    ``(0.0, f(0))`` with a warning when no trial passes (linesearch.py:99-103).

    If ``f`` offers ``f.scaled(step, alpha)`` (the optimizer's target function
    does: one fused ``theta = theta0 + alpha*step`` kernel), that is used instead
    of materialising ``alpha * step``; the arithmetic is the same.
    """
    if beta >= 1.0:
        raise ValueError(f"Invalid reduction factor beta = {beta}")
    if c < 0.0:
        raise ValueError(f"Invalid c = {c}")
    scaled = getattr(f, "scaled", None)

    def at(alpha):
        if scaled is not None:
            return scaled(step, alpha)
        return f(alpha * step)

    if verbose:
        print("\nStarting line search...")
    f_0 = float(scaled(step, 0.0) if scaled is not None else f(torch.zeros_like(step)))
    f_trial = float(at(init_alpha))
    if verbose:
        print(f"  f(0) = {f_0:.6f}")
        print(f"  f(init_alpha * step) = {f_trial:.6f}")

    slope = c * torch.dot(f_grad_0, step).item()
    if slope >= 0:
        msg = "`update_vec`-parameter in `simple_linesearch` is not a descent "
        msg += f"direction. The directional derivative is {slope:.6f}."
        warn(msg)

    alpha = init_alpha
    for _ in range(max_iter):
        if verbose:
            print(f"  Trying alpha = {alpha:.6f}, f(alpha * step) = {f_trial:.6f}")
        if float(f_trial) <= f_0 + alpha * slope:
            if verbose:
                print(f"Significant improvement for alpha = {alpha:.6f}")
            return alpha, f_trial
        alpha *= beta
        f_trial = at(alpha)

    warn("No suitable update could be found by the line search.")
    if verbose:
        print(f"No significant improvement. Using alpha = {0.0:.6f}")
    return 0.0, f_0
