"""Armijo back-off on the step length (host control; behaviour of the reference's
``hessianfree/linesearch.py:8-103``, Martens & Sutskever 2012, section 8.8).

Each trial costs one no-grad forward pass on parameters written by the fused
``theta = theta0 + alpha*step`` kernel (``f.scaled``), so nothing of size N is
allocated or copied per trial."""

from warnings import warn

import torch


def _evaluator(f, step):
    """``alpha -> f(alpha*step)``; uses the optimizer's fused form when offered."""
    fused = getattr(f, "scaled", None)
    if fused is not None:
        return lambda alpha: fused(step, alpha)
    return lambda alpha: f(torch.zeros_like(step)) if alpha == 0.0 else f(alpha * step)


def simple_linesearch(f, f_grad_0, step, init_alpha=1.0, beta=0.8, c=1e-2, max_iter=20,
                      verbose=False):
    """Return ``(alpha, f(alpha*step))`` for the first ``alpha`` in
    ``init_alpha * beta**k`` (``k < max_iter``) with
    ``f(alpha*step) <= f(0) + alpha * c * f_grad_0 . step``; if none qualifies,
    warn and return ``(0.0, f(0))``.  ``f`` maps a step vector to a float."""
    if beta >= 1.0:
        raise ValueError(f"Invalid reduction factor beta = {beta}")
    if c < 0.0:
        raise ValueError(f"Invalid c = {c}")
    say = print if verbose else (lambda *a: None)
    value_at = _evaluator(f, step)
    prefetch = getattr(f, "prefetch", None)  # graph-replayed evaluation: enqueue without a host sync
    if prefetch is not None:
        prefetch([(step, 0.0), (step, init_alpha)])

    say("\nStarting line search...")
    base = float(value_at(0.0))
    say(f"  f(0) = {base:.6f}")
    trial = float(value_at(init_alpha))
    say(f"  f(init_alpha * step) = {trial:.6f}")

    slope = c * torch.dot(f_grad_0, step).item()  # c * directional derivative
    if slope >= 0:
        warn("`update_vec`-parameter in `simple_linesearch` is not a descent "
             f"direction. The directional derivative is {slope:.6f}.")

    alpha, tries = init_alpha, 0
    while tries < max_iter:
        say(f"  Trying alpha = {alpha:.6f}, f(alpha * step) = {trial:.6f}")
        if float(trial) <= base + alpha * slope:
            say(f"Significant improvement for alpha = {alpha:.6f}")
            return alpha, trial
        alpha *= beta
        if prefetch is not None:  # this candidate and the next: one read-back for two values
            prefetch([(step, alpha), (step, alpha * beta)], needed=1)  # (the second: speculative)
        trial = value_at(alpha)
        tries += 1

    warn("No suitable update could be found by the line search.")
    say(f"No significant improvement. Using alpha = {0.0:.6f}")
    return 0.0, base
