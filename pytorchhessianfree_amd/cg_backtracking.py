"""CG-backtracking over the stored PCG iterates (reference
``hessianfree/cg_backtracking.py:6-112``): host control flow around a handful of
no-grad forward passes; the iterates themselves are the snapshot slab the PCG
update kernel wrote."""

import torch


def cg_backtracking(f, steps_list, verbose=False):
    """Exhaustive variant (cg_backtracking.py:6-50): evaluate every stored step,
    return ``(index of the smallest value, that value)``."""
    if verbose:
        print("\nBacktracking cg-iterations...")
    values = [float("inf") if s is None else f(s) for s in steps_list]
    best = torch.argmin(torch.Tensor(values))
    if verbose:
        for i, val in enumerate(values):
            if steps_list[i] is not None:
                mark = "* " if i == best else "  "
                print(f"{mark}cg-iteration {i}, loss = {val:.6f}")
    return best, values[best]


def cg_efficient_backtracking(f, steps_list, verbose=False):
    """Martens' variant (cg_backtracking.py:53-112): walk the stored steps from
    the last one backwards and stop at the first that does not improve on the best
    value seen; returns ``(index, value)`` of the best OBSERVED step."""
    if verbose:
        print("\nBacktracking cg-iterations...")
    seen = {}
    best_val, best_idx = float("inf"), None
    # ``f.prefetch`` (the optimizer's graph-replayed evaluation): candidates are enqueued two at a
    # time, so a pair of loss values costs ONE device->host read; the early exit below is unchanged
    # (at most one evaluation is wasted when the walk stops)
    prefetch = getattr(f, "prefetch", None)
    order = [idx for idx in range(len(steps_list) - 1, -1, -1) if steps_list[idx] is not None]
    for pos, idx in enumerate(order):
        if prefetch is not None:
            prefetch([(steps_list[i], 1.0) for i in order[pos:pos + 2]], needed=1)  # (the second: speculative)
        val = f(steps_list[idx])
        seen[idx] = val
        if val < best_val:
            best_val, best_idx = val, idx
        else:
            break
    if verbose:
        for idx, step in enumerate(steps_list):
            if step is None:
                continue
            if idx not in seen:
                print(f"  cg-iteration {idx}, loss not evaluated")
            else:
                mark = "* " if idx == best_idx else "  "
                print(f"{mark}cg-iteration {idx}, loss = {seen[idx]:.6f}")
    return best_idx, seen[best_idx]
