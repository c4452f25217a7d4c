"""GPU parity tests of the HIP PCG path (``pytorchhessianfree_amd.cg.cg`` ->
``libhfpcg.so``) against the oracle and the reference's golden vectors.

Stated tolerances (fp32 unless noted):
  * versus the oracle run in the kernels' own reduction arithmetic
    (``accumulate="fp64"``, matvec rounded from fp64 on both sides): iterates
    ``rtol 2e-5`` of the iterate's max-norm, ``m_k`` ``rtol 1e-5``, identical
    termination reason and iteration count;
  * versus the REFERENCE's golden vectors on its ill-conditioned dense test
    systems (cond ~1e5, where the reference itself moves by percents under a
    change of summation order -- see DESIGN.md "Parity"): first three iterates
    ``rtol 1e-4``, same reason, iteration count within +-2, same residual bound
    as tests/test_cg.py:87;
  * versus the reference's golden vectors on damped (well-conditioned) systems
    of the kind ``HessianFree.step`` produces: every stored iterate ``rtol 1e-4``,
    ``m_k`` ``rtol 1e-5``, identical reason, iteration count +-1;
  * float64: every iterate ``rtol 1e-9``.
"""

import warnings

import numpy as np
import pytest
import torch
from tol import within

from conftest import load_golden
from helpers import T, lowrank_operator

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _product():
    import pytorchhessianfree_amd as product

    return product


def _oracle():
    from oracle import pcg as oracle

    return oracle


def _maxrel(a, b):
    scale = max(float(np.abs(b).max()), 1e-30)
    return float(np.abs(a - b).max()) / scale


def _linear_keys():
    g = load_golden("cg_linear.npz")
    return [str(k) for k in g["index"]]


def _mv64(A64):
    """Matvec whose result is the fp64 product rounded once to fp32: the same on
    CPU and GPU up to fp64 summation order."""

    def f(v):
        return (A64 @ v.double()).to(v.dtype)

    return f


@pytest.mark.parametrize("key", _linear_keys())
def test_linear_systems_match_kernel_arithmetic_oracle(key):
    g = load_golden("cg_linear.npz")
    product, oracle = _product(), _oracle()
    A, b = T(g[key + "/A"]), T(g[key + "/b"])
    dim = A.shape[0]
    x0 = T(g[key + "/x0"]) if key + "/x0" in g else None
    minv = T(g[key + "/minv"]) if key + "/minv" in g else None
    martens = key.endswith("m1")
    kw = dict(max_iter=10 * dim, tol=1e-5, atol=1e-6, martens_conv_crit=martens,
              store_x_at_iters=list(range(10 * dim)))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ox, om, oreason = oracle.pcg(
            _mv64(A.double()), b, x0=x0, M=(lambda v: minv * v) if minv is not None else None,
            accumulate="fp64", **kw)
        Ad = A.double().to(DEV)
        M = None
        if minv is not None:
            md = minv.to(DEV)
            M = lambda v: md * v  # noqa: E731  (generic callable -> HF_M_EXTERNAL path)
        gx, gm, greason = product.cg(
            _mv64(Ad), b.to(DEV), x0=None if x0 is None else x0.to(DEV), M=M, **kw)
    assert greason == oreason
    assert len(gx) == len(ox)
    for i, (a, o) in enumerate(zip(gx, ox)):
        assert (a is None) == (o is None)
        if a is not None:
            assert _maxrel(a.cpu().numpy(), o.numpy()) < 2e-5, f"iterate {i}"
    if martens:
        gm = np.array([float(m) for m in gm])
        om = np.array([float(m) for m in om])
        np.testing.assert_allclose(gm, om, rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("key", _linear_keys())
def test_linear_systems_within_reference_envelope(key):
    g = load_golden("cg_linear.npz")
    product = _product()
    A, b = T(g[key + "/A"], DEV), T(g[key + "/b"], DEV)
    dim = A.shape[0]
    x0 = T(g[key + "/x0"], DEV) if key + "/x0" in g else None
    minv = T(g[key + "/minv"], DEV) if key + "/minv" in g else None
    M = product.DiagonalPreconditioner(1.0 / minv, 0.0, 1.0) if minv is not None else None
    martens = key.endswith("m1")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        gx, gm, greason = product.cg(
            lambda v: A @ v, b, x0=x0, M=M, max_iter=10 * dim, tol=1e-5, atol=1e-6,
            martens_conv_crit=martens, store_x_at_iters=list(range(10 * dim)))
    X = g[key + "/X"]
    assert greason == str(g[key + "/reason"])
    assert abs((len(gx) - 1) - (X.shape[0] - 1)) <= 2
    for i in range(min(4, len(gx), X.shape[0])):
        assert _maxrel(gx[i].cpu().numpy(), X[i]) < 1e-4, f"iterate {i}"
    if not martens:  # ran to the tolerance: the reference's own check (test_cg.py:87)
        res = torch.linalg.norm(A @ gx[-1] - b)
        assert res <= max(1e-5 * torch.linalg.norm(b), 1e-6) + 5e-6


def test_float64_matches_reference_golden():
    """float64 (tests/test_cg.py:177-178).  dim 50 has cond ~1e6: even float64 CG
    trajectories separate after ~15 iterations under a change of summation order,
    so late iterates are checked through the residual bound."""
    g = load_golden("cg_f64.npz")
    product = _product()
    for key in [str(k) for k in g["index"]]:
        A, b = T(g[key + "/A"], DEV), T(g[key + "/b"], DEV)
        dim = A.shape[0]
        gx, _, reason = product.cg(
            lambda v: A @ v, b, max_iter=10 * dim, tol=1e-5, atol=1e-6,
            store_x_at_iters=list(range(10 * dim)))
        X = g[key + "/X"]
        assert reason == str(g[key + "/reason"])
        assert abs(len(gx) - X.shape[0]) <= 2, key
        for i in range(min(len(gx), X.shape[0], 8)):
            assert _maxrel(gx[i].cpu().numpy(), X[i]) < 1e-9, (key, i)
        res = torch.linalg.norm(A @ gx[-1] - b)
        assert res <= max(1e-5 * torch.linalg.norm(b), 1e-6) + 5e-6


def _lowrank_keys():
    g = load_golden("cg_lowrank.npz")
    return [str(k) for k in g["index"]]


def _relnorm(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("key", _lowrank_keys())
def test_damped_lowrank_matches_reference_golden(key, fused):
    """The shape of the call in optimizer.py:265-274: damped operator, diag
    preconditioner, Martens test, automatic snapshot grid.

    These systems (cond 5e2..5e5) lose CG orthogonality in fp32: the reference's
    own fp32 run departs from the reference's float64 run (golden ``X64``) by up
    to 2e-1 after ~13 iterations and stops up to 15 iterations later.  Stated
    tolerance: early iterates (k <= 4) within 1e-4 of the reference; every stored
    iterate no farther from the float64 trajectory than 4x the reference's own
    fp32 distance (+1e-5), where the reference's distance is taken as its running
    maximum up to the next snapshot (the separation is exponential once it starts,
    so its onset may shift by an iteration or two); the same for m_k (+1e-6 |m|,
    window 3 iterations); same reason; iteration count within the reference's own
    fp32-vs-fp64 spread."""
    g = load_golden("cg_lowrank.npz")
    product = _product()
    A, B, damping = lowrank_operator(g, key, DEV)
    b = T(g[key + "/b"], DEV)
    x0 = T(g[key + "/x0"], DEV) if key + "/x0" in g else None
    M = None
    if int(g[key + "/precond"]):
        M = product.DiagonalPreconditioner(T(g[key + "/diag"], DEV), damping)
        if not fused:
            Mobj = M
            M = lambda v: Mobj(v)  # noqa: E731
    op = product.DampedCurvature(B, damping) if fused else A
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        gx, gm, reason = product.cg(op, b, x0=x0, M=M, max_iter=250,
                                    martens_conv_crit=True, store_x_at_iters=None)
    X, X64 = g[key + "/X"], g[key + "/X64"]
    m_ref, m64 = g[key + "/m"], g[key + "/m64"]
    assert reason == str(g[key + "/reason"])
    n_gpu, n_ref, n_64 = len(gx) - 1, X.shape[0] - 1, X64.shape[0] - 1
    within(abs(n_gpu - n_ref), max(1, abs(n_ref - n_64)), strict=False)
    grid = product.storing_grid(250)
    for i in range(n_gpu):
        assert (gx[i] is not None) == (i in grid), i
    last = min(n_gpu, n_ref, n_64)
    common = [i for i in range(last + 1)
              if gx[i] is not None and not np.isnan(X[i]).any() and not np.isnan(X64[i]).any()]
    e_ref = {i: _relnorm(X[i].astype(np.float64), X64[i]) for i in common}
    for pos, i in enumerate(common):
        xg = gx[i].cpu().numpy().astype(np.float64)
        e_gpu = _relnorm(xg, X64[i])
        # the onset of the fp32/fp64 separation may come one snapshot earlier
        window = max(e_ref[j] for j in common[: pos + 2])
        within(e_gpu, 4.0 * window + 1e-5, strict=False, note=(key, i, e_gpu, window))  # (<= 1.0 x window measured)
        if i <= 4:
            assert _relnorm(xg, X[i].astype(np.float64)) < 1e-4, (key, i)
    dm_ref = np.abs(m_ref[: last + 1].astype(np.float64) - m64[: last + 1])
    for i in range(last + 1):
        dm_gpu = abs(float(gm[i]) - m64[i])
        window = dm_ref[: min(i + 4, last + 1)].max()  # same: up to 3 iterations earlier
        within(dm_gpu, 4.0 * window + 1e-6 * abs(m64[i]) + 1e-7, strict=False, note=(key, i, dm_gpu, window))


@pytest.mark.parametrize("key", _lowrank_keys())
def test_damped_lowrank_matches_kernel_arithmetic_oracle(key):
    """Same systems, compared tightly against the oracle run in the kernels'
    reduction arithmetic with an order-independent (fp64-rounded) matvec."""
    g = load_golden("cg_lowrank.npz")
    product, oracle = _product(), _oracle()
    U, d, b = T(g[key + "/U"]), T(g[key + "/d"]), T(g[key + "/b"])
    damping = float(g[key + "/damping"])
    x0 = T(g[key + "/x0"]) if key + "/x0" in g else None
    diag = T(g[key + "/diag"])
    precond = int(g[key + "/precond"])

    def make_B(U64, d64):
        def B(v):
            v64 = v.double()
            return (d64 * v64 + U64 @ (U64.T @ v64)).to(v.dtype)
        return B

    B_cpu = make_B(U.double(), d.double())
    B_gpu = make_B(U.double().to(DEV), d.double().to(DEV))
    # the power (diag+lambda)^-0.75 is evaluated once on the GPU and shared, so
    # that libm-vs-device pow differences do not enter this comparison
    Mg = product.DiagonalPreconditioner(diag.to(DEV), damping) if precond else None
    minv_cpu = Mg.minv.cpu() if precond else None
    kw = dict(max_iter=250, martens_conv_crit=True, store_x_at_iters=None)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ox, om, oreason = oracle.pcg(
            lambda v: B_cpu(v) + damping * v, b, x0=x0,
            M=(lambda v: minv_cpu * v) if precond else None, accumulate="fp64", **kw)
        gx, gm, greason = product.cg(
            product.DampedCurvature(B_gpu, damping), b.to(DEV),
            x0=None if x0 is None else x0.to(DEV), M=Mg, **kw)
    assert greason == oreason and len(gx) == len(ox)
    for i, (a, o) in enumerate(zip(gx, ox)):
        assert (a is None) == (o is None)
        if a is not None:
            assert _maxrel(a.cpu().numpy(), o.numpy()) < 2e-5, (key, i)
    np.testing.assert_allclose(np.array([float(m) for m in gm]),
                               np.array([float(m) for m in om]), rtol=1e-5, atol=1e-6)


# ---- the reference's own property tests, on the GPU path ----------------------
def _spd(dim, seed):
    g = torch.Generator().manual_seed(seed)
    A = torch.rand((dim, dim), generator=g) - 0.5
    A = A @ A.T + 1e-3 * torch.eye(dim)
    x = torch.rand(dim, generator=g) - 0.5
    return A, A @ x, x


@pytest.mark.parametrize("dim", [3, 10, 50])
@pytest.mark.parametrize("tol,atol", [(1e-3, 1e-3), (1e-6, 1e-6), (1e-3, 1e-6)])
@pytest.mark.parametrize("precond", [False, True])
def test_residual_within_tolerance(dim, tol, atol, precond):
    """tests/test_cg.py:34-87."""
    product = _product()
    A, b, _ = _spd(dim, 5)
    A, b = A.to(DEV), b.to(DEV)
    M = product.DiagonalPreconditioner(torch.diag(A).clone(), 0.0, 1.0) if precond else None
    xs, _, _ = product.cg(lambda v: A @ v, b, M=M, max_iter=10 * dim, tol=tol, atol=atol)
    assert len([x for x in xs if x is not None]) == 1  # only the final iterate
    res = torch.linalg.norm(A @ xs[-1] - b)
    assert res <= max(tol * torch.linalg.norm(b), atol) + 5e-6


@pytest.mark.parametrize("dim", [3, 10, 50])
@pytest.mark.parametrize("precond", [False, True])
@pytest.mark.parametrize("warm", [False, True])
def test_m_iters_are_the_quadratic(dim, precond, warm):
    """tests/test_cg.py:98-156: m_i == 0.5 x_i^T A x_i - b^T x_i."""
    product = _product()
    A, b, _ = _spd(dim, 11)
    x0 = (2 * (torch.rand(dim, generator=torch.Generator().manual_seed(3)) - 0.5)) if warm else None
    Ad, bd = A.to(DEV), b.to(DEV)
    M = product.DiagonalPreconditioner(torch.diag(Ad).clone(), 0.0, 1.0) if precond else None
    xs, ms, _ = product.cg(lambda v: Ad @ v, bd, x0=None if x0 is None else x0.to(DEV), M=M,
                           max_iter=10 * dim, tol=1e-5, atol=1e-6, martens_conv_crit=True,
                           store_x_at_iters=list(range(10 * dim)))
    assert len(ms) == len(xs)
    A64, b64 = A.double(), b.double()
    for x, m in zip(xs, ms):
        x64 = x.cpu().double()
        q = 0.5 * x64 @ (A64 @ x64) - b64 @ x64
        # the reference asserts atol=1e-7 on CPU for |m| = O(1e-1..1); the GPU
        # value carries fp32 rounding of r and x: 3e-6 absolute (9.3e-7 the worst over the round's leases)
        within(abs(float(m) - float(q)), 3e-6)


@pytest.mark.parametrize("dim", [3, 10, 50])
def test_identity_preconditioner_is_bitwise_no_preconditioner(dim):
    """tests/test_cg.py:212-218 (float64 as in the reference)."""
    product = _product()
    A, b, _ = _spd(dim, 42)
    A, b = A.double().to(DEV), b.double().to(DEV)
    kw = dict(max_iter=10 * dim, tol=1e-5, atol=1e-6, store_x_at_iters=list(range(10 * dim)))
    x_none, _, _ = product.cg(lambda v: A @ v, b, M=None, **kw)
    x_id, _, _ = product.cg(lambda v: A @ v, b, M=lambda v: v, **kw)
    assert len(x_none) == len(x_id)
    for a, c in zip(x_none, x_id):
        assert torch.equal(a, c)


@pytest.mark.parametrize("dim", [3, 10, 50])
def test_exact_inverse_preconditioner_converges_in_one_iteration(dim):
    """tests/test_cg.py:220-224."""
    product = _product()
    A, b, _ = _spd(dim, 1)
    A, b = A.double().to(DEV), b.double().to(DEV)
    Ainv = torch.linalg.inv(A)
    xs, _, _ = product.cg(lambda v: A @ v, b, M=lambda v: Ainv @ v, max_iter=10 * dim,
                          tol=1e-5, atol=1e-6, store_x_at_iters=list(range(10 * dim)))
    assert len(xs) - 1 <= 1


def test_snapshot_pattern_and_final_iterate():
    """cg.py:181-187, :209-210, :229-230: which entries are tensors."""
    product = _product()
    n = 1003
    d = torch.linspace(1.0, 50.0, n, device=DEV)
    b = torch.ones(n, device=DEV)
    xs, ms, reason = product.cg(lambda v: d * v, b, max_iter=37, tol=0.0,
                                martens_conv_crit=False, store_x_at_iters=None)
    assert reason == "Number of iterations" and len(xs) == 38
    grid = product.storing_grid(37)
    for i, x in enumerate(xs[:-1]):
        assert (x is not None) == (i in grid)
    assert xs[-1] is not None and ms is None
    xs2, _, _ = product.cg(lambda v: d * v, b, max_iter=37, tol=0.0, store_x_at_iters=[0])
    assert xs2[0] is not None and all(x is None for x in xs2[1:-1])
    assert torch.equal(xs2[-1], xs[-1])


def test_nonpositive_curvature_warns_and_nan_diverges():
    product = _product()
    n = 64
    d = -torch.ones(n, device=DEV)
    b = torch.ones(n, device=DEV)
    with pytest.warns(UserWarning, match="Directional curvature pAp"):
        product.cg(lambda v: d * v, b, max_iter=2, tol=0.0)
    nan = torch.full((n,), float("nan"), device=DEV)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        _, _, reason = product.cg(lambda v: nan, b, max_iter=5)
    assert reason == "Divergence"


def test_cpu_tensors_are_refused():
    product = _product()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        product.cg(lambda v: v, torch.ones(4))


@pytest.mark.parametrize("n", [11_175_370, 25_557_032])
def test_full_size_vectors_properties(n):
    """BASELINE.json sizes (ResNet-18 / ResNet-50 parameter counts), checked
    through size-independent properties: on a diagonal SPD operator the exact
    solution is b/(d+lambda); m_k decreases monotonically; the snapshot at the
    last grid iteration equals what a shorter run returns as its final iterate
    (bitwise: the kernels are deterministic); linearity in b."""
    product = _product()
    gen = torch.Generator(device=DEV).manual_seed(0)
    d = torch.rand(n, device=DEV, generator=gen) * 4.0
    b = torch.randn(n, device=DEV, generator=gen)
    lam = 0.5
    op = product.DampedCurvature(lambda v: d * v, lam)
    M = product.DiagonalPreconditioner(d * 0.9, lam, 0.75)
    xs, ms, reason = product.cg(op, b, M=M, max_iter=40, martens_conv_crit=True,
                                store_x_at_iters=None)
    exact = b / (d + lam)
    err = torch.linalg.norm(xs[-1] - exact) / torch.linalg.norm(exact)
    within(err, 1e-4, note=(float(err), reason))
    m = torch.stack(ms).cpu()
    assert bool((m[1:] <= m[:-1] + 1e-3 * m.abs().max()).all())
    k = max(i for i in range(len(xs) - 1) if xs[i] is not None)
    if k >= 1:
        xs_short, _, _ = product.cg(op, b, M=M, max_iter=k, tol=0.0, store_x_at_iters=[])
        assert torch.equal(xs_short[-1], xs[k])
    xs2, _, _ = product.cg(op, 2.0 * b, M=M, max_iter=len(xs) - 1, tol=0.0)
    rel = torch.linalg.norm(xs2[-1] - 2.0 * xs[-1]) / torch.linalg.norm(xs2[-1])
    within(rel, 1e-5)


def test_lockstep_termination_rule_gives_identical_results():
    """Operators that end in a collective switch cg() to the deterministic lagged
    stop rule (every rank performs n_iters + LAG operator calls); results must be
    those of the opportunistic poll."""
    product = _product()
    n = 5003
    gen = torch.Generator(device=DEV).manual_seed(0)
    d = torch.rand(n, device=DEV, generator=gen) * 20 + 0.1
    b = torch.randn(n, device=DEV, generator=gen)

    class Op:
        def __init__(self, collective):
            self.collective = collective
            self.calls = 0

        def __call__(self, v):
            self.calls += 1
            return d * v

    outs = []
    for coll in (False, True):
        op = Op(coll)
        xs, ms, reason = product.cg(product.DampedCurvature(op, 0.1), b, max_iter=250,
                                    martens_conv_crit=True, store_x_at_iters=None)
        outs.append((xs, ms, reason, op.calls))
    (x0, m0, r0, c0), (x1, m1, r1, c1) = outs
    assert r0 == r1 == "Convergence (Martens)" and len(x0) == len(x1)
    for a, c in zip(x0, x1):
        assert (a is None) == (c is None)
        if a is not None:
            assert torch.equal(a, c)
    assert torch.equal(torch.stack(m0), torch.stack(m1))
    assert c1 == 1 + (len(x1) - 1) + 2  # A(x0) + n_iters + LAG speculative (no-op) iterations
    # the opportunistic poll sees the device's flag within a few speculative launches
    assert 1 + (len(x0) - 1) <= c0 <= 1 + (len(x0) - 1) + 8


@pytest.mark.parametrize("n", [1, 2, 3, 5, 255, 256, 257, 1023])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_tiny_and_ragged_lengths(n, dtype):
    """Vector lengths below / across the 16-byte vector width and the block tile
    (tail handling in every kernel), incl. max_iter = 1 and out-of-range snapshot
    requests, against the oracle in the kernels' arithmetic."""
    product, oracle = _product(), _oracle()
    g = torch.Generator().manual_seed(n)
    d = (torch.rand(n, generator=g) * 3 + 0.5).to(dtype)
    b = torch.randn(n, generator=g).to(dtype)
    diag = torch.rand(n, generator=g).to(dtype)
    dd, bd = d.to(DEV), b.to(DEV)
    Mg = product.DiagonalPreconditioner(diag.to(DEV), 0.3)
    minv = Mg.minv.cpu()
    for max_iter in (1, 7):
        kw = dict(max_iter=max_iter, tol=1e-6, martens_conv_crit=True, store_x_at_iters=[0, 1, 5, 99])
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ox, om, oreason = oracle.pcg(lambda v: d * v + 0.3 * v, b, M=lambda v: minv * v,
                                         accumulate="fp64", **kw)
            gx, gm, greason = product.cg(product.DampedCurvature(lambda v: dd * v, 0.3), bd, M=Mg, **kw)
        assert greason == oreason and len(gx) == len(ox)
        tol = 1e-5 if dtype == torch.float32 else 1e-12
        for a, o in zip(gx, ox):
            assert (a is None) == (o is None)
            if a is not None:
                assert _maxrel(a.cpu().numpy(), o.numpy()) < tol
        assert len(gm) == len(om)


@pytest.mark.parametrize("blocks", ["64", "1000"])
def test_kernel_grid_override_gives_the_same_solve(monkeypatch, blocks):
    """``HF_PCG_BLOCKS`` (workgroups of K1-K3; default 2 per CU; must agree on all ranks of a data-parallel run): another
    grid is another partition of the fp64 partial sums -- the solve equals the oracle in the kernels' arithmetic within
    the same bounds as with the default grid (iterates 2e-5, same reason and iteration count)."""
    product, oracle = _product(), _oracle()
    monkeypatch.setenv("HF_PCG_BLOCKS", blocks)
    n = 1_300_021 + int(blocks)  # (a length no other test uses: the workspace -- and its grid -- is cached per length)
    g = torch.Generator().manual_seed(7)
    d = torch.rand(n, generator=g) * 3 + 0.5
    b = torch.randn(n, generator=g)
    dd, bd = d.to(DEV), b.to(DEV)
    kw = dict(max_iter=25, tol=1e-6, martens_conv_crit=True, store_x_at_iters=[0, 1, 2, 5, 10])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ox, om, oreason = oracle.pcg(lambda v: d * v + 0.3 * v, b, accumulate="fp64", **kw)
        gx, gm, greason = product.cg(product.DampedCurvature(lambda v: dd * v, 0.3), bd, **kw)
    assert greason == oreason and len(gx) == len(ox)
    for a, o in zip(gx, ox):
        assert (a is None) == (o is None)
        if a is not None:
            within(_maxrel(a.cpu().numpy(), o.numpy()), 2e-5)
    for a, o in zip(gm, om):
        within(abs(float(a) - float(o)), 1e-5 * abs(float(o)) + 1e-7, strict=False)


def test_c_abi_rejects_misuse_on_device():
    """Error behaviour of the C ABI itself (include/hf_pcg.h): call order, alignment,
    inconsistent arguments -- checked through ctypes with real device pointers."""
    import ctypes

    from pytorchhessianfree_amd import _lib

    lib = _lib.load()
    n = 1000
    h = _lib.c_void_p()
    assert lib.hf_pcg_create(ctypes.byref(h), n, 0, 0) == 0
    buf = torch.zeros(8 * n, device=DEV)
    P = lambda t: _lib.c_void_p(t.data_ptr())  # noqa: E731
    x, r, p, b, bp = (buf[i * n:(i + 1) * n] for i in range(5))
    s = _lib.current_stream_ptr(buf.device)
    assert lib.hf_pcg_iterate(h, P(bp), 0.0, s) == -3          # iterate before begin/init
    assert lib.hf_pcg_init(h, P(bp), s) == -3                  # init before begin
    # misaligned vector
    assert lib.hf_pcg_begin(h, P(buf[1:n + 1]), P(r), P(p), P(b), None, 0, 10, 1e-5, -1.0, 0,
                            None, 0, 0, None, 0, None) == -2
    # diag preconditioner without minv; Martens without m_hist; snapshots without slab
    assert lib.hf_pcg_begin(h, P(x), P(r), P(p), P(b), None, 1, 10, 1e-5, -1.0, 0, None, 0, 0, None, 0, None) == -1
    assert lib.hf_pcg_begin(h, P(x), P(r), P(p), P(b), None, 0, 10, 1e-5, -1.0, 1, None, 0, 0, None, 0, None) == -1
    assert lib.hf_pcg_begin(h, P(x), P(r), P(p), P(b), None, 0, 10, 1e-5, -1.0, 0, None, 2, 0, None, 0, None) == -1
    assert lib.hf_pcg_begin(h, P(x), P(r), P(p), P(b), None, 0, 0, 1e-5, -1.0, 0, None, 0, 0, None, 0, None) == -1
    # a valid sequence on the raw ABI: A = 2 I, b = 1  ->  x = 0.5 after one iteration
    b.fill_(1.0)
    assert lib.hf_pcg_begin(h, P(x), P(r), P(p), P(b), None, 0, 10, 1e-5, -1.0, 0, None, 0, 0, None, 0, None) == 0
    assert lib.hf_pcg_update_p(h, None, s) == -3               # phases before init
    bp.zero_()                                                 # A x0 = 0
    assert lib.hf_pcg_init(h, P(bp), s) == 0
    assert lib.hf_pcg_init_external(h, P(bp), s) == -3         # wrong mode
    torch.mul(p, 2.0, out=bp)
    assert lib.hf_pcg_iterate(h, P(bp), 0.0, s) == 0
    st = _lib.Status()
    assert lib.hf_pcg_finish(h, ctypes.byref(st), s) == 0
    assert st.done == 4 and st.n_iters == 1                    # "Convergence (tolerances)"
    assert torch.allclose(x, torch.full_like(x, 0.5))
    assert lib.hf_pcg_destroy(h) == 0
    assert lib.hf_pack(P(x), None, None, None, 1, 1.0, 0, 0, s) == -1
    assert lib.hf_axpy_out(P(x), P(x), None, 1.0, n, 0, s) == -1
    assert lib.hf_precond_build(P(x), P(b), 0.1, 0.75, n, 7, s) == -1


def test_randomised_configurations_match_kernel_arithmetic_oracle():
    """200 random combinations of length (incl. 1..5, block-tile edges), preconditioner
    mode (none / fused diagonal / generic callable), warm start, Martens on/off,
    max_iter, snapshot request, damping, tol/atol -- fp32, against the oracle in the
    kernels' reduction arithmetic: same reason, same iteration count, same None
    pattern, iterates 3e-5, m_k 1e-4.  (float64 runs agree to 1e-15 initially and then
    separate like any two float64 CG runs -- scripts/experiments/fuzz_one.py -- so they are not
    part of this bitwise-style check.)"""
    import random

    product, oracle = _product(), _oracle()
    rnd = random.Random(1234)
    for case in range(200):
        n = rnd.choice([1, 2, 3, 4, 5, 7, 63, 64, 65, 255, 256, 257, 511, 1000, 1023, 1025, 4097, 20011])
        mode = rnd.choice(["none", "diag", "ext"])
        warm, martens = rnd.random() < 0.5, rnd.random() < 0.6
        max_iter = rnd.choice([1, 2, 5, 17, 60])
        store = rnd.choice([[], [0], None, list(range(0, 61, 3)), [0, 1, 2, 1000]])
        lam, tol, atol = rnd.choice([0.0, 0.3, 2.0]), rnd.choice([0.0, 1e-5, 1e-2]), rnd.choice([None, 1e-6])
        g = torch.Generator().manual_seed(case)
        d = torch.rand(n, generator=g) * 10 + 0.05
        b = torch.randn(n, generator=g)
        x0 = torch.randn(n, generator=g) if warm else None
        diag = torch.rand(n, generator=g)
        dd = d.to(DEV)
        Mg = product.DiagonalPreconditioner(diag.to(DEV), lam if lam else 0.1) if mode != "none" else None
        minv = Mg.minv.cpu() if Mg is not None else None
        Mc = (lambda v: minv * v) if mode != "none" else None
        Mgg = Mg if mode == "diag" else ((lambda v: Mg.minv * v) if mode == "ext" else None)
        kw = dict(max_iter=max_iter, tol=tol, atol=atol, martens_conv_crit=martens, store_x_at_iters=store)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ox, om, orr = oracle.pcg(lambda v: (d.double() * v.double()).float() + lam * v, b, x0=x0,
                                     M=Mc, accumulate="fp64", **kw)
            gx, gm, grr = product.cg(
                product.DampedCurvature(lambda v: (dd.double() * v.double()).float(), lam), b.to(DEV),
                x0=None if x0 is None else x0.to(DEV), M=Mgg, **kw)
        tag = (case, n, mode, warm, martens, max_iter, lam, tol, atol)
        assert orr == grr and len(ox) == len(gx), tag
        for a, o in zip(gx, ox):
            assert (a is None) == (o is None), tag
            if a is not None and np.isfinite(float(o.abs().max())):
                assert _maxrel(a.cpu().numpy(), o.numpy()) < 3e-5, tag
        if martens:
            np.testing.assert_allclose(np.array([float(m) for m in gm]), np.array([float(m) for m in om]),
                                       rtol=1e-4, atol=1e-6, equal_nan=True, err_msg=str(tag))
