"""CPU: the oracle reproduces the REFERENCE's golden vectors bit for bit (this is
what "oracle pinned" means; the vectors were produced by the real
``hessianfree.cg.cg`` in ``tests/golden/make_golden.py``)."""

import warnings

import numpy as np
import pytest
import torch

from conftest import load_golden
from helpers import T, lowrank_operator
from oracle import pcg as oracle


@pytest.fixture(autouse=True)
def _one_thread():
    # the fixtures were generated single-threaded; torch.dot's blocking (and with
    # it the last bit) depends on the thread count
    n = torch.get_num_threads()
    torch.set_num_threads(1)
    yield
    torch.set_num_threads(n)


def _same(xs, X):
    assert len(xs) == X.shape[0]
    for i, x in enumerate(xs):
        if x is None:
            assert np.isnan(X[i]).all()
        else:
            assert np.array_equal(x.numpy(), X[i]), i


def test_dense_systems_bitwise():
    g = load_golden("cg_linear.npz")
    for key in [str(k) for k in g["index"]]:
        A, b = T(g[key + "/A"]), T(g[key + "/b"])
        dim = A.shape[0]
        x0 = T(g[key + "/x0"]) if key + "/x0" in g else None
        Mmat = torch.diag(T(g[key + "/minv"])) if key + "/minv" in g else None
        martens = key.endswith("m1")
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            xs, ms, reason = oracle.pcg(
                lambda v: A @ v, b, x0=x0, M=(lambda v: Mmat @ v) if Mmat is not None else None,
                max_iter=10 * dim, tol=1e-5, atol=1e-6, martens_conv_crit=martens,
                store_x_at_iters=list(range(10 * dim)))
        assert reason == str(g[key + "/reason"]), key
        _same(xs, g[key + "/X"])
        if martens:
            assert np.array_equal(np.array([float(m) for m in ms], dtype=np.float32), g[key + "/m"])


def test_float64_systems_bitwise():
    g = load_golden("cg_f64.npz")
    for key in [str(k) for k in g["index"]]:
        A, b = T(g[key + "/A"]), T(g[key + "/b"])
        dim = A.shape[0]
        xs, _, reason = oracle.pcg(lambda v: A @ v, b, max_iter=10 * dim, tol=1e-5, atol=1e-6,
                                   store_x_at_iters=list(range(10 * dim)))
        assert reason == str(g[key + "/reason"])
        _same(xs, g[key + "/X"])


def test_damped_lowrank_bitwise():
    g = load_golden("cg_lowrank.npz")
    for key in [str(k) for k in g["index"]]:
        A, _, damping = lowrank_operator(g, key, "cpu")
        b = T(g[key + "/b"])
        x0 = T(g[key + "/x0"]) if key + "/x0" in g else None
        M = None
        if int(g[key + "/precond"]):
            diag = T(g[key + "/diag"])
            M = lambda v: torch.mul((diag + damping) ** -0.75, v)  # noqa: E731
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            xs, ms, reason = oracle.pcg(A, b, x0=x0, M=M, max_iter=250, martens_conv_crit=True,
                                        store_x_at_iters=None)
        assert reason == str(g[key + "/reason"])
        _same(xs, g[key + "/X"])
        assert np.array_equal(np.array([float(m) for m in ms], dtype=np.float32), g[key + "/m"])


def test_snapshot_grid_table():
    g = load_golden("tables.npz")
    import pytorchhessianfree_amd as product

    for key in g.files:
        if key.startswith("grid/"):
            max_iter = int(key.split("/")[1])
            assert oracle.snapshot_grid(max_iter) == g[key].tolist()
            assert product.storing_grid(max_iter) == g[key].tolist()
    # the table the build plan quotes
    assert oracle.snapshot_grid(250) == [0, 1, 2, 3, 4, 6, 8, 10, 13, 17, 23, 30, 39, 51, 66, 86,
                                         112, 146, 190, 247, 321]


def test_fp64_accumulate_mode_stays_in_the_reference_envelope():
    """The oracle's ``accumulate="fp64"`` mode (the HIP kernels' reduction
    arithmetic) is no farther from the reference's float64 trajectory than the
    reference's own fp32 run (x2, +1e-5) on the damped systems."""
    g = load_golden("cg_lowrank.npz")
    for key in [str(k) for k in g["index"]]:
        A, _, damping = lowrank_operator(g, key, "cpu")
        b = T(g[key + "/b"])
        x0 = T(g[key + "/x0"]) if key + "/x0" in g else None
        M = None
        if int(g[key + "/precond"]):
            diag = T(g[key + "/diag"])
            M = lambda v: torch.mul((diag + damping) ** -0.75, v)  # noqa: E731
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            xs, _, reason = oracle.pcg(A, b, x0=x0, M=M, max_iter=250, martens_conv_crit=True,
                                       store_x_at_iters=None, accumulate="fp64")
        X, X64 = g[key + "/X"], g[key + "/X64"]
        assert reason == str(g[key + "/reason"])
        last = min(len(xs), X.shape[0], X64.shape[0]) - 1
        common = [i for i in range(last + 1) if xs[i] is not None
                  and not np.isnan(X[i]).any() and not np.isnan(X64[i]).any()]

        def rel(a, c):
            return float(np.linalg.norm(a - c) / max(np.linalg.norm(c), 1e-30))

        e_ref = {i: rel(X[i].astype(np.float64), X64[i]) for i in common}
        for pos, i in enumerate(common):
            window = max(e_ref[j] for j in common[: pos + 2])
            assert rel(xs[i].numpy().astype(np.float64), X64[i]) <= 2 * window + 1e-5, (key, i)
