"""CPU: the host layer (``HessianFree.step`` / ``acc_step`` orchestration, LM
damping, CG-backtracking, line search, curvature operators, preconditioner
recipe) against traces of the REAL reference (tests/golden/*.npz).

The PCG loop itself has no CPU implementation in the product; these tests plug
the CPU oracle into the optimizer's ``_cg`` hook so that everything AROUND the
kernels is checked here, and the same traces are replayed through the HIP
kernels in ``tests/test_optimizer_gpu.py``."""

import warnings

import numpy as np
import pytest
import torch

import pytorchhessianfree_amd as hf
from conftest import load_golden
from helpers import T, mwe_nn, small_nn, trainable_vec
from oracle import pcg as oracle
from pytorchhessianfree_amd import curvature


@pytest.fixture(autouse=True)
def _one_thread():
    n = torch.get_num_threads()
    torch.set_num_threads(1)
    yield
    torch.set_num_threads(n)


def make_opt(params, **kw):
    opt = hf.HessianFree(params, **kw)
    opt._cg = oracle.pcg  # CPU oracle in place of the HIP kernels (tests only)
    return opt


def check_state(opt, g, prefix, n_steps):
    st = opt.state
    np.testing.assert_allclose(st["init_losses"], g[prefix + "init_losses"][:n_steps], rtol=1e-6)
    np.testing.assert_allclose(st["dampings"], g[prefix + "dampings"][:n_steps], rtol=1e-12)
    assert list(st["cg_reasons"]) == [str(s) for s in g[prefix + "cg_reasons"][:n_steps]]
    assert list(st["num_cg_iters"]) == g[prefix + "num_cg_iters"][:n_steps].tolist()
    assert [int(i) for i in st["best_cg_iters"]] == g[prefix + "best_cg_iters"][:n_steps].tolist()
    np.testing.assert_allclose(st["learning_rates"], g[prefix + "learning_rates"][:n_steps], rtol=1e-12)


def test_step_trace_run_mwe():
    """examples/run_mwe.py: 5 default steps on the 10-10-10 MLP."""
    g = load_golden("step_mwe.npz")
    model = mwe_nn(g)
    lossf = torch.nn.MSELoss()
    opt = make_opt(model.parameters())
    for s in range(5):
        inputs, targets = T(g[f"inputs/{s}"]), T(g[f"targets/{s}"])

        def forward():
            out = model(inputs)
            return lossf(out, targets), out

        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            final = opt.step(forward=forward)
        np.testing.assert_allclose(trainable_vec(model).numpy(), g[f"params/{s}"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(opt.state["x0"].numpy(), g[f"x0/{s}"], rtol=1e-5, atol=1e-7)
        assert abs(final - g["final_losses"][s]) < 1e-6
    check_state(opt, g, "state/", 5)


@pytest.mark.parametrize("curv", ["ggn", "hessian"])
@pytest.mark.parametrize("seed", [0, 1, 42])
def test_step_trace_small_nn(curv, seed):
    """tests/test_optimizer.py:31-90 (frozen first layer)."""
    g = load_golden("step_smallnn.npz")
    key = f"{curv}_s{seed}"
    model = small_nn(g, key)
    lossf = torch.nn.MSELoss()
    opt = make_opt(model.parameters(), curvature_opt=curv, damping=float(g[key + "/damping"]))
    for s in range(3):
        inputs, targets = T(g[f"{key}/inputs/{s}"]), T(g[f"{key}/targets/{s}"])

        def forward():
            out = model(inputs)
            return lossf(out, targets), out

        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            opt.step(forward)
        np.testing.assert_allclose(trainable_vec(model).numpy(), g[f"{key}/params/{s}"],
                                   rtol=1e-4, atol=1e-6)
    check_state(opt, g, key + "/state/", 3)


@pytest.mark.parametrize("curv", ["ggn", "hessian"])
def test_step_trace_preconditioned(curv):
    g = load_golden("step_precond.npz")
    key = curv
    model = small_nn(g, key)
    lossf = torch.nn.MSELoss()
    opt = make_opt(model.parameters(), curvature_opt=curv, damping=float(g[key + "/damping"]))
    for s in range(3):
        inputs, targets = T(g[f"{key}/inputs/{s}"]), T(g[f"{key}/targets/{s}"])

        def forward():
            out = model(inputs)
            return lossf(out, targets), out

        M = opt.get_preconditioner(model, lossf, inputs, targets, "mean", use_backpack=(s % 2 == 0))
        assert M is not None  # documented deviation: the reference returns None here
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            opt.step(forward, M_func=M)
        np.testing.assert_allclose(trainable_vec(model).numpy(), g[f"{key}/params/{s}"],
                                   rtol=1e-4, atol=1e-6)
    check_state(opt, g, key + "/state/", 3)


@pytest.mark.parametrize("cache", [True, False])
@pytest.mark.parametrize("curv", ["ggn", "hessian"])
@pytest.mark.parametrize("reduction", ["mean", "sum"])
def test_acc_step_trace(curv, reduction, cache):
    """tests/test_optimizer_acc.py:116-175: step on the whole batch == acc_step on
    [7, 8] chunks, both == the reference's parameters."""
    g = load_golden("acc_step.npz")
    key = f"{curv}_{reduction}"
    m1, m2 = small_nn(g, key), small_nn(g, key)
    lossf = torch.nn.MSELoss(reduction=reduction)
    o1 = make_opt(m1.parameters(), curvature_opt=curv, cg_max_iter=4)
    o2 = make_opt(m2.parameters(), curvature_opt=curv, cg_max_iter=4, cache_acc_graphs=cache)
    for s in range(3):
        datalist = [(T(g[f"{key}/inputs/{s}/{c}"]), T(g[f"{key}/targets/{s}/{c}"])) for c in (0, 1)]
        inputs = torch.cat([d[0] for d in datalist])
        targets = torch.cat([d[1] for d in datalist])

        def forward():
            out = m1(inputs)
            return lossf(out, targets), out

        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            o1.step(forward=forward)
            o2.acc_step(m2, lossf, datalist, reduction=reduction)
        np.testing.assert_allclose(trainable_vec(m1).numpy(), g[f"{key}/params_step/{s}"], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(trainable_vec(m2).numpy(), g[f"{key}/params_acc/{s}"], rtol=1e-4, atol=1e-6)
        assert torch.allclose(trainable_vec(m1), trainable_vec(m2), atol=1e-4)  # test_optimizer_acc.py:40-57
    check_state(o1, g, key + "/state_step/", 3)
    check_state(o2, g, key + "/state_acc/", 3)


def _distinct_lists(g, key, s, device="cpu"):
    return {
        role: [(T(g[f"{key}/{role}_inputs/{s}/{c}"], device), T(g[f"{key}/{role}_targets/{s}/{c}"], device))
               for c in (0, 1)]
        for role in ("loss", "grad", "mvp")
    }


@pytest.mark.parametrize("cache", [True, False])
@pytest.mark.parametrize("curv", ["ggn", "hessian"])
@pytest.mark.parametrize("reduction", ["mean", "sum"])
def test_acc_step_with_distinct_loss_grad_and_curvature_data(curv, reduction, cache):
    """optimizer.py:519-606 / README.md:147-150: loss on chunks [9, 6], gradient on [7, 8],
    curvature on the smaller [5, 4] -- trace of the real reference (acc_step_distinct.npz)."""
    g = load_golden("acc_step_distinct.npz")
    key = f"{curv}_{reduction}"
    model = small_nn(g, key)
    lossf = torch.nn.MSELoss(reduction=reduction)
    opt = make_opt(model.parameters(), curvature_opt=curv, cg_max_iter=6, cache_acc_graphs=cache)
    for s in range(3):
        d = _distinct_lists(g, key, s)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            opt.acc_step(model, lossf, d["loss"], grad_datalist=d["grad"], mvp_datalist=d["mvp"],
                         reduction=reduction)
        np.testing.assert_allclose(trainable_vec(model).numpy(), g[f"{key}/params/{s}"], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(opt.state["x0"].numpy(), g[f"{key}/x0/{s}"], rtol=1e-4, atol=1e-6)
    check_state(opt, g, key + "/state/", 3)


@pytest.mark.parametrize("reduction", ["mean", "sum"])
@pytest.mark.parametrize("curv", ["ggn", "hessian"])
def test_test_reduction(curv, reduction):
    """tests/test_optimizer_acc.py:77-109."""
    g = load_golden("acc_step.npz")
    model = small_nn(g, f"{curv}_{reduction}")
    lossf = torch.nn.MSELoss(reduction=reduction)
    gen = torch.Generator().manual_seed(1)
    datalist = [(torch.rand(n, 7, generator=gen), torch.rand(n, 3, generator=gen)) for n in (4, 3, 7)]
    opt = make_opt(model.parameters(), curvature_opt=curv)
    opt.test_reduction(model, lossf, datalist, reduction)
    with pytest.raises(RuntimeError):
        opt.test_reduction(model, lossf, datalist, "mean" if reduction == "sum" else "sum")


def test_quadratic_is_solved_in_one_newton_step():
    """tests/test_optimizer.py:97-155."""
    g = load_golden("quadratic.npz")
    for key in [str(k) for k in g["index"]]:
        A, b, c = T(g[key + "/A"]), T(g[key + "/b"]), T(g[key + "/c"])
        params = T(g[key + "/init"]).clone().requires_grad_(True)

        def forward():
            return 0.5 * params.T @ A @ params + params.T @ b + c, None

        opt = make_opt([params], curvature_opt="hessian", lr=1.0, use_linesearch=False, damping=0.0,
                       adapt_damping=False, use_cg_backtracking=False)
        opt.step(forward=forward)
        assert torch.allclose(params.detach(), torch.linalg.solve(A, -b), atol=1e-3)
        np.testing.assert_allclose(params.detach().numpy(), g[key + "/after"], rtol=1e-4, atol=1e-5)
        assert opt.state["num_cg_iters"] == g[key + "/num_cg_iters"].tolist()
        assert opt.state["cg_reasons"][0] == str(g[key + "/cg_reason"])


def test_constructor_validation():
    p = [torch.nn.Parameter(torch.zeros(3))]
    for bad in (dict(curvature_opt="fisher"), dict(damping=-1.0), dict(cg_max_iter=0), dict(lr=-0.1)):
        with pytest.raises(ValueError):
            hf.HessianFree(p, **bad)
    with pytest.raises(ValueError):
        hf.HessianFree([{"params": p}, {"params": [torch.nn.Parameter(torch.zeros(2))]}])
    with pytest.warns(UserWarning, match="won't get adapted"):
        opt = hf.HessianFree(p, damping=0.0)
    assert opt.adapt_damping is False
    assert opt.defaults == dict(curvature_opt="ggn", damping=0.0, cg_max_iter=250, lr=1.0)


# ---- curvature products ---------------------------------------------------------
def test_curvature_products_match_reference_and_explicit_matrices():
    """``Gv``/``Hv`` against the reference's vectors AND against explicitly
    materialised ``J^T H_L J`` / Hessian (the reference has no independent check
    of its GGN product)."""
    g = load_golden("curvature.npz")
    for key in [str(k) for k in g["index"]]:
        model = small_nn(g, key)
        lossf = torch.nn.MSELoss(reduction=key.split("_")[-1])
        inputs, targets, v = T(g[key + "/inputs"]), T(g[key + "/targets"]), T(g[key + "/v"])
        params = [p for p in model.parameters() if p.requires_grad]
        out = model(inputs)
        loss = lossf(out, targets)
        Gv = hf.HessianFree._Gv(loss, out, params, v)
        Hv = hf.HessianFree._Hv(loss, params, v)
        np.testing.assert_allclose(Gv.numpy(), g[key + "/Gv"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(Hv.numpy(), g[key + "/Hv"], rtol=1e-5, atol=1e-6)
        grads = torch.autograd.grad(loss, params, retain_graph=True)
        np.testing.assert_allclose(curvature.flatten_into(grads, params).numpy(), g[key + "/grad"],
                                   rtol=1e-6, atol=1e-7)

        # explicit J (outputs x params), H_L (outputs x outputs), Hessian (params x params)
        n, m = v.numel(), out.numel()
        J = torch.zeros(m, n, dtype=torch.float64)
        for i in range(m):
            gi = torch.autograd.grad(out.reshape(-1)[i], params, retain_graph=True)
            J[i] = torch.cat([a.reshape(-1) for a in gi]).double()
        o = out.detach().clone().requires_grad_(True)
        lo = lossf(o, targets)
        (dl,) = torch.autograd.grad(lo, o, create_graph=True)
        HL = torch.stack([torch.autograd.grad(dl.reshape(-1)[i], o, retain_graph=True)[0].reshape(-1)
                          for i in range(m)]).double()
        np.testing.assert_allclose(Gv.numpy(), (J.T @ (HL @ (J @ v.double()))).numpy(), rtol=1e-4, atol=1e-6)
        gflat = torch.cat([a.reshape(-1) for a in torch.autograd.grad(loss, params, create_graph=True)])
        H = torch.stack([torch.cat([a.reshape(-1) for a in
                                    torch.autograd.grad(gflat[i], params, retain_graph=True)])
                         for i in range(n)]).double()
        np.testing.assert_allclose(Hv.numpy(), (H @ v.double()).numpy(), rtol=1e-4, atol=1e-6)


def test_diag_empirical_fisher_and_recipe():
    """tests/test_preconditioners.py:54-127."""
    g = load_golden("curvature.npz")
    for key in [str(k) for k in g["index"]]:
        model = small_nn(g, key)
        red = key.split("_")[-1]
        lossf = torch.nn.MSELoss(reduction=red)
        inputs, targets, v = T(g[key + "/inputs"]), T(g[key + "/targets"]), T(g[key + "/v"])
        for n in (1, 16):
            ref = g[f"{key}/diagEF_n{n}"]
            d_ag = hf.diag_EF_autograd(model, lossf, inputs[:n], targets[:n], red)
            d_bp = hf.diag_EF_backpack(model, lossf, inputs[:n], targets[:n], red)
            np.testing.assert_allclose(d_ag.numpy(), ref, rtol=1e-5, atol=1e-8)
            np.testing.assert_allclose(d_bp.numpy(), ref, rtol=1e-5, atol=1e-8)
        M = hf.diag_to_preconditioner(T(g[f"{key}/diagEF_n16"]), 0.1, 0.75)
        np.testing.assert_allclose(M(v).numpy(), g[key + "/Minv_v"], rtol=1e-6)
    torch.manual_seed(0)
    d = torch.rand(10)
    P = torch.diag((d + 0.1) ** 0.75)
    M = hf.diag_to_preconditioner(d, 0.1, 0.75)
    for _ in range(5):
        vec = torch.rand(10)
        assert torch.allclose(P @ M(vec), vec)
    with pytest.raises(ValueError):
        hf.diag_EF_autograd(None, None, None, None, "max")


# ---- back-tracking / line search / utils -----------------------------------------
def test_backtracking_toy():
    """tests/test_cg_backtracking.py:8-44."""
    g = load_golden("tables.npz")
    steps = [2.0, 1.0, None, 2.7, 2.4, None, None, 7.3]
    bi, bf = hf.cg_backtracking(lambda s: s + 10, steps)
    ei, ef = hf.cg_efficient_backtracking(lambda s: s + 10, steps)
    assert int(bi) == 1 == int(g["bt/exhaustive"][0]) and bf == g["bt/exhaustive"][1]
    assert int(ei) == 4 == int(g["bt/efficient"][0]) and ef == g["bt/efficient"][1]


def test_linesearch_table():
    g = load_golden("tables.npz")

    def f(step):
        return float(((1.0 + step) ** 4).sum())

    for g0, st, a0, a_ref, f_ref in g["ls/rows"]:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            a, fa = hf.simple_linesearch(f, torch.tensor([g0], dtype=torch.float32),
                                         torch.tensor([st], dtype=torch.float32), init_alpha=a0)
        assert a == pytest.approx(a_ref, rel=1e-12) and fa == pytest.approx(f_ref, rel=1e-6)
    with pytest.raises(ValueError):
        hf.simple_linesearch(f, torch.ones(1), torch.ones(1), beta=1.0)
    with pytest.warns(UserWarning, match="not a descent"):
        hf.simple_linesearch(f, torch.tensor([4.0]), torch.tensor([1.0]))


def test_vector_utils():
    """hessianfree/utils.py:8-76: frozen parameters are skipped, leftovers warn,
    non-tensors raise."""
    lin1, lin2 = torch.nn.Linear(2, 2), torch.nn.Linear(2, 1)
    for p in lin1.parameters():
        p.requires_grad = False
    params = list(lin1.parameters()) + list(lin2.parameters())
    vec = torch.arange(3.0)
    hf.vector_to_trainparams(vec, params)
    assert torch.equal(lin2.weight.data.reshape(-1), vec[:2]) and lin2.bias.data_ptr() == vec[2:].data_ptr()
    with pytest.warns(UserWarning, match="Not all entries"):
        hf.vector_to_trainparams(torch.arange(5.0), params)
    views = hf.vector_to_parameter_list(torch.arange(3.0), list(lin2.parameters()))
    assert [tuple(v.shape) for v in views] == [(1, 2), (1,)]
    with pytest.warns(UserWarning, match="Not all entries"):
        hf.vector_to_parameter_list(torch.arange(4.0), list(lin2.parameters()))
    with pytest.raises(TypeError):
        hf.vector_to_parameter_list([1.0, 2.0], params)
    with pytest.raises(TypeError):
        hf.vector_to_trainparams([1.0], params)


def test_state_dict_round_trip_keeps_warm_start_and_damping():
    """Checkpoint / resume through torch.optim.Optimizer.state_dict (string-keyed
    state as in optimizer.py:183-192; examples/run_small_nn.py:47-52 reads it)."""
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.Tanh(), torch.nn.Linear(3, 2))
    x, t = torch.rand(8, 4), torch.rand(8, 2)

    def forward():
        out = net(x)
        return torch.nn.functional.mse_loss(out, t), out

    opt = make_opt(net.parameters())
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        opt.step(forward)
    sd = opt.state_dict()
    assert set(sd["state"].keys()) >= {"x0", "init_losses", "dampings", "cg_reasons", "num_cg_iters",
                                       "best_cg_iters", "learning_rates"}
    opt2 = make_opt(net.parameters())
    opt2.load_state_dict(sd)
    assert torch.equal(opt2.state["x0"], opt.state["x0"])
    assert opt2._group["damping"] == opt._group["damping"] != 1.0
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        opt2.step(forward)
    assert opt2.state["dampings"][-1] == sd["param_groups"][0]["damping"]
    assert opt2.param_groups[0]["damping"] == opt2._group["damping"]


def test_modelprep_is_transparent_on_cpu():
    """modelprep patches keep parameters, their order and the state_dict; on CPU
    tensors (and in training mode) every patched layer falls back to the stock op."""
    from pytorchhessianfree_amd import modelprep
    from pytorchhessianfree_amd import testproblems as tp

    model, (x, t), lossf = tp.resnet18_mnist(batch_size=3)
    names = [n for n, _ in model.named_parameters()]
    ids = [id(p) for p in model.parameters()]
    ref = model(x)
    modelprep.prepare_model(model)
    assert [n for n, _ in model.named_parameters()] == names
    assert [id(p) for p in model.parameters()] == ids
    assert torch.equal(model(x), ref)
    assert modelprep.fuse_eval_batchnorm(model) == 0 and modelprep.fuse_residual_blocks(model) == 0  # idempotent
    model.train()
    assert model(x).shape == ref.shape
    model.eval()
    # curvature products through the patched model equal the stock ones on CPU
    stock, _, _ = tp.resnet18_mnist(batch_size=3)
    stock.load_state_dict(model.state_dict())
    v = torch.randn(tp.count_trainable(stock), generator=torch.Generator().manual_seed(0))
    res = []
    for m in (stock, model):
        ps = list(m.parameters())
        out = m(x)
        res.append(curvature.GGNOperator(lossf(out, t), out, ps)(v))
    assert torch.allclose(res[0], res[1], rtol=1e-5, atol=1e-7)
    with modelprep.first_order_only():
        assert modelprep._Mode.first_order_only is True
    assert modelprep._Mode.first_order_only is False


def test_modelprep_layout_helpers_on_cpu_tensors():
    """Pure-Python pieces of the tangent plumbing: recognising the first-channels slice of a
    wider NCHW / NHWC buffer (``_slice_ld``) and the layers that are GEMMs on one kernel
    tap (``_point``)."""
    from pytorchhessianfree_amd import modelprep as mp

    cl = torch.channels_last
    for fmt in (torch.contiguous_format, cl):
        x = torch.zeros(3, 8, 5, 4).contiguous(memory_format=fmt)
        wide = torch.zeros(3, 16, 5, 4).contiguous(memory_format=fmt)
        assert mp._slice_ld(x, x) == 0
        assert mp._slice_ld(wide[:, :8], x) == (16 if fmt is cl else 16 * 20)
        assert mp._slice_ld(wide[:, 8:], x) == (16 if fmt is cl else 16 * 20)   # any channel offset
        assert mp._slice_ld(wide[:, ::2], x) is None                            # every other channel
        assert mp._slice_ld(wide[:, :8, :, :3], x) is None                      # wrong shape
        assert mp._slice_ld(x.double(), x) is None
    other = torch.zeros(3, 8, 5, 4).contiguous(memory_format=cl)
    assert mp._slice_ld(other, torch.zeros(3, 8, 5, 4)) is None                 # dense, but in the other layout
    one = torch.zeros(4, 16, 1, 1)
    assert mp._slice_ld(torch.zeros(4, 32, 1, 1)[:, :16], one) == 32            # 1x1 maps: layouts coincide

    x1 = torch.zeros(2, 8, 1, 1)
    w3, w5, w1, w2 = (torch.zeros(6, 8, k, k) for k in (3, 5, 1, 2))
    assert mp._point(x1, w3, [1, 1], [1, 1], True) == 1
    assert mp._point(x1, w5, [2, 2], [1, 1], True) == 2
    assert mp._point(x1, w1, [0, 0], [1, 1], True) == 0
    assert mp._point(x1, w3, [1, 1], [1, 1], False) is None                     # NCHW: the tap is strided
    assert mp._point(x1, w3, [0, 0], [1, 1], True) is None                      # not "same" padding
    assert mp._point(x1, w3, [1, 1], [2, 2], True) is None                      # dilated
    assert mp._point(x1, w2, [1, 1], [1, 1], True) is None                      # even kernel
    assert mp._point(torch.zeros(2, 8, 2, 2), w3, [1, 1], [1, 1], True) is None  # larger map


def test_fused_residual_block_is_verified_against_the_modules_own_forward():
    """A module that merely LOOKS like a torchvision BasicBlock (conv1, bn1, relu, conv2, bn2,
    downsample) but computes something else must keep its own function: the fused forward is
    compared with the stock forward on first use and reverted on a mismatch."""
    from pytorchhessianfree_amd import modelprep

    class LookAlike(torch.nn.Module):
        def __init__(self, gate):
            super().__init__()
            self.conv1 = torch.nn.Conv2d(4, 4, 3, 1, 1, bias=False)
            self.bn1 = torch.nn.BatchNorm2d(4)
            self.relu = torch.nn.ReLU()
            self.conv2 = torch.nn.Conv2d(4, 4, 3, 1, 1, bias=False)
            self.bn2 = torch.nn.BatchNorm2d(4)
            self.downsample = None
            self.gate = gate

        def forward(self, x):
            out = self.relu(self.bn1(self.conv1(x)))
            out = self.bn2(self.conv2(out))
            return self.relu(self.gate * out + x)  # gate != 1: not the torchvision function

    torch.manual_seed(0)
    x = torch.randn(2, 4, 5, 5)
    for gate, reverted in ((1.0, False), (0.5, True)):
        block = LookAlike(gate).eval()
        want = block(x)
        assert modelprep.fuse_residual_blocks(block) == 1
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            got = block(x)
        assert torch.allclose(got, want, atol=1e-6)
        assert any("fused residual-block forward" in str(m.message) for m in w) == reverted
        assert torch.allclose(block(x), want, atol=1e-6)  # and on every later call


def test_live_taps_match_a_brute_force_convolution():
    """``engine._live_taps``: a kernel tap is live iff the weight gradient of an all-ones problem is
    non-zero there (what ``hf_conv2d_nhwc`` skips, what pack / unpack / the compact all-reduce
    leave out) -- against a brute-force weight gradient on the CPU."""
    from pytorchhessianfree_amd.engine import _live_taps

    for h, w, r, s, stride, pad in [(1, 1, 3, 3, (1, 1), (1, 1)), (2, 2, 3, 3, (2, 2), (1, 1)),
                                    (2, 2, 3, 3, (1, 1), (1, 1)), (7, 7, 3, 3, (1, 1), (1, 1)),
                                    (3, 1, 3, 3, (1, 1), (1, 1)), (4, 4, 1, 1, (2, 2), (0, 0)),
                                    (2, 3, 4, 4, (2, 1), (1, 2)), (1, 1, 5, 5, (1, 1), (2, 2))]:
        x = torch.ones(1, 1, h, w)
        wgt = torch.ones(1, 1, r, s, requires_grad=True)
        y = torch.nn.functional.conv2d(x, wgt, None, stride, pad)
        (gw,) = torch.autograd.grad(y.sum(), wgt)
        want = sum(1 << (i * s + j) for i in range(r) for j in range(s) if gw[0, 0, i, j] != 0)
        got = _live_taps(h, w, r, s, stride, pad)
        if r * s > 16:
            assert got == 0  # does not fit the kernels' 16-bit masks: everything is copied
        else:
            assert got == (0 if want == (1 << (r * s)) - 1 else want), (h, w, r, s, stride, pad)


def test_relu_margin_reports_the_smallest_relu_input():
    from pytorchhessianfree_amd import testproblems as tp

    net = torch.nn.Sequential(torch.nn.Linear(3, 4), torch.nn.ReLU(), torch.nn.Linear(4, 2), torch.nn.ReLU()).double()
    x = torch.randn(5, 3, dtype=torch.float64, generator=torch.Generator().manual_seed(2))
    z1 = net[0](x)
    z2 = net[2](torch.relu(z1))
    want = min(float(z1.abs().min()), float(z2.abs().min()))
    assert tp.relu_margin(net, x) == want
    # the seeds the parity tests and bench.py draw their ResNet-18 batches from
    model, (xb, _), _ = tp.resnet18_mnist(batch_size=32, data_seed=tp.RESNET18_B32_SEPARATED_SEEDS[0])
    assert tp.relu_margin(model.double(), xb.double()) > 8e-7


def test_configure_is_explicit_and_idempotent(monkeypatch, tmp_path):
    """``config.configure()`` and its switches: ``HF_ALLOW_WINOGRAD=1`` / ``HF_ALLOW_WRW_XDLOPS=1`` leave MIOpen's
    solver families alone, ``HF_MIOPEN_DB=<dir>`` seeds a writable copy of the shipped records there, ``off`` sets no
    user db; variables the user already set are kept; a second call is a no-op.  ``HF_NHWC_FIND=1`` keeps MIOpen's find
    step for NHWC problems (``modelprep._miopen_mode``)."""
    import os

    from pytorchhessianfree_amd import config, modelprep

    for k in ("MIOPEN_DEBUG_CONV_WINOGRAD", "MIOPEN_DEBUG_GROUP_CONV_IMPLICIT_GEMM_HIP_WRW_XDLOPS", "MIOPEN_USER_DB_PATH",
              "PYTORCH_MIOPEN_SUGGEST_NHWC", "HF_ALLOW_WINOGRAD", "HF_ALLOW_WRW_XDLOPS", "HF_MIOPEN_DB", "HF_NHWC_FIND"):
        monkeypatch.delenv(k, raising=False)
    saved = dict(config._state)
    bench = torch.backends.cudnn.benchmark
    try:
        monkeypatch.setenv("HF_MIOPEN_DB", str(tmp_path / "db"))
        got = config.configure(force=True, find=False)
        assert got["MIOPEN_DEBUG_CONV_WINOGRAD"] == "0"
        assert got["MIOPEN_DEBUG_GROUP_CONV_IMPLICIT_GEMM_HIP_WRW_XDLOPS"] == "0"
        assert got["MIOPEN_USER_DB_PATH"] == str(tmp_path / "db") and os.path.isdir(tmp_path / "db")
        assert sorted(os.listdir(tmp_path / "db")) == sorted(
            f for f in os.listdir(config._SHIPPED_DB) if os.path.isfile(os.path.join(config._SHIPPED_DB, f)))
        assert config.configure(find=False) is got  # idempotent
        for k in ("MIOPEN_DEBUG_CONV_WINOGRAD", "MIOPEN_DEBUG_GROUP_CONV_IMPLICIT_GEMM_HIP_WRW_XDLOPS",
                  "MIOPEN_USER_DB_PATH"):
            monkeypatch.delenv(k, raising=False)
        monkeypatch.setenv("HF_ALLOW_WINOGRAD", "1")
        monkeypatch.setenv("HF_ALLOW_WRW_XDLOPS", "1")
        monkeypatch.setenv("HF_MIOPEN_DB", "off")
        got = config.configure(force=True, find=False)
        assert "MIOPEN_DEBUG_CONV_WINOGRAD" not in got and "MIOPEN_DEBUG_CONV_WINOGRAD" not in os.environ
        assert "MIOPEN_DEBUG_GROUP_CONV_IMPLICIT_GEMM_HIP_WRW_XDLOPS" not in got
        assert got["MIOPEN_USER_DB_PATH"] is None
        assert modelprep._miopen_mode(True).active
        monkeypatch.setenv("HF_NHWC_FIND", "1")
        assert not modelprep._miopen_mode(True).active
    finally:
        config._state.update(saved)
        torch.backends.cudnn.benchmark = bench


def test_train_mode_batchnorm_hessian_closed_forms():
    """The closed forms behind ``hf_bn_train_hessian_coeffs`` / ``hf_bn_train_hessian_apply`` (csrc/hf_bn.hip; Hessian
    products, optimizer.py:450-455, through a BatchNorm that normalises with BATCH statistics): the tangent of the
    layer's adjoint from the row sums the sweeps take -- against float64 double backward through
    ``z = gamma * xhat + beta, y = relu(z)`` under a loss with a non-trivial second derivative, 1e-12."""
    dt = torch.float64
    gen = torch.Generator().manual_seed(0)
    m, k, eps = 37, 5, 1e-5
    a = torch.randn(m, k, dtype=dt, generator=gen).requires_grad_()
    gam = torch.randn(k, dtype=dt, generator=gen).requires_grad_()
    bet = torch.randn(k, dtype=dt, generator=gen).requires_grad_()
    wl = torch.randn(m, k, dtype=dt, generator=gen)
    mu = a.mean(0)
    r = (((a - mu) ** 2).mean(0) + eps).rsqrt()
    xh = (a - mu) * r
    z = gam * xh + bet
    y = torch.relu(z)
    loss = (y * wl).sum() + (y ** 3).sum() / 3
    ga, gg, gb = torch.autograd.grad(loss, (a, gam, bet), create_graph=True)
    da = torch.randn(m, k, dtype=dt, generator=gen)
    dgam, dbet = torch.randn(k, dtype=dt, generator=gen), torch.randn(k, dtype=dt, generator=gen)
    want_a, want_g, want_b = torch.autograd.grad((ga * da).sum() + (gg * dgam).sum() + (gb * dbet).sum(), (a, gam, bet))
    with torch.no_grad():
        mask = (z > 0).to(dt)
        g_z = mask * (wl + y ** 2)                     # first-order masked cotangent
        g_gam, g_bet = (g_z * xh).sum(0), g_z.sum(0)   # first-order parameter gradients
        G = gam * g_z
        g_a = r * (G - G.mean(0) - xh * (G * xh).mean(0))
        assert torch.allclose(g_a, ga, atol=1e-12) and torch.allclose(g_gam, gg, atol=1e-12)
        # tangent sweep
        S1, Sx = da.mean(0), (da * xh).mean(0)
        dxh = r * (da - S1 - xh * Sx)
        dy = mask * (dgam * xh + gam * dxh + dbet)
        dg_z = mask * (2 * y * dy)                     # second-order masked cotangent (the loss' own curvature)
        # the five row sums the kernels read, then the closed forms
        s_gx, s_g, s_ga = (dg_z * xh).sum(0), dg_z.sum(0), r * (g_z * da).sum(0)
        corr = -r * (S1 * g_bet + Sx * g_gam)
        dgg = s_gx + s_ga + corr
        mG, m2, mGx = (dgam * g_bet + gam * s_g) / m, (dgam * g_gam + gam * dgg) / m, gam * g_gam / m
        c = (-r * Sx, r * dgam, r * gam, -r * r * mGx, r * r * mGx * Sx - r * m2, -r * mG + r * r * mGx * S1)
        got_a = c[0] * g_a + c[1] * g_z + c[2] * dg_z + c[3] * da + c[4] * xh + c[5]
    assert float((got_a - want_a).abs().max()) < 1e-12
    assert float((dgg - want_g).abs().max()) < 1e-12 and float((s_g - want_b).abs().max()) < 1e-12


def test_path_report_names_the_path_of_every_call():
    """``HessianFree.path_report()`` (round 6: no silent performance cliffs): per kind of call the path taken -- here,
    with the CPU oracle in the solver's place and no GPU, ``eager`` for ``step`` and ``acc_step``, ``user`` when the
    caller supplies gradient and product (optimizer.py:218-247) -- and nothing before the first call; no warning
    without ``graph_matvec=True`` on a CUDA model."""
    g = load_golden("step_mwe.npz")
    model = mwe_nn(g)
    lossf = torch.nn.MSELoss()
    opt = make_opt(model.parameters())
    assert opt.path_report() == {"step": None, "acc_step": None}
    inputs, targets = T(g["inputs/0"]), T(g["targets/0"])

    def forward():
        out = model(inputs)
        return lossf(out, targets), out

    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        opt.step(forward=forward)
        rep = opt.path_report()
        assert rep["step"]["path"] == "eager" and rep["step"]["what"] == opt.PATHS["eager"] and rep["acc_step"] is None
        opt.acc_step(model, lossf, [(inputs[:8], targets[:8]), (inputs[8:], targets[8:])], reduction="mean")
        assert opt.path_report()["acc_step"]["path"] == "eager"
        params = [p for p in model.parameters() if p.requires_grad]
        n = sum(p.numel() for p in params)
        opt.step(forward=forward, grad=torch.ones(n), mvp=lambda v: 2.0 * v)
        assert opt.path_report()["step"]["path"] == "user"
    assert not [w for w in rec if "slower path" in str(w.message)]
