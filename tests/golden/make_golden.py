"""Generate the golden vectors in ``tests/golden/*.npz`` from the REAL reference.

Run in the build container only (``/root/reference`` exists there and nowhere
else)::

    python tests/golden/make_golden.py

What it does
  1. imports the reference package ``hessianfree`` from ``/root/reference``
     (``cg.py``, ``cg_backtracking.py``, ``linesearch.py``, ``utils.py`` are pure
     torch; ``optimizer.py`` / ``preconditioners.py`` import BackPACK, which is
     not installable here, so ``oracle.backpack_restated`` is registered under
     that name first -- see that module's docstring);
  2. runs the reference on seeded inputs and stores INPUTS AND OUTPUTS as plain
     arrays (no reference source or bytecode is stored);
  3. asserts, while it is at it, that ``oracle.pcg.pcg`` reproduces the
     reference's ``cg`` bit-for-bit on every CG case (this is what "oracle
     pinned" means).

The fixtures are consumed by ``tests/test_oracle_golden.py`` (CPU) and by the
``-m gpu`` parity tests.
"""

import copy
import io
import json
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from oracle import backpack_restated  # noqa: E402
from oracle import pcg as oracle_pcg  # noqa: E402

backpack_restated.install_as_backpack()
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(REF, "tests"))

from hessianfree.cg import cg as ref_cg  # noqa: E402
from hessianfree.cg_backtracking import (  # noqa: E402
    cg_backtracking as ref_bt,
    cg_efficient_backtracking as ref_ebt,
)
from hessianfree.linesearch import simple_linesearch as ref_ls  # noqa: E402
from hessianfree.optimizer import HessianFree as RefHF  # noqa: E402
from hessianfree.preconditioners import (  # noqa: E402
    diag_EF_autograd as ref_diag_ag,
    diag_EF_backpack as ref_diag_bp,
    diag_to_preconditioner as ref_d2p,
)
from test_utils import get_linear_system, get_small_nn_testproblem  # noqa: E402
from torch.nn.utils.convert_parameters import parameters_to_vector  # noqa: E402


def npy(t):
    return t.detach().cpu().numpy().copy()


def save(name, store):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **store)
    print(f"wrote {name}: {len(store)} arrays, {os.path.getsize(path)/1024:.1f} KiB")


def stack_iters(x_iters, like):
    """[n_iters+1, N] array with NaN rows where the reference stored None."""
    out = np.full((len(x_iters), like.numel()), np.nan, dtype=npy(like).dtype)
    for i, x in enumerate(x_iters):
        if x is not None:
            out[i] = npy(x)
    return out


def assert_oracle_identical(ref_out, ora_out):
    (rx, rm, rr), (ox, om, orr) = ref_out, ora_out
    assert rr == orr, (rr, orr)
    assert len(rx) == len(ox)
    for a, b in zip(rx, ox):
        assert (a is None) == (b is None)
        if a is not None:
            assert torch.equal(a, b)
    assert (rm is None) == (om is None)
    if rm is not None:
        assert len(rm) == len(om)
        for a, b in zip(rm, om):
            assert torch.equal(a, b)


# ------------------------------------------------------------------------------
# 1. cg on the reference's dense SPD systems (tests/test_utils.py:6-16)
# ------------------------------------------------------------------------------
def make_cg_linear():
    store, index = {}, []
    for dim in (3, 10, 50):
        for seed in (0, 1, 42):
            for precond in (0, 1):
                for x0_none in (0, 1):
                    for martens in (0, 1):
                        A, b, _ = get_linear_system(dim, seed=seed)
                        # same RNG draw as tests/test_cg.py:122
                        x0 = None if x0_none else 2 * (torch.rand(dim) - 0.5)
                        minv = torch.diag(A) ** (-1) if precond else None
                        Mmat = torch.diag(minv) if precond else None
                        kw = dict(
                            x0=x0,
                            M=(lambda v: Mmat @ v) if precond else None,
                            max_iter=10 * dim,
                            tol=1e-5,
                            atol=1e-6,
                            martens_conv_crit=bool(martens),
                            store_x_at_iters=list(range(10 * dim)),
                        )
                        with warnings.catch_warnings():
                            warnings.simplefilter("ignore")
                            out = ref_cg(lambda v: A @ v, b, **kw)
                            ora = oracle_pcg.pcg(lambda v: A @ v, b, **kw)
                        assert_oracle_identical(out, ora)
                        key = f"d{dim}_s{seed}_p{precond}_x{x0_none}_m{martens}"
                        index.append(key)
                        store[key + "/A"] = npy(A)
                        store[key + "/b"] = npy(b)
                        if x0 is not None:
                            store[key + "/x0"] = npy(x0)
                        if precond:
                            store[key + "/minv"] = npy(minv)
                        store[key + "/X"] = stack_iters(out[0], b)
                        if martens:
                            store[key + "/m"] = np.array([float(m) for m in out[1]], dtype=np.float32)
                        store[key + "/reason"] = np.array(out[2])
    store["index"] = np.array(index)
    save("cg_linear.npz", store)


def make_cg_f64():
    """float64 cases (tests/test_cg.py:159-224)."""
    store, index = {}, []
    for dim in (3, 10, 50):
        for seed in (0, 1, 42):
            A, b, _ = get_linear_system(dim, seed=seed)
            A, b = A.double(), b.double()
            kw = dict(
                max_iter=10 * dim,
                tol=1e-5,
                atol=1e-6,
                martens_conv_crit=False,
                store_x_at_iters=list(range(10 * dim)),
            )
            out = ref_cg(lambda v: A @ v, b, **kw)
            ora = oracle_pcg.pcg(lambda v: A @ v, b, **kw)
            assert_oracle_identical(out, ora)
            key = f"d{dim}_s{seed}"
            index.append(key)
            store[key + "/A"] = npy(A)
            store[key + "/b"] = npy(b)
            store[key + "/X"] = stack_iters(out[0], b)
            store[key + "/reason"] = np.array(out[2])
    store["index"] = np.array(index)
    save("cg_f64.npz", store)


# ------------------------------------------------------------------------------
# 2. a longer solve: damped low-rank + diagonal operator, diag-EF style
#    preconditioner, Martens test, automatic snapshot grid (the shape of the
#    call in optimizer.py:265-274)
# ------------------------------------------------------------------------------
def lowrank_problem(n, rank, seed, damping):
    g = torch.Generator().manual_seed(seed)
    U = torch.randn(n, rank, generator=g) / (rank**0.5)
    d = torch.rand(n, generator=g) * 0.5
    b = torch.randn(n, generator=g)
    diag = d + (U * U).sum(1)

    def A(v):
        return d * v + U @ (U.T @ v) + damping * v

    return U, d, b, diag, A


def make_cg_lowrank():
    store, index = {}, []
    for n, rank, seed, damping, precond, warm in [
        (4099, 8, 0, 1.0, 1, 0),
        (4099, 8, 1, 0.1, 0, 0),
        (4099, 8, 2, 0.01, 1, 1),
        (1021, 16, 3, 1e-3, 0, 1),
    ]:
        U, d, b, diag, A = lowrank_problem(n, rank, seed, damping)
        M = ref_d2p(diag, damping) if precond else None
        x0 = None
        if warm:
            g = torch.Generator().manual_seed(100 + seed)
            x0 = 0.1 * torch.randn(n, generator=g)
        kw = dict(
            x0=x0, M=M, max_iter=250, martens_conv_crit=True, store_x_at_iters=None
        )
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            out = ref_cg(A, b, **kw)
            ora = oracle_pcg.pcg(A, b, **kw)
        assert_oracle_identical(out, ora)
        key = f"n{n}_r{rank}_s{seed}"
        index.append(key)
        store[key + "/U"] = npy(U)
        store[key + "/d"] = npy(d)
        store[key + "/b"] = npy(b)
        store[key + "/diag"] = npy(diag)
        store[key + "/damping"] = np.array(damping)
        store[key + "/precond"] = np.array(precond)
        if x0 is not None:
            store[key + "/x0"] = npy(x0)
        store[key + "/X"] = stack_iters(out[0], b)
        store[key + "/m"] = np.array([float(m) for m in out[1]], dtype=np.float32)
        store[key + "/reason"] = np.array(out[2])
        # the reference itself in float64 on the same inputs: the trajectory
        # both fp32 runs (reference and HIP) approximate
        U64, d64, b64 = U.double(), d.double(), b.double()
        M64 = ref_d2p(diag.double(), damping) if precond else None
        kw64 = dict(kw, x0=None if x0 is None else x0.double(), M=M64)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            out64 = ref_cg(lambda v: d64 * v + U64 @ (U64.T @ v) + damping * v, b64, **kw64)
        store[key + "/X64"] = stack_iters(out64[0], b64)
        store[key + "/m64"] = np.array([float(m) for m in out64[1]], dtype=np.float64)
        print(f"  {key}: {len(out[0])-1} iters, {out[2]} (fp64: {len(out64[0])-1})")
    store["index"] = np.array(index)
    save("cg_lowrank.npz", store)


# ------------------------------------------------------------------------------
# 3. snapshot grid, back-tracking toy, line-search toy
# ------------------------------------------------------------------------------
def make_small_tables():
    store = {}
    # The grid function is nested inside the reference's cg(); observe it
    # through the x_iters pattern of a run that cannot terminate early.
    for max_iter in (1, 2, 4, 10, 37, 250, 400):
        n = 8
        A = torch.diag(torch.linspace(1.0, 2.0, n))
        b = torch.ones(n)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            xs, _, _ = ref_cg(
                lambda v: A @ v, b, max_iter=max_iter, tol=0.0, store_x_at_iters=None
            )
        stored = [i for i, x in enumerate(xs[:-1]) if x is not None]
        # the last entry is always set; whether it was "on the grid" is
        # recovered from the oracle's table below
        grid = oracle_pcg.snapshot_grid(max_iter)
        assert stored == [i for i in grid if i < len(xs) - 1], (stored, grid)
        store[f"grid/{max_iter}"] = np.array(grid)
    steps = [2.0, 1.0, None, 2.7, 2.4, None, None, 7.3]
    bi, bf = ref_bt(lambda s: s + 10, steps)
    ei, ef = ref_ebt(lambda s: s + 10, steps)
    store["bt/exhaustive"] = np.array([int(bi), float(bf)])
    store["bt/efficient"] = np.array([int(ei), float(ef)])

    # Armijo line search on a 1-D quartic: records (alpha, f) for three setups
    def f(step):
        return float(((1.0 + step) ** 4).sum())

    rows = []
    for g0, st, a0 in [(4.0, -1.0, 1.0), (4.0, -3.0, 1.0), (4.0, -3.0, 0.5), (4.0, 1.0, 1.0)]:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            a, fa = ref_ls(f, torch.tensor([g0]), torch.tensor([st]), init_alpha=a0)
        rows.append([g0, st, a0, float(a), float(fa)])
    store["ls/rows"] = np.array(rows)
    save("tables.npz", store)


# ------------------------------------------------------------------------------
# 4. curvature products and the diagonal empirical Fisher on the small nets
# ------------------------------------------------------------------------------
def model_arrays(model):
    return {k: npy(v) for k, v in model.state_dict().items()}


def make_curvature():
    store, index = {}, []
    for seed in (0, 1, 42):
        for reduction in ("mean", "sum"):
            torch.manual_seed(seed)
            model, (inputs, targets), _ = get_small_nn_testproblem(N=16)
            lossf = torch.nn.MSELoss(reduction=reduction)
            plist = [p for p in model.parameters() if p.requires_grad]
            n = sum(p.numel() for p in plist)
            v = torch.randn(n)
            outputs = model(inputs)
            loss = lossf(outputs, targets)
            grad = parameters_to_vector(
                torch.autograd.grad(loss, plist, create_graph=True, retain_graph=True)
            ).detach()
            Gv = RefHF._Gv(loss, outputs, plist, v)
            Hv = RefHF._Hv(loss, plist, v)
            key = f"smallnn_s{seed}_{reduction}"
            index.append(key)
            for k, a in model_arrays(model).items():
                store[f"{key}/model/{k}"] = a
            store[key + "/inputs"] = npy(inputs)
            store[key + "/targets"] = npy(targets)
            store[key + "/v"] = npy(v)
            store[key + "/loss"] = np.array(float(loss))
            store[key + "/grad"] = npy(grad)
            store[key + "/Gv"] = npy(Gv)
            store[key + "/Hv"] = npy(Hv)
            for nsub in (1, 16):
                d_ag = ref_diag_ag(model, lossf, inputs[:nsub], targets[:nsub], reduction)
                d_bp = ref_diag_bp(model, lossf, inputs[:nsub], targets[:nsub], reduction)
                assert torch.allclose(d_ag, d_bp)
                store[f"{key}/diagEF_n{nsub}"] = npy(d_ag)
            Mf = ref_d2p(d_ag, 0.1, 0.75)
            store[key + "/Minv_v"] = npy(Mf(v))
    store["index"] = np.array(index)
    save("curvature.npz", store)


# ------------------------------------------------------------------------------
# 5. full step() / acc_step() traces
# ------------------------------------------------------------------------------
def state_arrays(opt):
    st = opt.state
    return {
        "init_losses": np.array(st["init_losses"], dtype=np.float64),
        "dampings": np.array(st["dampings"], dtype=np.float64),
        "cg_reasons": np.array(st["cg_reasons"]),
        "num_cg_iters": np.array(st["num_cg_iters"]),
        "best_cg_iters": np.array([int(i) for i in st["best_cg_iters"]]),
        "learning_rates": np.array(st["learning_rates"], dtype=np.float64),
    }


def trainable_vec(model):
    return npy(parameters_to_vector([p for p in model.parameters() if p.requires_grad]))


def quiet(fn, *a, **k):
    buf = io.StringIO()
    old = sys.stdout
    sys.stdout = buf
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            return fn(*a, **k)
    finally:
        sys.stdout = old


def make_step_mwe():
    """examples/run_mwe.py:10-37 (MLP 10-10-10, batch 16, 5 steps)."""
    torch.manual_seed(0)
    dim, batch = 10, 16
    model = torch.nn.Sequential(
        torch.nn.Linear(dim, dim, bias=False), torch.nn.ReLU(), torch.nn.Linear(dim, dim)
    )
    lossf = torch.nn.MSELoss()
    store = {f"model/{k}": a for k, a in model_arrays(model).items()}
    opt = RefHF(model.parameters(), verbose=False)
    finals = []
    for s in range(5):
        inputs, targets = torch.rand(batch, dim), torch.rand(batch, dim)
        store[f"inputs/{s}"], store[f"targets/{s}"] = npy(inputs), npy(targets)

        def forward():
            out = model(inputs)
            return lossf(out, targets), out

        finals.append(quiet(opt.step, forward=forward))
        store[f"params/{s}"] = trainable_vec(model)
        store[f"x0/{s}"] = npy(opt.state["x0"])
    store["final_losses"] = np.array(finals, dtype=np.float64)
    store.update({"state/" + k: v for k, v in state_arrays(opt).items()})
    print("  mwe:", json.dumps({k: v.tolist() for k, v in state_arrays(opt).items()}))
    save("step_mwe.npz", store)


def make_step_smallnn():
    """tests/test_optimizer.py:31-90 (3 steps, unpreconditioned because the
    reference's get_preconditioner returns None) and examples/run_small_nn.py
    (batch 32, default damping, 2 steps)."""
    store, index = {}, []
    for curv in ("ggn", "hessian"):
        for seed in (0, 1, 42):
            torch.manual_seed(seed)
            model, _, lossf = get_small_nn_testproblem()
            damping = 1.5 if curv == "hessian" else 0.1
            key = f"{curv}_s{seed}"
            index.append(key)
            for k, a in model_arrays(model).items():
                store[f"{key}/model/{k}"] = a
            store[key + "/damping"] = np.array(damping)
            opt = RefHF(model.parameters(), curvature_opt=curv, damping=damping)
            finals = []
            for s in range(3):
                _, (inputs, targets), _ = get_small_nn_testproblem()
                store[f"{key}/inputs/{s}"] = npy(inputs)
                store[f"{key}/targets/{s}"] = npy(targets)

                def forward():
                    out = model(inputs)
                    return lossf(out, targets), out

                finals.append(quiet(opt.step, forward))
                store[f"{key}/params/{s}"] = trainable_vec(model)
            store[key + "/final_losses"] = np.array(finals, dtype=np.float64)
            for k, v in state_arrays(opt).items():
                store[f"{key}/state/{k}"] = v
    store["index"] = np.array(index)
    save("step_smallnn.npz", store)


def make_step_precond():
    """A genuinely preconditioned step (diag_EF_preconditioner called directly,
    because optimizer.py:943-952 drops the return value)."""
    from hessianfree.preconditioners import diag_EF_preconditioner

    store, index = {}, []
    for curv in ("ggn", "hessian"):
        torch.manual_seed(7)
        model, _, lossf = get_small_nn_testproblem()
        damping = 1.5 if curv == "hessian" else 0.1
        key = curv
        index.append(key)
        for k, a in model_arrays(model).items():
            store[f"{key}/model/{k}"] = a
        store[key + "/damping"] = np.array(damping)
        opt = RefHF(model.parameters(), curvature_opt=curv, damping=damping)
        finals = []
        for s in range(3):
            _, (inputs, targets), _ = get_small_nn_testproblem(N=32)
            store[f"{key}/inputs/{s}"] = npy(inputs)
            store[f"{key}/targets/{s}"] = npy(targets)

            def forward():
                out = model(inputs)
                return lossf(out, targets), out

            M = diag_EF_preconditioner(
                model, lossf, inputs, targets, "mean",
                damping=opt.param_groups[0]["damping"], use_backpack=False,
            )
            finals.append(quiet(opt.step, forward, M_func=M))
            store[f"{key}/params/{s}"] = trainable_vec(model)
        store[key + "/final_losses"] = np.array(finals, dtype=np.float64)
        for k, v in state_arrays(opt).items():
            store[f"{key}/state/{k}"] = v
    store["index"] = np.array(index)
    save("step_precond.npz", store)


def make_acc_step():
    """tests/test_optimizer_acc.py:116-175 (cg_max_iter=4, 3 steps, [7,8] chunks)."""
    store, index = {}, []
    for curv in ("ggn", "hessian"):
        for reduction in ("mean", "sum"):
            torch.manual_seed(0)
            model_1, _, _ = get_small_nn_testproblem()
            model_2 = copy.deepcopy(model_1)
            lossf = torch.nn.MSELoss(reduction=reduction)
            key = f"{curv}_{reduction}"
            index.append(key)
            for k, a in model_arrays(model_1).items():
                store[f"{key}/model/{k}"] = a
            o1 = RefHF(model_1.parameters(), curvature_opt=curv, cg_max_iter=4)
            o2 = RefHF(model_2.parameters(), curvature_opt=curv, cg_max_iter=4)
            for s in range(3):
                datalist = []
                for n in (7, 8):
                    _, data, _ = get_small_nn_testproblem(N=n)
                    datalist.append(data)
                for c, (i_, t_) in enumerate(datalist):
                    store[f"{key}/inputs/{s}/{c}"] = npy(i_)
                    store[f"{key}/targets/{s}/{c}"] = npy(t_)
                inputs = torch.cat([d[0] for d in datalist]).clone()
                targets = torch.cat([d[1] for d in datalist]).clone()

                def forward():
                    out = model_1(inputs)
                    return lossf(out, targets), out

                quiet(o1.step, forward=forward)
                quiet(o2.acc_step, model_2, lossf, datalist, reduction=reduction)
                store[f"{key}/params_step/{s}"] = trainable_vec(model_1)
                store[f"{key}/params_acc/{s}"] = trainable_vec(model_2)
            for k, v in state_arrays(o1).items():
                store[f"{key}/state_step/{k}"] = v
            for k, v in state_arrays(o2).items():
                store[f"{key}/state_acc/{k}"] = v
    store["index"] = np.array(index)
    save("acc_step.npz", store)


def make_acc_step_distinct():
    """``acc_step`` with DIFFERENT data for the loss, the gradient and the curvature -- the
    usage the reference recommends (README.md:147-150; optimizer.py:519-606): loss on
    chunks [9, 6], gradient on [7, 8], curvature products on the smaller list [5, 4]."""
    store, index = {}, []
    sizes = {"loss": (9, 6), "grad": (7, 8), "mvp": (5, 4)}
    for curv in ("ggn", "hessian"):
        for reduction in ("mean", "sum"):
            torch.manual_seed(0)
            model, _, _ = get_small_nn_testproblem()
            lossf = torch.nn.MSELoss(reduction=reduction)
            key = f"{curv}_{reduction}"
            index.append(key)
            for k, a in model_arrays(model).items():
                store[f"{key}/model/{k}"] = a
            opt = RefHF(model.parameters(), curvature_opt=curv, cg_max_iter=6)
            for s in range(3):
                lists = {}
                for role, ns in sizes.items():
                    lists[role] = []
                    for c, n in enumerate(ns):
                        _, data, _ = get_small_nn_testproblem(N=n)
                        lists[role].append(data)
                        store[f"{key}/{role}_inputs/{s}/{c}"] = npy(data[0])
                        store[f"{key}/{role}_targets/{s}/{c}"] = npy(data[1])
                quiet(opt.acc_step, model, lossf, lists["loss"], grad_datalist=lists["grad"],
                      mvp_datalist=lists["mvp"], reduction=reduction)
                store[f"{key}/params/{s}"] = trainable_vec(model)
                store[f"{key}/x0/{s}"] = npy(opt.state["x0"])
            for k, v in state_arrays(opt).items():
                store[f"{key}/state/{k}"] = v
    store["index"] = np.array(index)
    save("acc_step_distinct.npz", store)


def make_quadratic():
    """tests/test_optimizer.py:97-155: one undamped Newton step on a quadratic."""
    store, index = {}, []
    for seed in (0, 1, 42):
        for dim in (3, 5, 10):
            torch.manual_seed(seed)
            init = torch.rand((dim, 1)) - 0.5
            A, b, _ = get_linear_system(dim, seed=seed)
            b = b.reshape(dim, 1)
            c = torch.rand(1) - 0.5
            params = init.clone().detach().requires_grad_(True)

            def forward():
                return 0.5 * params.T @ A @ params + params.T @ b + c, None

            opt = RefHF(
                [params], curvature_opt="hessian", lr=1.0, use_linesearch=False,
                damping=0.0, adapt_damping=False, use_cg_backtracking=False,
            )
            quiet(opt.step, forward=forward)
            key = f"s{seed}_d{dim}"
            index.append(key)
            store[key + "/A"], store[key + "/b"], store[key + "/c"] = npy(A), npy(b), npy(c)
            store[key + "/init"] = npy(init)
            store[key + "/after"] = npy(params)
            store[key + "/num_cg_iters"] = np.array(opt.state["num_cg_iters"])
            store[key + "/cg_reason"] = np.array(opt.state["cg_reasons"][0])
    store["index"] = np.array(index)
    save("quadratic.npz", store)


if __name__ == "__main__":
    torch.set_num_threads(1)  # reduction order of the reference's dots is then fixed
    makers = [make_cg_linear, make_cg_f64, make_cg_lowrank, make_small_tables, make_curvature,
              make_step_mwe, make_step_smallnn, make_step_precond, make_acc_step,
              make_acc_step_distinct, make_quadratic]
    only = set(sys.argv[1:])  # e.g. `make_golden.py make_acc_step_distinct` regenerates one file
    for fn in makers:
        if not only or fn.__name__ in only:
            fn()
