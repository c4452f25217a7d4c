"""Golden traces of the REAL reference on the conv-net configurations of BASELINE.json (configs[1], [3], [4]).

Run in the build container only (``/root/reference`` exists there and nowhere else)::

    python tests/golden/make_golden_convnets.py            # all files (~10 min on 8 cores)
    python tests/golden/make_golden_convnets.py resnet18   # one family

What it does: imports ``hessianfree.optimizer.HessianFree`` / ``hessianfree.cg.cg`` /
``hessianfree.preconditioners`` from ``/root/reference`` (behind ``oracle.backpack_restated``: BackPACK is not
installable here, see tests/golden/make_golden.py) and drives them on the STOCK CPU models of
``pytorchhessianfree_amd.testproblems`` -- the same seeded constructors the GPU tests call, so neither the 11 M /
25 M-entry weight vectors nor the batches have to be stored: the fixture holds their SHA-1 digests, which the GPU
tests check before they compare anything (CPU RNG streams of the pinned torch build are identical on the build
container's Xeon and the GPU box's EPYC: gpurun_out/r5a/diag2.jsonl).  Stored per run: the optimizer's ``state``
lists (``init_losses, dampings, cg_reasons, num_cg_iters, best_cg_iters, learning_rates``), final losses, and a
fixed index SAMPLE (every tensor: 16 entries, + 4 096 over the whole vector) of parameters / updates / PCG iterates /
products, with the full vectors' l2 norms -- KB, not MB.

The GPU tests (``-m gpu``) take their reference side from these files instead of re-running a CPU path on the GPU
box's host cores (whose fp32 rounding moves with the host's thread count; VERDICT r4 missing #2 / weak #3).
Generated with ``torch.set_num_threads(8)``: the fixtures regenerate bit-identically with that setting.
"""

import hashlib
import os
import sys
import time
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import make_golden as mg  # noqa: E402  (registers the BackPACK restatement, imports the reference)
from hessianfree.preconditioners import diag_EF_preconditioner as ref_diag_precond  # noqa: E402
from pytorchhessianfree_amd import testproblems as tp  # noqa: E402
from torch.nn.utils.convert_parameters import parameters_to_vector  # noqa: E402

THREADS = 8
SEEDS = tp.RESNET18_B32_SEPARATED_SEEDS
npy, RefHF, ref_cg, quiet = mg.npy, mg.RefHF, mg.ref_cg, mg.quiet


def sha(t):
    return hashlib.sha1(t.detach().contiguous().cpu().numpy().tobytes()).hexdigest()


def flat(params):
    return parameters_to_vector([p.detach() for p in params])


def sample_index(params, seed=20251003, per_tensor=16, extra=4096):
    """Sorted unique indices into the flat vector: ``per_tensor`` per parameter tensor + ``extra`` anywhere."""
    g = np.random.default_rng(seed)
    idx, off = [], 0
    for p in params:
        n = p.numel()
        idx.append(off + g.choice(n, size=min(per_tensor, n), replace=False))
        off += n
    idx.append(g.integers(0, off, size=extra))
    return np.unique(np.concatenate(idx)).astype(np.int64)


def put_index(store, key, params):
    """The index sample of this run's vectors: stored once per vector length (``index_<n>``), named by ``<key>/n``."""
    n = sum(p.numel() for p in params)
    store[key + "/n"] = np.array(n)
    if f"index_{n}" not in store:
        store[f"index_{n}"] = sample_index(params)
    return store[f"index_{n}"]


def put_vec(store, key, v, idx):
    v = v.detach().double()
    store[key + "/sample"] = v[torch.from_numpy(idx)].numpy().astype(np.float32 if v.abs().max() < 3e38 else np.float64)
    store[key + "/norm"] = np.array(float(v.norm()))
    store[key + "/absmax"] = np.array(float(v.abs().max()))


def put_state(store, key, opt, finals):
    for k, v in mg.state_arrays(opt).items():
        store[f"{key}/state/{k}"] = v
    store[key + "/final_losses"] = np.array(finals, dtype=np.float64)


def run_steps(store, key, make, seeds, steps, curv="ggn", l2=0.0, precond=False, acc=None, opt_kw=None, mk=None,
              train=False, prep=None):
    """``steps`` calls of the reference's ``step`` (``acc``: ``acc_step`` on chunks of those sizes) on fresh batches."""
    mk, opt_kw = mk or {}, opt_kw or {}
    model, _, lossf0 = make(device="cpu", data_seed=seeds[0], **mk)
    if train:
        model.train()
    if prep is not None:  # (e.g. freezing layers: the optimizer then works in the subspace of trainable parameters)
        prep(model)
    lossf = tp.l2_regularized(lossf0, model, l2) if l2 else lossf0
    params = [p for p in model.parameters() if p.requires_grad]
    idx = put_index(store, key, params)
    store[key + "/init_sha1"] = np.array(sha(flat(params)))
    opt = RefHF(model.parameters(), curvature_opt=curv, **opt_kw)
    finals = []
    t0 = time.time()
    for i in range(steps):
        _, (x, t), _ = make(device="cpu", data_seed=seeds[i], **mk)
        store[f"{key}/inputs_sha1/{i}"] = np.array(sha(x))
        store[f"{key}/targets/{i}"] = npy(t)
        before = flat(params).clone()

        def forward():
            out = model(x)
            return lossf(out, t), out

        if acc is not None:
            chunks, o = [], 0
            for n in acc:
                chunks.append((x[o:o + n].contiguous(), t[o:o + n].contiguous()))
                o += n
            finals.append(quiet(opt.acc_step, model, lossf, chunks, reduction="mean"))
        else:
            M = None
            if precond:
                # (called directly: optimizer.py:943-952 drops the return value of get_preconditioner)
                M = quiet(ref_diag_precond, model, lossf, x, t, "mean", damping=opt.param_groups[0]["damping"],
                          use_backpack=False)
            finals.append(quiet(opt.step, forward, M_func=M))
        after = flat(params)
        put_vec(store, f"{key}/params/{i}", after, idx)
        put_vec(store, f"{key}/update/{i}", after - before, idx)
        put_vec(store, f"{key}/x0/{i}", opt.state["x0"], idx)
        store[f"{key}/damping_after/{i}"] = np.array(opt.param_groups[0]["damping"])
    put_state(store, key, opt, finals)
    print(f"  {key}: {time.time() - t0:.1f} s  iters {opt.state['num_cg_iters']}  best {[int(b) for b in opt.state['best_cg_iters']]}  "
          f"reasons {opt.state['cg_reasons']}  finals {finals}", flush=True)


def double_twin(model, loss_of, x):
    """The same problem in float64 (the reference's code is dtype-agnostic): what its fp32 results are rounded from."""
    import copy

    m64 = copy.deepcopy(model).double()
    return m64, loss_of(m64), x.double()


def run_solve(store, key, model, loss_of, x, t, curv, lam, cg_kw, diag_precond=False, sample_iters=None):
    """One damped PCG solve by the reference's ``cg`` on the reference's curvature product.  ``loss_of(model)``: the
    loss function (a regulariser is bound to its model's weights).  Gradient / diagonal / products are stored twice:
    as the reference computes them (fp32) and from the same code in float64 (``.../f64``) -- the distance between the
    two is the reference's OWN fp32 error, which bounds how closely any fp32 implementation can be asked to match it."""
    lossf = loss_of(model)
    params = [p for p in model.parameters() if p.requires_grad]
    idx = put_index(store, key, params)
    store[key + "/init_sha1"] = np.array(sha(flat(params)))
    store[key + "/inputs_sha1"] = np.array(sha(x))
    out = model(x)
    loss = lossf(out, t)
    grad = parameters_to_vector(torch.autograd.grad(loss, params, retain_graph=True)).detach()
    put_vec(store, key + "/grad", grad, idx)
    m64, lossf64, x64 = double_twin(model, loss_of, x)
    p64 = [p for p in m64.parameters() if p.requires_grad]
    out64 = m64(x64)
    loss64 = lossf64(out64, t.double() if t.dtype.is_floating_point else t)  # (class indices stay; MSE targets follow)
    put_vec(store, key + "/grad/f64", parameters_to_vector(torch.autograd.grad(loss64, p64, retain_graph=True)), idx)
    store[key + "/loss_f64"] = np.array(float(loss64.detach()))
    store[key + "/loss"] = np.array(float(loss.detach()))
    store[key + "/logits"] = npy(out)

    def B(v):
        if curv == "ggn":
            return RefHF._Gv(loss, out, params, v).detach()
        return RefHF._Hv(loss, params, v).detach()

    M = None
    if diag_precond:
        from hessianfree.preconditioners import diag_EF_autograd, diag_to_preconditioner

        diag = diag_EF_autograd(model, lossf, x, t, "mean")
        put_vec(store, key + "/diag", diag, idx)
        put_vec(store, key + "/diag/f64", diag_EF_autograd(m64, lossf64, x64, t, "mean"), idx)
        M = diag_to_preconditioner(diag, lam, 0.75)
    t0 = time.time()
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        x_iters, m_iters, reason = quiet_keep_warnings(ref_cg, lambda v: B(v) + lam * v, -grad, M=M, **cg_kw)
    nonpos = sorted(int(str(w.message).split("iteration ")[1].split(".")[0]) for w in rec
                    if "Directional curvature" in str(w.message))
    store[key + "/reason"] = np.array(reason)
    store[key + "/n_iters"] = np.array(len(x_iters) - 1)
    if m_iters is not None:  # (cg.py:95-97: the quadratic model's values exist only with Martens' criterion on)
        store[key + "/m_iters"] = np.array([float(m) for m in m_iters], dtype=np.float64)
    store[key + "/nonpos_iters"] = np.array(nonpos, dtype=np.int64)
    stored = [i for i, xi in enumerate(x_iters) if xi is not None]
    keep = [i for i in stored if sample_iters is None or i in sample_iters or i == len(x_iters) - 1]
    store[key + "/stored_iters"] = np.array(keep, dtype=np.int64)
    for i in keep:
        put_vec(store, f"{key}/x/{i}", x_iters[i], idx)
    if m_iters is None:
        # m(x_i) = 0.5 x^T A x - b^T x at the stored iterates, by the reference's operator in float64 accumulation
        b = -grad
        vals = []
        for i in keep:
            xi = x_iters[i]
            Ax = B(xi) + lam * xi
            vals.append(float(0.5 * torch.dot(xi.double(), Ax.double()) - torch.dot(b.double(), xi.double())))
        store[key + "/m_at_stored"] = np.array(vals, dtype=np.float64)
    print(f"  {key}: {time.time() - t0:.1f} s  {reason}  n_iters {len(x_iters) - 1}  nonpos {nonpos[:8]}", flush=True)

    def B64(v):
        if curv == "ggn":
            return RefHF._Gv(loss64, out64, p64, v).detach()
        return RefHF._Hv(loss64, p64, v).detach()

    return B, grad, idx, B64


def quiet_keep_warnings(fn, *a, **k):
    import io

    buf, old = io.StringIO(), sys.stdout
    sys.stdout = buf
    try:
        return fn(*a, **k)
    finally:
        sys.stdout = old


def put_product(store, key, B, n, idx, seed, B64=None):
    v = torch.randn(n, generator=torch.Generator().manual_seed(seed))
    store[key + "/n"] = np.array(n)
    store[key + "/v_seed"] = np.array(seed)
    store[key + "/v_sha1"] = np.array(sha(v))
    put_vec(store, key, B(v), idx)
    if B64 is not None:
        put_vec(store, key + "/f64", B64(v.double()), idx)


# ------------------------------------------------------------------------------------------------------------------
def make_resnet18():
    store = {}
    # configs[1]: three default steps on the separated-seed batches (the session / data-parallel tests' reference)
    run_steps(store, "steps", tp.resnet18_mnist, SEEDS, 3, mk=dict(batch_size=32))
    # acc_step on ragged chunks [20, 12] (weights N_k / sum N, optimizer.py:677-684), cg_max_iter = 6
    run_steps(store, "acc_20_12", tp.resnet18_mnist, SEEDS, 2, acc=(20, 12), opt_kw=dict(cg_max_iter=6),
              mk=dict(batch_size=32))
    # acc_step on chunks [16, 16], default settings
    run_steps(store, "acc_16_16", tp.resnet18_mnist, SEEDS, 2, acc=(16, 16), mk=dict(batch_size=32))
    # Hessian curvature, one default step
    run_steps(store, "hessian_step", tp.resnet18_mnist, SEEDS, 1, curv="hessian", mk=dict(batch_size=32))
    # the damped GGN solve of the bench's problem: to Martens' criterion, and 250 forced iterations
    model, (x, t), lossf = tp.resnet18_mnist(batch_size=32, device="cpu", data_seed=SEEDS[0])
    B, grad, idx, B64 = run_solve(store, "solve_martens", model, lambda m: lossf, x, t, "ggn", 1e-3,
                                  dict(max_iter=80, martens_conv_crit=True, store_x_at_iters=list(range(81))),
                                  sample_iters=set(range(0, 12)))
    put_product(store, "ggn_product", B, grad.numel(), idx, seed=41, B64=B64)
    grid = [0, 1, 2, 3, 4, 6, 8, 10, 13, 17, 23, 30, 39, 51, 66, 86, 112, 146, 190, 247]
    run_solve(store, "solve_250", model, lambda m: lossf, x, t, "ggn", 1e-3,
              dict(max_iter=250, tol=0.0, martens_conv_crit=False, store_x_at_iters=grid))
    params = [p for p in model.parameters() if p.requires_grad]
    out = model(x)
    loss = lossf(out, t)
    m64, l64, x64 = double_twin(model, lambda m: lossf, x)
    p64 = [p for p in m64.parameters() if p.requires_grad]
    loss64 = l64(m64(x64), t)
    put_product(store, "hessian_product", lambda v: RefHF._Hv(loss, params, v).detach(), grad.numel(), idx, seed=43,
                B64=lambda v: RefHF._Hv(loss64, p64, v).detach())
    # train-mode BatchNorm (what examples/run_resnet18_mnist.py runs), batch 16: product, short solve
    model, (x, t), lossf = tp.resnet18_mnist(batch_size=16, device="cpu", data_seed=5)
    model.train()
    B, grad, idx, B64 = run_solve(store, "train_solve", model, lambda m: lossf, x, t, "ggn", 1.0,
                                  dict(max_iter=8, martens_conv_crit=True, store_x_at_iters=list(range(9))))
    put_product(store, "train_product", B, grad.numel(), idx, seed=2, B64=B64)
    mg.save("convnet_resnet18.npz", store)


def make_resnet18_train_hessian():
    """``curvature_opt="hessian"`` on the TRAIN-mode ResNet-18 (the model of examples/run_resnet18_mnist.py:19-35, which
    never calls ``model.eval()``), batch 16: the reference's ``_Hv`` (optimizer.py:450-455) -- one product, and a short
    damped solve on it.  A file of its own: the other ResNet-18 traces stay byte-identical."""
    store = {}
    model, (x, t), lossf = tp.resnet18_mnist(batch_size=16, device="cpu", data_seed=5)
    model.train()
    B, grad, idx, B64 = run_solve(store, "train_hessian_solve", model, lambda m: lossf, x, t, "hessian", 1.0,
                                  dict(max_iter=6, martens_conv_crit=True, store_x_at_iters=list(range(7))))
    put_product(store, "train_hessian_product", B, grad.numel(), idx, seed=7, B64=B64)
    mg.save("convnet_resnet18_train_hessian.npz", store)


def make_resnet18_frozen():
    """The ResNet-18 of configs[1] with its stem and layer1 FROZEN (``requires_grad = False``: the reference computes
    "in the subspace of trainable parameters", optimizer.py:121-123, utils.py:31-32; its own test problem freezes the
    first layer, tests/test_utils.py:39-43): three default steps, the gradient / one GGN product (+ float64 twins) and
    a short damped solve on the 11 024 138-entry trainable vector.  A file of its own: the other traces stay
    byte-identical."""
    store = {}
    run_steps(store, "steps", tp.resnet18_mnist, SEEDS, 3, mk=dict(batch_size=32), prep=tp.freeze_stem_and_layer1)
    model, (x, t), lossf = tp.resnet18_mnist(batch_size=32, device="cpu", data_seed=SEEDS[0])
    tp.freeze_stem_and_layer1(model)
    B, grad, idx, B64 = run_solve(store, "solve_martens", model, lambda m: lossf, x, t, "ggn", 1e-3,
                                  dict(max_iter=40, martens_conv_crit=True, store_x_at_iters=list(range(41))),
                                  sample_iters=set(range(0, 12)))
    put_product(store, "ggn_product", B, grad.numel(), idx, seed=53, B64=B64)
    # Hessian curvature on the frozen model: one default step, one product (+ float64 twin)
    run_steps(store, "hessian_step", tp.resnet18_mnist, SEEDS, 1, curv="hessian", mk=dict(batch_size=32),
              prep=tp.freeze_stem_and_layer1)
    params = [p for p in model.parameters() if p.requires_grad]
    loss = lossf(model(x), t)
    m64, l64, x64 = double_twin(model, lambda m: lossf, x)
    p64 = [p for p in m64.parameters() if p.requires_grad]
    loss64 = l64(m64(x64), t)
    put_product(store, "hessian_product", lambda v: RefHF._Hv(loss, params, v).detach(), grad.numel(), idx, seed=61,
                B64=lambda v: RefHF._Hv(loss64, p64, v).detach())
    mg.save("convnet_resnet18_frozen.npz", store)


def make_resnet18_mse():
    """The ResNet-18 of configs[1] under the loss of the reference's own examples and tests (``nn.MSELoss``,
    examples/run_mwe.py:19, tests/test_utils.py:47) against one-hot targets: two default steps, the gradient / one GGN
    product (+ float64 twins) and a short damped solve.  A file of its own."""
    store = {}
    run_steps(store, "steps", tp.resnet18_mnist_mse, SEEDS, 2, mk=dict(batch_size=32))
    model, (x, t), lossf = tp.resnet18_mnist_mse(batch_size=32, device="cpu", data_seed=SEEDS[0])
    B, grad, idx, B64 = run_solve(store, "solve_martens", model, lambda m: lossf, x, t, "ggn", 1e-3,
                                  dict(max_iter=30, martens_conv_crit=True, store_x_at_iters=list(range(31))),
                                  sample_iters=set(range(0, 12)))
    put_product(store, "ggn_product", B, grad.numel(), idx, seed=59, B64=B64)
    # acc_step under the MSE loss on ragged chunks [20, 12] (the loss of the reference's own acc tests,
    # tests/test_optimizer_acc.py:47-60), cg_max_iter = 6
    run_steps(store, "acc_20_12", tp.resnet18_mnist_mse, SEEDS, 2, acc=(20, 12), opt_kw=dict(cg_max_iter=6),
              mk=dict(batch_size=32))
    mg.save("convnet_resnet18_mse.npz", store)


def make_allcnnc():
    store = {}
    run_steps(store, "ggn_steps", tp.allcnnc_cifar100, (11, 12, 13), 3, mk=dict(batch_size=32))
    # configs[3] as stated: Hessian + L2 + diagonal empirical-Fisher preconditioner rebuilt per step
    run_steps(store, "config4_steps", tp.allcnnc_cifar100, (11, 12), 2, curv="hessian", l2=5e-4, precond=True,
              mk=dict(batch_size=32))
    run_steps(store, "config4_step_seed21", tp.allcnnc_cifar100, (21,), 1, curv="hessian", l2=5e-4, precond=True,
              mk=dict(batch_size=32))
    for lam in (1.0, 0.01):
        model, (x, t), lossf0 = tp.allcnnc_cifar100(batch_size=32, device="cpu")
        B, grad, idx, B64 = run_solve(store, f"config4_solve_lam{lam}", model,
                                      lambda m: tp.l2_regularized(lossf0, m, 5e-4), x, t, "hessian", lam,
                                      dict(max_iter=40, martens_conv_crit=True, store_x_at_iters=list(range(41))),
                                      diag_precond=True, sample_iters=set(range(0, 12)))
        if lam == 1.0:
            put_product(store, "hessian_l2_product", B, grad.numel(), idx, seed=47, B64=B64)
    model, (x, t), lossf = tp.allcnnc_cifar100(batch_size=32, device="cpu")
    params = [p for p in model.parameters() if p.requires_grad]
    idx = put_index(store, "products", params)
    out = model(x)
    loss = lossf(out, t)
    m64, l64, x64 = double_twin(model, lambda m: lossf, x)
    p64 = [p for p in m64.parameters() if p.requires_grad]
    out64 = m64(x64)
    loss64 = l64(out64, t)
    store["products/init_sha1"] = np.array(sha(flat(params)))
    store["products/inputs_sha1"] = np.array(sha(x))
    store["products/logits"] = npy(out)
    put_vec(store, "products/grad", parameters_to_vector(torch.autograd.grad(loss, params, retain_graph=True)), idx)
    put_vec(store, "products/grad/f64", parameters_to_vector(torch.autograd.grad(loss64, p64, retain_graph=True)), idx)
    n = sum(p.numel() for p in params)
    put_product(store, "products/ggn", lambda v: RefHF._Gv(loss, out, params, v).detach(), n, idx, seed=44,
                B64=lambda v: RefHF._Gv(loss64, out64, p64, v).detach())
    put_product(store, "products/hessian", lambda v: RefHF._Hv(loss, params, v).detach(), n, idx, seed=45,
                B64=lambda v: RefHF._Hv(loss64, p64, v).detach())
    mg.save("convnet_allcnnc.npz", store)


def make_bottleneck():
    store = {}
    # The Bottleneck (ResNet-50 topology, N = 25 557 032) net on 32x32 images, batch 4; 5 PCG iterations, no
    # back-tracking (tests/test_session_gpu.py::test_bottleneck_net_session_steps...)
    run_steps(store, "steps", tp.resnet50_small_images, (11, 12), 2,
              opt_kw=dict(cg_max_iter=5, use_cg_backtracking=False), mk=dict(batch_size=4, image=32))
    mg.save("convnet_bottleneck.npz", store)


def mlp25m(device="cpu", data_seed=1):
    """BASELINE configs[4]'s vector length on a plain MLP (tests/test_optimizer_gpu.py::_mlp25m)."""
    torch.manual_seed(0)
    net = torch.nn.Sequential(
        torch.nn.Linear(3072, 4096), torch.nn.Tanh(), torch.nn.Linear(4096, 3072), torch.nn.Tanh(),
        torch.nn.Linear(3072, 100),
    )
    g = torch.Generator().manual_seed(data_seed)
    x = torch.rand(64, 3072, generator=g)
    t = torch.randint(0, 100, (64,), generator=g)
    return net.to(device), (x.to(device), t.to(device)), torch.nn.CrossEntropyLoss()


def make_mlp25m():
    store = {}
    run_steps(store, "steps", mlp25m, (1, 1), 2, opt_kw=dict(cg_max_iter=12, damping=0.5))
    mg.save("convnet_mlp25m.npz", store)


if __name__ == "__main__":
    torch.set_num_threads(THREADS)
    makers = {"resnet18": make_resnet18, "resnet18_train_hessian": make_resnet18_train_hessian,
              "resnet18_frozen": make_resnet18_frozen, "resnet18_mse": make_resnet18_mse, "allcnnc": make_allcnnc, "bottleneck": make_bottleneck, "mlp25m": make_mlp25m}
    only = sys.argv[1:] or list(makers)
    for name in only:
        t0 = time.time()
        print(name, flush=True)
        makers[name]()
        print(f"{name}: {time.time() - t0:.0f} s", flush=True)
