"""Shared test helpers: rebuild the reference's small test problems from the
arrays stored in tests/golden (no RNG agreement with the generator needed)."""

import numpy as np
import torch


def T(a, device="cpu", dtype=None):
    t = torch.from_numpy(np.asarray(a).copy())
    if dtype is not None:
        t = t.to(dtype)
    return t.to(device)


def small_nn(arrays, prefix, device="cpu", freeze_layer1=True):
    """7-5-5-3 MLP of /root/reference/tests/test_utils.py:19-52 with weights taken
    from the golden file."""
    model = torch.nn.Sequential(
        torch.nn.Linear(7, 5),
        torch.nn.ReLU(),
        torch.nn.Sequential(torch.nn.Linear(5, 5), torch.nn.ReLU()),
        torch.nn.Linear(5, 3),
    )
    sd = {k: T(arrays[f"{prefix}/model/{k}"]) for k in model.state_dict().keys()}
    model.load_state_dict(sd)
    model = model.to(device)
    if freeze_layer1:
        for p in next(model.children()).parameters():
            p.requires_grad = False
    return model


def mwe_nn(arrays, device="cpu"):
    """MLP of /root/reference/examples/run_mwe.py:16-20."""
    model = torch.nn.Sequential(
        torch.nn.Linear(10, 10, bias=False), torch.nn.ReLU(), torch.nn.Linear(10, 10)
    )
    sd = {k: T(arrays[f"model/{k}"]) for k in model.state_dict().keys()}
    model.load_state_dict(sd)
    return model.to(device)


def trainable_vec(model):
    return torch.cat([p.detach().reshape(-1) for p in model.parameters() if p.requires_grad])


def lowrank_operator(g, key, device, dtype=torch.float32):
    """A v = d*v + U U^T v + damping*v  (tests/golden/make_golden.py:lowrank_problem)."""
    U = T(g[key + "/U"], device, dtype)
    d = T(g[key + "/d"], device, dtype)
    damping = float(g[key + "/damping"])

    def B(v):  # undamped part
        return d * v + U @ (U.T @ v)

    def A(v):
        return B(v) + damping * v

    return A, B, damping


_CPU_STEPS = {}


def cpu_resnet18_default_steps(n_steps):
    """``n_steps`` default ``HessianFree.step()`` calls of the single-process CPU path on the 32-sample ResNet-18
    batches of ``RESNET18_B32_SEPARATED_SEEDS`` -- stock model, torch autograd, host logic with the oracle PCG (the
    reference's algorithm, pinned bit for bit by tests/golden/make_golden.py).  Returns ``(state, finals, params)``
    truncated to ``n_steps``; computed once per pytest process (the longest run so far serves the shorter ones:
    the steps are sequential)."""
    import warnings

    import pytorchhessianfree_amd as hf
    from oracle import pcg as oracle
    from pytorchhessianfree_amd import testproblems as tp

    run = _CPU_STEPS.get("run")
    if run is None:
        seeds = tp.RESNET18_B32_SEPARATED_SEEDS
        model, _, lossf = tp.resnet18_mnist(batch_size=32, device="cpu", data_seed=seeds[0])
        opt = hf.HessianFree(model.parameters())
        opt._cg = oracle.pcg
        run = _CPU_STEPS["run"] = dict(model=model, lossf=lossf, opt=opt, finals=[], params=[])
    model, lossf, opt = run["model"], run["lossf"], run["opt"]
    while len(run["finals"]) < n_steps:  # (continue the same run: the steps are sequential)
        i = len(run["finals"])
        _, (x, t), _ = tp.resnet18_mnist(batch_size=32, device="cpu", data_seed=tp.RESNET18_B32_SEPARATED_SEEDS[i])

        def forward():
            o = model(x)
            return lossf(o, t), o

        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            run["finals"].append(opt.step(forward))
        run["params"].append(torch.cat([p.detach().reshape(-1) for p in opt._params_list]).numpy().copy())
    state = {k: list(v[:n_steps]) for k, v in opt.state.items() if isinstance(v, list)}
    return state, run["finals"][:n_steps], run["params"][n_steps - 1]
