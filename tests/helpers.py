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
