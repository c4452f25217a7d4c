"""Shared test helpers: rebuild the reference's small test problems from the
arrays stored in tests/golden (no RNG agreement with the generator needed)."""

import numpy as np
import torch

from tol import within


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


# ---------------------------------------------------------------------------------------------------------
# Conv-net reference traces (tests/golden/convnet_*.npz, written by tests/golden/make_golden_convnets.py from the REAL
# reference in the build container): the GPU tests take their reference side from these files instead of re-running
# a CPU path on the GPU box's host cores.
# ---------------------------------------------------------------------------------------------------------
_GOLDEN_CACHE = {}


def convnet_golden(family):
    import os

    if family not in _GOLDEN_CACHE:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"convnet_{family}.npz")
        _GOLDEN_CACHE[family] = np.load(path, allow_pickle=False)
    return _GOLDEN_CACHE[family]


def sha1_of(t):
    import hashlib

    return hashlib.sha1(t.detach().contiguous().cpu().numpy().tobytes()).hexdigest()


class RefTrace:
    """One run of the reference stored under ``key`` of a conv-net fixture: ``state`` (the optimizer's lists),
    ``finals``, index sample + norms of the vectors the generator kept."""

    def __init__(self, family, key):
        self.g, self.key = convnet_golden(family), key
        g = self.g
        self.index = torch.from_numpy(g[f"index_{int(g[key + '/n'])}"])
        names = ("init_losses", "dampings", "cg_reasons", "num_cg_iters", "best_cg_iters", "learning_rates")
        if f"{key}/state/init_losses" in g.files:
            self.state = {k: g[f"{key}/state/{k}"].tolist() for k in names}
            self.finals = g[key + "/final_losses"].tolist()

    def _path(self, name):
        return self.key if not name else f"{self.key}/{name}"

    def has(self, name):
        return self._path(name) in self.g.files

    def scalar(self, name):
        return self.g[self._path(name)].item()

    def array(self, name):
        return self.g[self._path(name)]

    def check_inputs(self, params=None, x=None, step=None):
        """The problem the GPU test built IS the generator's: digests of the initial flat parameter vector and of the
        input batch (both made on the CPU from seeds, before any ``.to(device)``)."""
        if params is not None:
            flat = torch.cat([p.detach().reshape(-1).cpu() for p in params])
            assert sha1_of(flat) == str(self.array("init_sha1")), "initial parameters differ from the generator's"
        if x is not None:
            name = "inputs_sha1" if step is None else f"inputs_sha1/{step}"
            assert sha1_of(x) == str(self.array(name)), "input batch differs from the generator's"

    def probe(self):
        """The CPU-seeded vector a stored product was taken of (``put_product``): regenerated, digest checked."""
        n = int(self.scalar("n"))
        v = torch.randn(n, generator=torch.Generator().manual_seed(int(self.scalar("v_seed"))))
        assert sha1_of(v) == str(self.array("v_sha1")), "probe vector differs from the generator's"
        return v

    # a stored vector ``name`` ("" = the key itself): index sample, l2 norm and max-norm of the full vector
    def sample(self, name=""):
        return torch.from_numpy(self.array((name + "/" if name else "") + "sample").astype(np.float64))

    def _got(self, v):
        return v.detach().reshape(-1)[self.index.to(v.device)].double().cpu()

    def vec_err(self, name, v):
        """max |v[idx] - ref[idx]| / max |ref| over the stored index sample (``v``: the full vector, any device)."""
        absmax = self.scalar((name + "/" if name else "") + "absmax")
        return float((self._got(v) - self.sample(name)).abs().max() / max(absmax, 1e-300))

    def vec_rel_l2(self, name, v):
        """||v[idx] - ref[idx]|| / ||ref[idx]|| over the stored index sample."""
        ref = self.sample(name)
        return float((self._got(v) - ref).norm() / ref.norm().clamp_min(1e-300))

    def vec_cos(self, name, v):
        ref, got = self.sample(name), self._got(v)
        return float(got @ ref / (got.norm() * ref.norm()).clamp_min(1e-300))

    # ``name + "/f64"``: the same quantity from the reference's code run in float64 (the generator stores both)
    def own_err(self, name):
        """The REFERENCE's own fp32 error: max |ref32[idx] - ref64[idx]| / max |ref64| on the index sample."""
        f64 = (name + "/" if name else "") + "f64"
        return float((self.sample(name) - self.sample(f64)).abs().max() / max(self.scalar(f64 + "/absmax"), 1e-300))

    def vec_err64(self, name, v):
        """max |v[idx] - ref64[idx]| / max |ref64|: distance to the float64 value the reference's result is rounded from."""
        return self.vec_err((name + "/" if name else "") + "f64", v)

    def own_rel_l2(self, name):
        """The reference's own fp32 error in the l2 norm of the index sample: ||ref32 - ref64|| / ||ref64||."""
        f64 = self.sample((name + "/" if name else "") + "f64")
        return float((self.sample(name) - f64).norm() / f64.norm().clamp_min(1e-300))

    def envelope(self, name, tight):
        """Bound for the distance of an fp32 result to the reference's fp32 result: ``tight`` (what is asked of the
        distance to float64) + three times the reference's own fp32 distance to float64 (a result AT the float64
        value sits one such distance away)."""
        return tight + 3.0 * self.own_err(name)

    def norm_err(self, name, v):
        want = self.scalar((name + "/" if name else "") + "norm")
        return abs(float(v.detach().double().norm()) - want) / max(want, 1e-300)


def compare_trace(got_state, got_finals, ref, steps=None, loss_tol=(1e-5, 3e-5), final_tol=(1e-4, 5e-4), iters=2,
                  best=None):
    """The discrete entries of a ``step`` / ``acc_step`` trace identical to the reference's (termination reasons,
    learning rates, damping schedule), initial / final losses within the stated fp32 tolerances -- ``(first step, later
    steps)``: a later step starts from parameters that differ like any two fp32 runs, and back-tracking / the line
    search pick between nearly tied candidates (measured over the round-5 leases: initial losses <= 3.7e-6 / final
    losses <= 5.8e-5 relative on the later steps of the ResNet-18 and All-CNN-C runs) --, iteration counts within
    ``iters``."""
    def pair(t):
        return t if isinstance(t, tuple) else (t, t)

    sc = ref.state
    n = len(got_state["init_losses"]) if steps is None else steps
    for i, (a, b) in enumerate(zip(got_state["init_losses"][:n], sc["init_losses"][:n])):
        within(abs(a - b), pair(loss_tol)[min(i, 1)] * abs(b), strict=False, note=(got_state["init_losses"], sc["init_losses"]))
    assert list(got_state["cg_reasons"][:n]) == list(sc["cg_reasons"][:n]), (got_state["cg_reasons"], sc["cg_reasons"])
    assert list(got_state["learning_rates"][:n]) == list(sc["learning_rates"][:n])
    assert list(got_state["dampings"][:n]) == list(sc["dampings"][:n])
    for a, b in zip(got_state["num_cg_iters"][:n], sc["num_cg_iters"][:n]):
        assert abs(a - b) <= iters, (got_state["num_cg_iters"], sc["num_cg_iters"])
    for i, (a, b) in enumerate(zip(got_finals[:n], ref.finals[:n])):
        within(abs(a - b), pair(final_tol)[min(i, 1)] * abs(b), strict=False, note=(got_finals, ref.finals))
    if best is not None:
        for a, b in zip(got_state["best_cg_iters"][:n], sc["best_cg_iters"][:n]):
            assert abs(int(a) - int(b)) <= best, (got_state["best_cg_iters"], sc["best_cg_iters"])
