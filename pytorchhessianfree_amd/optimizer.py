"""``HessianFree`` -- drop-in for the reference's ``hessianfree/optimizer.py:18-952``
with the Newton-step solve running on the MI355X.

Public surface preserved: constructor arguments and their validation
(optimizer.py:23-123), ``step`` (:126-363), ``acc_step`` (:519-606),
``test_reduction`` (:817-926), ``get_preconditioner`` (:928-952) and the string-
keyed ``state`` (``x0``, ``init_losses``, ``final_losses``, ``dampings``,
``cg_reasons``, ``num_cg_iters``, ``best_cg_iters``, ``learning_rates``).

What runs where
  * the PCG loop: ``pytorchhessianfree_amd.cg.cg`` -> HIP kernels, damping fused
    (``DampedCurvature``), diagonal preconditioner fused;
  * curvature products: ``curvature.GGNOperator`` / ``HessianOperator`` (graphs
    recorded once per step, multi-tensor gather into one HBM vector);
  * trial parameter writes ``theta = theta0 + alpha*step`` for LM damping,
    CG-backtracking, line search and the final update: one fused kernel on a
    persistent flat parameter arena (the reference re-concatenates and re-binds
    all parameters for each of them, optimizer.py:288-294, :349-350);
  * host control (LM rule, backtracking order, Armijo rule) is restated 1:1.

Extensions (keyword-only, default = reference behaviour)
  * ``process_group`` / ``shard_weight``: data parallelism over the batch.  Every
    rank calls ``step`` with the ``forward`` of ITS shard; loss, gradient and
    every curvature product are summed over ranks with weight ``shard_weight``
    (default ``1/world_size``, i.e. equal shards of a mean-reduced loss) -- the
    single-process accumulation of optimizer.py:677-684 turned into one
    all-reduce.  All ranks then run the identical PCG on identical data.
  * ``graph_matvec``: replay the local curvature product as a hipGraph.
  * ``get_preconditioner`` RETURNS the preconditioner (the reference computes it
    and drops it, optimizer.py:943-952 -- documented deviation).
"""

from contextlib import nullcontext
from warnings import warn

import torch

from . import curvature
from .cg import DampedCurvature, cg
from .cg_backtracking import cg_efficient_backtracking
from .linesearch import simple_linesearch
from .preconditioners import diag_EF_preconditioner
from .optimizer_acc import _Accumulation
from .optimizer_session import _SessionSteps, _SessionTrials  # noqa: F401
from .utils import ParameterArena


class HessianFree(_Accumulation, _SessionSteps, torch.optim.Optimizer):
    def __init__(
        self,
        params,
        curvature_opt="ggn",
        damping=1.0,
        adapt_damping=True,
        cg_max_iter=250,
        cg_decay_x0=0.95,
        use_cg_backtracking=True,
        lr=1.0,
        use_linesearch=True,
        verbose=False,
        *,
        process_group=None,
        shard_weight=None,
        graph_matvec=False,
        cache_acc_graphs=True,
    ):
        if curvature_opt not in ["hessian", "ggn"]:
            raise ValueError(f"Invalid curvature_opt = {curvature_opt}")
        if damping < 0.0:
            raise ValueError(f"Invalid damping = {damping}")
        self.adapt_damping = adapt_damping
        if damping == 0.0 and adapt_damping:
            self.adapt_damping = False
            warn("The damping is set to `0.0` and won't get adapted.")
        if cg_max_iter is not None and cg_max_iter < 1:
            raise ValueError(f"Invalid cg_max_iter: {cg_max_iter}")
        if lr < 0.0:
            raise ValueError(f"Invalid learning rate lr = {lr}")
        self.cg_decay_x0 = cg_decay_x0
        self.use_cg_backtracking = use_cg_backtracking
        self.use_linesearch = use_linesearch

        defaults = dict(curvature_opt=curvature_opt, damping=damping, cg_max_iter=cg_max_iter, lr=lr)
        super().__init__(params, defaults)
        if len(self.param_groups) != 1:
            raise ValueError("`HessianFree` does not support per-parameter options.")

        self.verbose = verbose
        self._params = self.param_groups[0]["params"]
        self._params_list = [p for p in self._params if p.requires_grad]
        self.device = self._params_list[0].device

        # data parallelism
        self.process_group = process_group
        if process_group is not None:
            world = torch.distributed.get_world_size(process_group)
            self.shard_weight = (1.0 / world) if shard_weight is None else float(shard_weight)
        else:
            self.shard_weight = 1.0 if shard_weight is None else float(shard_weight)
        self.graph_matvec = bool(graph_matvec)
        self.cache_acc_graphs = bool(cache_acc_graphs)

        self._acc_comm = process_group  # group `_acc` sums over (None: this process only)
        self._acc_comm_active = False   # True while `acc_step` runs its data-parallel `step`
        self._in_acc_step = False       # True while `acc_step` runs `step` at all
        self._session_decline = None    # why the persistent session was not taken (last refusal)
        self._acc_counts = {}           # id(datalist) -> samples over all ranks (per acc_step)
        self._arena = None
        self._cg = cg  # the HIP PCG; tests swap in the CPU oracle to check host logic
        # persistent engine session (session.py): one engine + graphs for all steps
        self._session = None
        self._session_failures = 0
        self._session_off = False
        # ... and its acc_step counterpart (session.AccumulatedSession: one engine per data chunk)
        self._acc_session = None
        self._acc_session_failures = 0
        self._acc_session_off = False
        # which path the last step() / acc_step() took and why a faster one was declined (`path_report`)
        self._paths = {"step": None, "acc_step": None}
        self._declines = {"step": None, "acc_step": None}
        self._slow_path_warned = set()

    # ------------------------------------------------------------------------
    # helpers
    # ------------------------------------------------------------------------
    @property
    def _group(self):
        # looked up on every use: ``load_state_dict`` REPLACES the param-group dicts
        # (the reference caches the dict at construction, optimizer.py:118, and would
        # keep adapting the damping of a stale one after a checkpoint restore)
        return self.param_groups[0]

    def _log(self, *a):
        if self.verbose:
            print(*a)

    def _reduce_scalar(self, value):
        """Weighted sum over ranks of a per-shard scalar (float in, float out)."""
        if self.process_group is None:
            return value
        t = torch.tensor([value * self.shard_weight], dtype=torch.float64, device=self.device)
        torch.distributed.all_reduce(t, group=self.process_group)
        return float(t.item())

    def _reduce_vector(self, vec):
        if self.process_group is not None:
            torch.distributed.all_reduce(vec, group=self.process_group)
        return vec

    def _flat(self, tensors):
        """parameters_to_vector replacement (+ shard weight)."""
        return curvature.flatten_into(tensors, self._params_list, scale=self.shard_weight)

    # the paths a step can take, fastest first
    PATHS = {
        "session": "persistent engine session (engine, product graph, PCG iteration graph kept across steps)",
        "acc-session": "accumulated engine session (acc_step on the fused engine, graphs kept across calls)",
        "engine-graphed": "fused engine rebuilt and re-captured every step (hipGraph replay per product)",
        "autograd-graphed": "autograd sweeps re-captured every step (hipGraph replay per product)",
        "eager": "eager launches (no hipGraph)",
        "user": "user-supplied gradient / product",
    }

    def path_report(self):
        """Which path the last ``step()`` / ``acc_step()`` took -- ``session`` | ``acc-session`` | ``engine-graphed`` |
        ``autograd-graphed`` | ``eager`` | ``user`` -- and, where a faster one exists and was declined, the reason
        (which layer / loss / parameter subset / shape the engine or the session refused).  All of the speed of this
        package lives in the engine and its sessions; this is where to look when a step is slower than expected."""
        out = {}
        for kind in ("step", "acc_step"):
            path = self._paths[kind]
            out[kind] = None if path is None else {
                "path": path, "what": self.PATHS[path], "declined": self._declines[kind]}
        sess = self._session
        if out["step"] is not None and sess is not None and getattr(sess, "group", None) is not None:
            # data parallel: which form of the product the session validated / measured on the communicator
            out["step"]["data_parallel"] = {
                "product": "two-phase (chunked / overlapped all-reduce)" if sess.split is not None
                           else "single graph + one compact all-reduce",
                "validation": getattr(sess, "mode_validation", None), "timing_ms": getattr(sess, "mode_timing", None)}
        return out

    def _note_path(self, kind, path, decline=None):
        """Record the path of this call; with ``graph_matvec=True`` a call that ends BELOW the session warns -- once
        per optimizer and kind of call, with the reason."""
        self._paths[kind] = path
        self._declines[kind] = decline if path not in ("session", "acc-session") else None
        top = "session" if kind == "step" else "acc-session"
        if (self.graph_matvec and path not in (top, "user") and self.device.type == "cuda"
                and kind not in self._slow_path_warned):
            self._slow_path_warned.add(kind)
            warn(f"HessianFree(graph_matvec=True).{kind}() runs on the slower path '{path}' ({self.PATHS[path]}), not "
                 f"on the {self.PATHS[top].split(' (')[0]}"
                 + (f": {decline}" if decline else "") + ".  See HessianFree.path_report().")

    def _ensure_arena(self):
        if self._arena is None:
            self._arena = ParameterArena(self._params)
        else:
            self._arena.ensure_bound()
        return self._arena

    # ------------------------------------------------------------------------
    # step
    # ------------------------------------------------------------------------
    def step(self, forward, grad=None, mvp=None, M_func=None, test_deterministic=False, *, _session=None):
        """One Hessian-free update; arguments as optimizer.py:126-180.  ``forward()``
        returns ``(loss, outputs)``; ``grad`` / ``mvp`` / ``M_func`` optionally
        override the gradient vector, the curvature product ``x -> B x`` and the
        preconditioner ``x -> M^-1 x``.  Returns the final loss (``None`` without
        line search unless ``verbose``)."""
        state = self.state
        state.setdefault("x0", None)
        for key in ("init_losses", "final_losses", "dampings", "cg_reasons", "num_cg_iters",
                    "best_cg_iters", "learning_rates"):
            state.setdefault(key, [])

        if self.verbose:
            print("\nInformation on parameters...")
            print("  Total number of parameters: ", sum(p.numel() for p in self._params))
            print("  Number of trainable parameters: ",
                  sum(p.numel() for p in self._params if p.requires_grad))
            print("  Device = ", self.device)

        arena = self._ensure_arena()
        if test_deterministic:
            self._test_forward_determinisitc(forward)

        # ---- loss, gradient, curvature operator (optimizer.py:216-247) ---------
        mvp, grad, init_loss, sess = self.linearise(forward, grad, mvp, _session=_session)
        self._log(f"\nInitial loss = {init_loss:.6f}")
        state["init_losses"].append(init_loss)

        if test_deterministic:
            self._test_mvp_deterministic(mvp)

        # ---- PCG (optimizer.py:256-281) -------------------------------------
        damping = self._group["damping"]
        state["dampings"].append(damping)
        x_iters, m_iters, cg_reason = self._cg(
            A=DampedCurvature(mvp, damping,
                              lockstep=self.process_group is not None or self._acc_comm_active),
            b=-grad,
            x0=state["x0"],
            M=M_func,
            max_iter=self._group["cg_max_iter"],
            martens_conv_crit=True,
            store_x_at_iters=None if self.use_cg_backtracking else [0],
            verbose=self.verbose,
        )
        state["cg_reasons"].append(cg_reason)
        state["num_cg_iters"].append(len(x_iters) - 1)
        step_vec = x_iters[-1]
        # warm start of the next solve: decayed FINAL iterate (optimizer.py:281)
        self._set_x0(self.cg_decay_x0 * x_iters[-1])

        # ---- target function on the flat arena (optimizer.py:288-294) -----------
        params_vec = arena.snapshot()

        if sess is not None:
            # trial points as graph replays on the session's static buffers, values cached and
            # read back in batches (what follows calls tfunc exactly as the reference does)
            trials = _SessionTrials(self, sess, arena, params_vec)
            trial, prefetch = trials.value, trials.prefetch
        else:
            prefetch = None

            @torch.no_grad()
            def trial(step, alpha):
                if alpha == 0.0:
                    arena.theta.copy_(params_vec)
                else:
                    arena.write(params_vec, step, alpha)
                return self._reduce_scalar(forward()[0].item())

        def tfunc(step):
            return trial(step, 1.0)

        tfunc.scaled = trial
        tfunc.prefetch = prefetch

        # ---- Levenberg-Marquardt damping (optimizer.py:299-306) ----------------
        assert x_iters[0] is not None and x_iters[-1] is not None
        if self.adapt_damping:
            if prefetch is not None:
                prefetch([(x_iters[0], 1.0), (x_iters[-1], 1.0)])  # both values with one read-back
            self._adapt_damping(
                f_0=tfunc(x_iters[0]), f_step=tfunc(x_iters[-1]),
                m_0=m_iters[0], m_step=m_iters[-1],
            )

        # ---- CG-backtracking (optimizer.py:311-318) ----------------------------
        if self.use_cg_backtracking:
            best_cg_iter, _ = cg_efficient_backtracking(f=tfunc, steps_list=x_iters,
                                                        verbose=self.verbose)
            state["best_cg_iters"].append(best_cg_iter)
            step_vec = x_iters[best_cg_iter]

        # ---- line search (optimizer.py:323-340) ---------------------------------
        lr = self._group["lr"]
        if not self.use_linesearch:
            self._log(f"\nConstant lr = {lr:.6f}")
            final_loss = None
        else:
            lr, final_loss = simple_linesearch(f=tfunc, f_grad_0=grad, step=step_vec,
                                               init_alpha=lr, verbose=self.verbose)
        state["learning_rates"].append(lr)

        # ---- parameter update (optimizer.py:349-350) ----------------------------
        self._log(f"\nParameter update with lr = {lr:.6f}")
        with torch.no_grad():
            if lr == 0.0:
                arena.theta.copy_(params_vec)
            else:
                arena.write(params_vec, step_vec, lr)

        if self.verbose:
            if final_loss is None:
                final_loss = self._reduce_scalar(forward()[0].item())
            state["final_losses"].append(final_loss)
            print(f"Initial loss = {init_loss:.6f} --> final loss = {final_loss:.6f}")
        return final_loss

    # ------------------------------------------------------------------------
    def linearise(self, forward, grad=None, mvp=None, *, _session=None):
        """Loss, gradient and curvature operator of one step (optimizer.py:216-247) -- exactly what ``step``
        hands to ``cg()``: returns ``(mvp, grad, initial loss, session)`` with ``session`` the persistent
        engine session when it serves this step (``mvp`` is then the session itself: product graph(s), the
        data-parallel all-reduce chunked and overlapped), else ``None`` (``mvp``: engine / autograd operator,
        hipGraph-replayed with ``graph_matvec``).  Public so that a caller who drives ``cg()`` himself
        (``bench.py``) measures the operator ``step`` uses, not one he built."""
        self._ensure_arena()
        curvature_opt = self._group["curvature_opt"]
        user_grad, user_mvp = grad is not None, mvp is not None
        holder = {"why": []}
        kind = "acc_step" if self._in_acc_step else "step"

        def setup():
            """Forward pass (+ gradient, + curvature operator).  Runs on the capture
            stream when the matvec is to be replayed as a hipGraph."""
            with torch.no_grad() if (user_grad and user_mvp) else nullcontext():
                loss, outputs = forward()
            holder["loss"] = loss
            grads = None
            if not user_grad:
                grads = torch.autograd.grad(
                    loss, self._params_list, create_graph=(curvature_opt == "hessian"),
                    retain_graph=True, allow_unused=True,
                )
                holder["grad"] = self._reduce_vector(self._flat(grads))
            if user_mvp:
                return None
            if curvature_opt == "hessian":
                return curvature.hessian_operator(
                    loss, outputs, self._params_list,
                    grad_with_graph=None if (grads is None or any(g is None for g in grads)) else grads,
                    weight=self.shard_weight, group=self.process_group, why=holder["why"])
            return curvature.ggn_operator(loss, outputs, self._params_list,
                                          weight=self.shard_weight, group=self.process_group, why=holder["why"])

        sess = _session  # (acc_step: the accumulated session has already been brought to this step's data)
        if sess is not None:
            self._note_path("acc_step", "acc-session")
            return sess, sess.gradient(), sess.base_loss, sess
        if (self.graph_matvec and not user_mvp and not user_grad
                and self.device.type == "cuda" and not self._session_off and self._cg is cg):
            sess, init_loss = self._session_step(forward)
        if sess is not None:
            mvp = sess
            grad = self._reduce_vector(sess.gradient())
            self._note_path("step", "session")
        else:
            if self.graph_matvec and not user_mvp and self.device.type == "cuda":
                mvp = curvature.maybe_graphed(setup, params=self._params_list)
            else:
                op = setup()
                mvp = mvp if user_mvp else op
            if not user_grad:
                grad = holder["grad"]
            init_loss = self._reduce_scalar(holder["loss"].item())
            if kind == "step":  # (acc_step's generic accumulation has reported itself: `_acc_session_step`)
                if user_mvp:
                    path = "user"
                elif isinstance(mvp, curvature.GraphedOperator):
                    path = "engine-graphed" if "engine" in getattr(getattr(mvp, "op", None), "mode", "") else "autograd-graphed"
                else:
                    path = "eager"
                decline = self._session_decline if self.graph_matvec and not user_mvp and not user_grad else None
                if holder["why"] and not decline:
                    decline = "; ".join(dict.fromkeys(holder["why"]))
                self._note_path("step", path, decline)
        return mvp, grad, init_loss, sess

    def _test_forward_determinisitc(self, forward):
        """Two forward passes must agree (optimizer.py:365-412); warns otherwise."""
        self._log("\nTest deterministic behavior of `forward`...")
        loss_1, out_1 = forward()
        loss_2, out_2 = forward()
        ok = True
        if out_1 is not None and out_2 is not None:
            same = torch.allclose(out_1, out_2)
            self._log("  Test outputs: " + ("passed" if same else "failed"))
            ok = ok and same
        same = torch.allclose(loss_1, loss_2)
        self._log("  Test loss values: " + ("passed" if same else "failed"))
        ok = ok and same
        if not ok:
            msg = "Non-determinisitc behaviour detected. Consider setting your "
            msg += "model to evaluation mode, i.e. `model.eval()`."
            warn(msg)
        else:
            self._log("  All tests passed")

    def _test_mvp_deterministic(self, mvp):
        """Two products with the same random vector must agree (optimizer.py:414-448)."""
        self._log("\nTest deterministic behavior of `mvp`...")
        x = torch.randn(sum(p.numel() for p in self._params_list),
                        dtype=self._params_list[0].dtype, device=self._params_list[0].device)
        x = x.to(self.device)
        first = mvp(x).clone()
        second = mvp(x)
        if not torch.allclose(first, second):
            self._log("  Test mvps: failed")
            msg = "Non-determinisitc behaviour detected. Consider setting your "
            msg += "model to evaluation mode, i.e. `model.eval()`."
            warn(msg)
        else:
            self._log("  Test mvps: passed\n  All tests passed")

    @staticmethod
    def _Hv(loss, params_list, vec):
        """``H vec`` on the flat vector (optimizer.py:450-455)."""
        return curvature.HessianOperator(loss, params_list)(vec)

    @staticmethod
    def _Gv(loss, outputs, params_list, vec):
        """``J^T H_L J vec`` on the flat vector (optimizer.py:457-462)."""
        return curvature.GGNOperator(loss, outputs, params_list)(vec)

    def _adapt_damping(self, f_0, f_step, m_0, m_step):
        """Levenberg-Marquardt rule (optimizer.py:464-506): ``rho`` = actual over
        predicted reduction; damping x3/2 if rho < 1/4, x2/3 if rho > 3/4."""
        rho = (f_step - f_0) / (m_step - m_0)
        if self.verbose:
            print("\nLM-heurisitc: Adapt damping...")
            print(f"  f_0    = {f_0:.6f}\n  f_step = {f_step:.6f}")
            print(f"  m_0    = {m_0:.6f}\n  m_step = {m_step:.6f}")
            print(f"  Reduction ratio rho = {rho:.6f}")
        if rho < 0.25:
            self._group["damping"] *= 3 / 2
        elif rho > 0.75:
            self._group["damping"] *= 2 / 3
        self._log(f"  Damping is set to {self._group['damping']:.6f}")
        if rho < 0:
            msg = "The reduction ratio `rho` is negative. This might result in "
            msg += "a bad cg-initialization in the next step."
            warn(msg)

    def _set_x0(self, new_x0):
        self.state["x0"] = new_x0
