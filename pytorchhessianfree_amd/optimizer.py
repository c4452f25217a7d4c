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
from .utils import ParameterArena


class HessianFree(torch.optim.Optimizer):
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
        holder = {}

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
                    weight=self.shard_weight, group=self.process_group)
            return curvature.ggn_operator(loss, outputs, self._params_list,
                                          weight=self.shard_weight, group=self.process_group)

        sess = _session  # (acc_step: the accumulated session has already been brought to this step's data)
        if sess is not None:
            return sess, sess.gradient(), sess.base_loss, sess
        if (self.graph_matvec and not user_mvp and not user_grad
                and self.device.type == "cuda" and not self._session_off and self._cg is cg):
            sess, init_loss = self._session_step(forward)
        if sess is not None:
            mvp = sess
            grad = self._reduce_vector(sess.gradient())
        else:
            if self.graph_matvec and not user_mvp and self.device.type == "cuda":
                mvp = curvature.maybe_graphed(setup, params=self._params_list)
            else:
                op = setup()
                mvp = mvp if user_mvp else op
            if not user_grad:
                grad = holder["grad"]
            init_loss = self._reduce_scalar(holder["loss"].item())
        return mvp, grad, init_loss, sess

    # ------------------------------------------------------------------------
    def _session_step(self, forward):
        """Start a step on the persistent engine session (session.py), creating it on first use.
        Runs the caller's ``forward()`` ONCE: its loss value is the step's initial loss and must be
        reproduced by the session's own forward pass; its autograd graph names the targets.  Returns
        ``(session, initial loss)`` or ``(None, None)`` -- then the generic path runs (and after
        repeated failures the session is not tried again)."""
        sess, a, b = self._session_step_local(forward)
        if self.process_group is not None:
            # one decision for all ranks: the session path and the generic path issue different
            # collectives, so a rank whose session was refused takes every rank with it
            ok = torch.tensor([1 if sess is not None else 0], dtype=torch.int32, device=self.device)
            torch.distributed.all_reduce(ok, op=torch.distributed.ReduceOp.MIN, group=self.process_group)
            if int(ok.item()) == 0:
                # EVERY rank switches the session off, whether it had one or not: a rank whose session was
                # merely refused this once would otherwise issue this all-reduce again on the next step while
                # its peers go straight to the generic path's gradient all-reduce
                self._session, self._session_off = None, True
                return None, None
        if sess is None:
            return None, None
        if getattr(sess, "mode_pending", False):  # (data parallel, once: single graph or chunked all-reduce)
            sess.choose_product_mode()
        sess.base_loss = self._reduce_scalar(b)
        return sess, self._reduce_scalar(a)

    def _session_step_local(self, forward):
        from .modelprep import session_forward
        from .session import EngineSession

        import os

        # From its second step on the session answers the model's forward pass itself, so comparing the caller's
        # loss with the session's says nothing about the forward pass any more.  Every K-th step (K =
        # ``HF_SESSION_VERIFY_EVERY``, default 16; ``HF_SESSION_VERIFY=1``: every step) the model therefore runs
        # its OWN forward pass and the session must reproduce its logits (1e-4) and loss (1e-5): a layer swapped,
        # frozen or re-configured behind the captured graphs shows up here instead of never
        every = 1 if os.environ.get("HF_SESSION_VERIFY") == "1" else int(os.environ.get("HF_SESSION_VERIFY_EVERY", "16"))
        verify = self._session is not None and every > 0 and self._session.steps % every == 0
        with session_forward(None if verify else self._session):  # (an existing session answers the forward pass)
            loss, outputs = forward()
        if not isinstance(outputs, torch.Tensor) or loss.grad_fn is None:
            self._session_off = True
            return None, None, None
        sess = self._session
        args = (loss, outputs, self._params_list, self.shard_weight, self.process_group)
        hessian = self._group["curvature_opt"] == "hessian"
        if sess is not None and sess.engine.hessian != hessian:
            sess = None
        spec = sess.accepts(*args) if sess is not None else None
        if spec is None:
            self._session = sess = None
            verify = False
            if getattr(outputs, "_hf_model", None) is not None:
                sess = EngineSession.try_create(*args, hessian=hessian)
            spec = sess.accepts(*args) if sess is not None else None
            if spec is None:
                self._session_failures += 1
                if self._session_failures >= 2:
                    self._session_off = True
                return None, None, None
            self._session = sess
        own = sess.begin_step(outputs, spec)
        drift = torch.zeros((), device=own.device)
        if verify:
            want = outputs.detach()
            drift = (sess.engine.logits - want).abs().max() / want.abs().max().clamp_min(1e-30)
        a, b, bad, drift = torch.stack([loss.detach().float().reshape(()), own.reshape(()),
                                        sess.engine.bad_targets.float().reshape(()), drift.float().reshape(())]).tolist()
        if bad or not abs(a - b) <= 1e-5 * max(1.0, abs(a)) or not drift <= 1e-4:
            warn(f"persistent engine session: its forward pass gives loss {b!r}, `forward()` gives {a!r}"
                 + (f" (logits differ by {drift:.1e} from the model's own forward pass)" if verify else "")
                 + "; using the generic path from now on")
            self._session, self._session_off = None, True
            return None, None, None
        self._session_failures = 0
        return sess, a, b

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

    # ------------------------------------------------------------------------
    # acc_step: loss / gradient / curvature accumulated over lists of mini-batches
    # ------------------------------------------------------------------------
    def acc_step(self, model, loss_func, loss_datalist, grad_datalist=None, mvp_datalist=None,
                 M_func=None, reduction="mean", test_deterministic=False):
        """optimizer.py:519-606.  With a process group every rank passes ITS data
        lists; ``mean`` weights are then ``N_chunk / N_total over all ranks``."""
        forward, grad, mvp, sess = self.acc_linearise(model, loss_func, loss_datalist, grad_datalist, mvp_datalist,
                                                      reduction)
        # `step` must not re-weight what `_acc` already reduced over ranks
        saved = (self.process_group, self.shard_weight)
        self.process_group, self.shard_weight = None, 1.0
        self._acc_comm_active = self._acc_comm is not None
        try:
            return self.step(forward=forward, grad=grad, mvp=mvp, M_func=M_func,
                             test_deterministic=test_deterministic, _session=sess)
        finally:
            self.process_group, self.shard_weight = saved
            self._acc_comm_active = False
            self._acc_counts = {}

    def acc_linearise(self, model, loss_func, loss_datalist, grad_datalist=None, mvp_datalist=None,
                      reduction="mean"):
        """What ``acc_step`` hands to ``step``: ``(forward, grad, mvp, session)`` (optimizer.py:519-606).  With the
        accumulated engine session (``session`` not ``None``) loss, gradient, trial losses and products are graph
        replays over one fused engine per chunk and ``grad`` / ``mvp`` are ``None`` (``step`` takes them from the
        session); else the generic accumulation.  Public for callers who drive ``cg()`` themselves (bench.py)."""
        grad_datalist = loss_datalist if grad_datalist is None else grad_datalist
        mvp_datalist = loss_datalist if mvp_datalist is None else mvp_datalist
        curvature_opt = self._group["curvature_opt"]
        if reduction not in ["mean", "sum"]:
            raise ValueError(f"Invalid reduction {reduction}")
        self._count_samples(loss_datalist, grad_datalist, mvp_datalist)

        def forward():
            return self._acc_loss(model, loss_func, loss_datalist, reduction), None

        sess = self._acc_session_step(model, loss_func, (loss_datalist, grad_datalist, mvp_datalist), reduction,
                                      curvature_opt)
        if sess is not None:
            return forward, None, None, sess
        grad = self._acc_grad(model, loss_func, grad_datalist, reduction)
        if self.cache_acc_graphs:
            mvp = self._acc_mvp_cached(model, loss_func, mvp_datalist, curvature_opt, reduction)
        else:

            def mvp(x):
                return self._acc_mvp(model, loss_func, mvp_datalist, curvature_opt, reduction, x)

            if self._acc_comm is not None:
                mvp.collective = True
        return forward, grad, mvp, None

    def _acc_session_step(self, model, loss_func, lists, reduction, curvature_opt):
        """The accumulated engine session for this ``acc_step`` call, brought to its data (created on first use,
        reused while model, loss, list structure and chunk shapes stay the same), or ``None`` -- then the generic
        accumulation runs (and after repeated refusals the session is not tried again).  Under data
        parallelism the decision is taken for all ranks together (one MIN all-reduce)."""
        if not (self.graph_matvec and self.device.type == "cuda" and not self._acc_session_off and self._cg is cg):
            return None
        sess = self._acc_session_step_local(model, loss_func, lists, reduction, curvature_opt)
        if self._acc_comm is not None:
            ok = torch.tensor([1 if sess is not None else 0], dtype=torch.int32, device=self.device)
            torch.distributed.all_reduce(ok, op=torch.distributed.ReduceOp.MIN, group=self._acc_comm)
            if int(ok.item()) == 0:
                self._acc_session, self._acc_session_off = None, True
                return None
            if sess is not None:  # (the count-weighted loss summed over the ranks)
                sess.base_loss = sess.reduce_losses(sess.loss_buf.reshape(1)).tolist()[0]
        return sess

    def _acc_session_step_local(self, model, loss_func, lists, reduction, curvature_opt):
        import os

        from .session import AccumulatedSession, _NoEngine

        self._ensure_arena()
        memo = {}

        def dev(t):  # (chunks the lists share stay shared: the session keys its engines on tensor identity)
            if id(t) not in memo:
                memo[id(t)] = t.to(self.device)
            return memo[id(t)]

        try:
            dlists = tuple([(dev(x), dev(t)) for x, t in dl] for dl in lists)
        except (TypeError, ValueError, AttributeError):
            self._acc_session_off = True
            return None
        counts = [self._total_count(dl) for dl in lists]
        hessian = curvature_opt == "hessian"
        args = (model, loss_func, dlists, self._params_list, reduction, counts, hessian, self._acc_comm)
        sess = self._acc_session
        slots = sess.accepts(*args) if sess is not None else None
        if slots is None:
            self._acc_session = sess = None
            sess = AccumulatedSession.try_create(model, loss_func, dlists, self._params_list, reduction, counts,
                                                 hessian=hessian, group=self._acc_comm)
            slots = sess.accepts(*args) if sess is not None else None
            if slots is None:
                self._acc_session_failures += 1
                if self._acc_session_failures >= 2:
                    self._acc_session_off = True
                return None
            self._acc_session = sess
        every = 1 if os.environ.get("HF_SESSION_VERIFY") == "1" else int(os.environ.get("HF_SESSION_VERIFY_EVERY", "16"))
        try:
            sess.begin_step(slots, verify=every > 0 and sess.steps > 0 and sess.steps % every == 0,
                            reduce=self._acc_comm is None)
        except _NoEngine:
            warn("accumulated engine session: it no longer reproduces the model (or a target is outside the "
                 "classes); using the generic accumulation from now on")
            self._acc_session, self._acc_session_off = None, True
            return None
        self._acc_session_failures = 0
        return sess

    def _count_samples(self, *datalists):
        """Samples per data list, summed over the ranks of a data-parallel run -- ONCE
        per ``acc_step`` (one all-reduce for all lists), not once per product: the
        counts are constants of the step, and reading them back inside ``mvp`` would
        put a second collective and a host sync into every PCG iteration."""
        # one entry per ARGUMENT (loss, gradient, curvature list), whether or not some of them are
        # the same object: every rank then reduces a vector of the same length even if the ranks
        # alias their lists differently
        local = [float(sum(targets.shape[0] for _, targets in dl)) for dl in datalists]
        if self._acc_comm is not None:
            t = torch.tensor(local, dtype=torch.float64, device=self.device)
            torch.distributed.all_reduce(t, group=self._acc_comm)
            local = t.tolist()
        self._acc_counts = {id(dl): cnt for dl, cnt in zip(datalists, local)}

    def _total_count(self, datalist):
        count = self._acc_counts.get(id(datalist))
        if count is None:  # called outside acc_step (test_reduction, direct use)
            count = float(sum(targets.shape[0] for _, targets in datalist))
            if self._acc_comm is not None:
                t = torch.tensor([count], dtype=torch.float64, device=self.device)
                torch.distributed.all_reduce(t, group=self._acc_comm)
                count = float(t.item())
        return count

    def _acc(self, model, loss_func, datalist, device, with_grad, init_result, eval_mb, reduction):
        """Generic accumulator (optimizer.py:608-684): ``sum_k N_k q_k / sum_k N_k``
        (``mean``) or ``sum_k q_k`` (``sum``) over the chunks -- and over ranks."""
        if reduction not in ["mean", "sum"]:
            raise ValueError(f"Invalid reduction {reduction}")
        count = self._total_count(datalist)
        total = init_result
        for inputs, targets in datalist:
            n_chunk = targets.shape[0]
            inputs, targets = inputs.to(device), targets.to(device)
            with nullcontext() if with_grad else torch.no_grad():
                outputs = model(inputs)
                loss = loss_func(outputs, targets)
            piece = eval_mb(loss, outputs)
            if reduction == "mean":
                total += n_chunk * piece
            else:
                total += piece
        if self._acc_comm is not None:  # the same sum, continued over the ranks: ONE collective
            if not isinstance(total, torch.Tensor):
                total = torch.tensor(float(total), device=device)
            torch.distributed.all_reduce(total, group=self._acc_comm)
        return total / count if reduction == "mean" else total

    def _acc_loss(self, model, loss_func, datalist, reduction):
        """optimizer.py:686-723."""
        return self._acc(model, loss_func, datalist, device=self.device, with_grad=False,
                         init_result=0.0, eval_mb=lambda loss, outputs: loss.detach(),
                         reduction=reduction)

    def _zeros_flat(self):
        ref = self._params_list[0]
        return torch.zeros(sum(p.numel() for p in self._params_list), dtype=ref.dtype,
                           device=ref.device)

    def _acc_grad(self, model, loss_func, datalist, reduction):
        """optimizer.py:725-765."""

        def eval_mb(loss, outputs):
            g = torch.autograd.grad(loss, self._params_list, allow_unused=True)
            return curvature.flatten_into(g, self._params_list)

        return self._acc(model, loss_func, datalist, device=self.device, with_grad=True,
                         init_result=self._zeros_flat(), eval_mb=eval_mb, reduction=reduction)

    def _acc_mvp(self, model, loss_func, datalist, curvature_opt, reduction, x):
        """optimizer.py:767-814."""

        def eval_mb(loss, outputs):
            if curvature_opt == "hessian":
                return curvature.HessianOperator(loss, self._params_list)(x)
            return curvature.GGNOperator(loss, outputs, self._params_list)(x)

        return self._acc(model, loss_func, datalist, device=self.device, with_grad=True,
                         init_result=self._zeros_flat(), eval_mb=eval_mb, reduction=reduction)

    def _acc_mvp_cached(self, model, loss_func, datalist, curvature_opt, reduction):
        """``_acc_mvp`` with the per-chunk forward graphs built ONCE per step instead
        of once per chunk per product (the reference rebuilds them on every call and
        says so, optimizer.py:537-540).  Same weighted sum
        in the same order.  Costs the memory of all chunk graphs; disable with
        ``cache_acc_graphs=False`` for batches that only fit chunk by chunk."""
        if reduction not in ["mean", "sum"]:
            raise ValueError(f"Invalid reduction {reduction}")
        chunks = []
        for inputs, targets in datalist:
            inputs, targets = inputs.to(self.device), targets.to(self.device)
            outputs = model(inputs)
            loss = loss_func(outputs, targets)
            if curvature_opt == "hessian":
                op = curvature.HessianOperator(loss, self._params_list)
            else:
                op = curvature.GGNOperator(loss, outputs, self._params_list)
            chunks.append((targets.shape[0], op))

        count = self._total_count(datalist)  # over all ranks; constant for the step

        def mvp(x):
            total = self._zeros_flat()
            for n_chunk, op in chunks:
                piece = op(x)
                if reduction == "mean":
                    total += n_chunk * piece
                else:
                    total += piece
            if self._acc_comm is not None:  # one collective, no host read-back
                torch.distributed.all_reduce(total, group=self._acc_comm)
            return total / count if reduction == "mean" else total

        if self._acc_comm is not None:
            mvp.collective = True  # cg() must use its lockstep stop rule
        return mvp

    # ------------------------------------------------------------------------
    def test_reduction(self, model, loss_func, datalist, reduction):
        """Accumulated vs whole-batch loss / gradient / product must agree
        (``rtol=1e-2, atol=1e-4``), else ``RuntimeError`` (optimizer.py:817-926)."""
        self._log(f"\nTest reduction {reduction}...")
        msg = "This test is only meaningful for a data list with at least two entries."
        assert len(datalist) > 1, msg
        x = torch.randn(sum(p.numel() for p in self._params_list),
                        dtype=self._params_list[0].dtype, device=self._params_list[0].device)
        x = x.to(self.device)
        curvature_opt = self._group["curvature_opt"]
        saved, self._acc_comm = self._acc_comm, None  # a local self-test: no communication
        try:
            acc_loss = self._acc_loss(model, loss_func, datalist, reduction)
            acc_grad = self._acc_grad(model, loss_func, datalist, reduction)
            acc_mvp = self._acc_mvp(model, loss_func, datalist, curvature_opt, reduction, x)
        finally:
            self._acc_comm = saved

        ref_inputs = torch.cat([d[0] for d in datalist], dim=0).to(self.device)
        ref_targets = torch.cat([d[1] for d in datalist], dim=0).to(self.device)
        ref_outputs = model(ref_inputs)
        ref_loss = loss_func(ref_outputs, ref_targets)
        ref_grad = curvature.flatten_into(
            torch.autograd.grad(ref_loss, self._params_list, create_graph=True, allow_unused=True),
            self._params_list)
        if curvature_opt == "ggn":
            ref_mvp = self._Gv(ref_loss, ref_outputs, self._params_list, x)
        else:
            ref_mvp = self._Hv(ref_loss, self._params_list, x)

        passed = True
        for name, ref, acc in [("loss values", ref_loss, acc_loss), ("gradients", ref_grad, acc_grad),
                               ("mvps", ref_mvp, acc_mvp)]:
            acc_t = acc if isinstance(acc, torch.Tensor) else torch.tensor(acc)
            ok = torch.allclose(acc_t.to(ref.dtype), ref.detach(), rtol=1e-2, atol=1e-4)
            self._log(f"  Test {name}: " + ("passed" if ok else "failed"))
            passed = passed and ok
        if not passed:
            error_msg = f"Inconsistent results for reduction {reduction}. "
            error_msg += "This could also be the result of non-deterministic "
            error_msg += "behavior or simply due to using the GPU."
            raise RuntimeError(error_msg)
        self._log("  All tests passed")

    def get_preconditioner(self, model, loss_func, inputs, targets, reduction, exponent=None,
                           use_backpack=True):
        """Diagonal empirical-Fisher preconditioner at the CURRENT damping
        (optimizer.py:928-952).  Unlike the reference, the result is returned.  With a persistent engine
        session for ``model`` (from the second ``step`` on) the diagonal comes from ONE adjoint sweep of the engine
        plus per-sample weight-gradient launches (``engine.diag_ef``) instead of one backward pass per sample
        (``use_backpack=False``) / a batched per-sample-gradient pass (``True``): the same quantity."""
        diag = self._engine_diag_ef(model, loss_func, inputs, targets, reduction)
        if diag is not None:
            from .preconditioners import diag_to_preconditioner

            damping = self._group["damping"]
            return (diag_to_preconditioner(diag, damping) if exponent is None
                    else diag_to_preconditioner(diag, damping, exponent))
        return diag_EF_preconditioner(model, loss_func, inputs, targets, reduction,
                                      damping=self._group["damping"], exponent=exponent,
                                      use_backpack=use_backpack)

    def _engine_diag_ef(self, model, loss_func, inputs, targets, reduction):
        """``sum_i g_i^2`` (/ N) on the session's engine, or ``None`` (no session for this model / shape / loss, train
        mode, data parallelism): the caller then takes the autograd construction."""
        import os

        sess = self._session
        if (sess is None or self.process_group is not None
                or reduction not in ("mean", "sum")):
            return None
        eng = sess.engine
        if (model is not eng.model_ref or eng.train_bn or eng.loss_spec is None
                or not isinstance(inputs, torch.Tensor) or tuple(inputs.shape) != tuple(eng.x_in.shape)):
            return None
        from .engine import ce_loss_spec
        from .modelprep import session_forward
        from .session import _quadratic_signature

        self._ensure_arena()
        with session_forward(sess):  # (the forward pass is one replay of the session's graph)
            out = model(inputs)
        if out is not getattr(sess, "_override_out", None):
            return None  # (the session did not answer this forward pass: another mode / shape)
        loss = loss_func(out, targets)
        spec = ce_loss_spec(loss, out, check_values=False)
        if (spec is None or spec["reduction"] != reduction or spec["reduction"] != eng.loss_spec["reduction"]
                or _quadratic_signature(spec) != _quadratic_signature(eng.loss_spec)
                or tuple(spec["targets"].shape) != tuple(eng._targets.shape)):
            return None
        with torch.no_grad():
            eng.set_targets(spec["targets"])
            eng._loss_head()
            if bool(eng.bad_targets):
                return None
            # (one graph replay: 5.8 ms against 74 ms for the per-sample autograd loop, round 4)
            return sess.diag_ef(reduction)

class _SessionTrials:
    """``tfunc`` of optimizer.py:288-294 on a persistent engine session: a trial point
    ``theta0 + alpha*step`` is one ``hf_axpy_out`` launch on the flat arena plus ONE graph launch
    (weights into kernel layout, forward pass, loss); the loss stays in a device array until a value
    is needed, so that several trial points cost one device->host read.  Values are cached per
    (step vector, alpha): LM damping, CG-backtracking and the line search ask for some points more
    than once (the last CG iterate; the back-tracked step at ``alpha = 1``; the base point)."""

    def __init__(self, opt, sess, arena, params_vec):
        self.opt, self.sess, self.arena, self.base = opt, sess, arena, params_vec
        self.cache = {(0, 0.0): sess.base_loss}  # alpha = 0: the loss at theta0 (this step's forward replay)
        self.pending = []
        self._keep = []

    @staticmethod
    def _key(step, alpha):
        return (0, 0.0) if alpha == 0.0 else (step.data_ptr(), float(alpha))

    @torch.no_grad()
    def prefetch(self, points, needed=None):
        """Enqueue the evaluation of ``[(step, alpha), ...]`` (no host synchronisation).  ``needed``: only the
        first ``needed`` points are certainly consumed, the rest is speculation (the next back-tracking / Armijo
        candidate).  With train-mode BatchNorm every evaluated point moves the running statistics -- as every
        ``forward()`` of the reference does -- so there only the points the reference itself would evaluate
        are evaluated (no speculation)."""
        if needed is not None and getattr(getattr(self.sess, "engine", None), "train_bn", False):
            points = points[:needed]
        for step, alpha in points:
            key = self._key(step, alpha)
            if step is None or key in self.cache or any(k == key for k, _ in self.pending):
                continue
            if len(self.pending) >= self.sess.losses.numel():
                self.flush()
            self.arena.write(self.base, step, alpha)
            self.sess.forward_loss(len(self.pending))
            self.pending.append((key, len(self.pending)))
            self._keep.append(step)  # (the key is the vector's address: it must not be recycled within the step)

    def flush(self):
        if not self.pending:
            return
        vals = self.sess.losses[: len(self.pending)]
        opt = self.opt
        reducer = getattr(self.sess, "reduce_losses", None)
        if reducer is not None:  # (acc_step: count-weighted over the chunks already; summed over the ranks here)
            vals = reducer(vals)
        elif opt.process_group is not None:  # weighted sum over the ranks' shards, all values at once
            vals = vals.double() * opt.shard_weight
            torch.distributed.all_reduce(vals, group=opt.process_group)
        for (key, _), val in zip(self.pending, vals.tolist()):
            self.cache[key] = val
        self.pending = []

    def value(self, step, alpha):
        key = self._key(step, alpha)
        if key not in self.cache:
            self.prefetch([(step, alpha)])
            self.flush()
        return self.cache[key]
