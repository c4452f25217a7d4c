"""``acc_step`` of the reference (optimizer.py:519-606) and the accumulation helpers ``_acc*`` (:608-814),
``test_reduction`` (:817-926): loss / gradient / curvature products accumulated over lists of mini-batches -- on the
accumulated engine session where the model family is covered (``session.AccumulatedSession``), else by the generic
accumulation with per-chunk operators cached per step.  Mixin of ``optimizer.HessianFree``."""

from contextlib import nullcontext
from warnings import warn

import torch

from . import curvature
from .cg import cg


class _Accumulation:
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
        self._in_acc_step = True
        try:
            return self.step(forward=forward, grad=grad, mvp=mvp, M_func=M_func,
                             test_deterministic=test_deterministic, _session=sess)
        finally:
            self.process_group, self.shard_weight = saved
            self._acc_comm_active = False
            self._in_acc_step = False
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
        if not (self.graph_matvec and self.device.type == "cuda" and self._cg is cg):
            self._note_path("acc_step", "eager")  # (not asked for graphs / no GPU: nothing was declined)
            return None
        if self._acc_session_off:
            self._note_path("acc_step", "eager", self._acc_decline)
            return None
        sess = self._acc_session_step_local(model, loss_func, lists, reduction, curvature_opt)
        if self._acc_comm is not None:
            ok = torch.tensor([1 if sess is not None else 0], dtype=torch.int32, device=self.device)
            torch.distributed.all_reduce(ok, op=torch.distributed.ReduceOp.MIN, group=self._acc_comm)
            if int(ok.item()) == 0:
                if sess is not None:
                    self._acc_decline = "another rank's accumulated session was refused (the ranks decide together)"
                self._acc_session, self._acc_session_off = None, True
                sess = None
            if sess is not None:  # (the count-weighted loss summed over the ranks)
                sess.base_loss = sess.reduce_losses(sess.loss_buf.reshape(1)).tolist()[0]
        if sess is None:
            self._note_path("acc_step", "eager", self._acc_decline)
        else:
            self._note_path("acc_step", "acc-session")
        return sess

    _acc_decline = None  # why the accumulated session was not taken (last refusal)

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
            self._acc_decline = "the data lists are not lists of (inputs, targets) tensor pairs"
            return None
        counts = [self._total_count(dl) for dl in lists]
        hessian = curvature_opt == "hessian"
        args = (model, loss_func, dlists, self._params_list, reduction, counts, hessian, self._acc_comm)
        sess = self._acc_session
        slots = sess.accepts(*args) if sess is not None else None
        if slots is None:
            self._acc_session = sess = None
            why = []
            sess = AccumulatedSession.try_create(model, loss_func, dlists, self._params_list, reduction, counts,
                                                 hessian=hessian, group=self._acc_comm, why=why)
            slots = sess.accepts(*args) if sess is not None else None
            if slots is None:
                self._acc_decline = ("; ".join(dict.fromkeys(why)) if why else
                                     "a freshly built accumulated session does not accept this call's own data lists")
                self._acc_session_failures += 1
                if self._acc_session_failures >= 2:
                    self._acc_session_off = True
                    self._acc_decline += " (refused twice: not tried again)"
                return None
            self._acc_session = sess
        every = 1 if os.environ.get("HF_SESSION_VERIFY") == "1" else int(os.environ.get("HF_SESSION_VERIFY_EVERY", "16"))
        try:
            sess.begin_step(slots, verify=every > 0 and sess.steps > 0 and sess.steps % every == 0,
                            reduce=self._acc_comm is None)
        except _NoEngine as exc:
            warn(f"accumulated engine session: {exc.reason}; using the generic accumulation from now on")
            self._acc_session, self._acc_session_off = None, True
            self._acc_decline = exc.reason + " (session ended: not tried again)"
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
