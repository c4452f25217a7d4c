"""What ``HessianFree.step`` does on a persistent engine session (``session.EngineSession``): the step itself with
forward / gradient / trial losses as graph replays (reference optimizer.py:216-234, :288-350), the diagonal
empirical-Fisher preconditioner from the session's engine (``get_preconditioner``, optimizer.py:928-952 -- which here
RETURNS it) and the batched trial-loss evaluation ``_SessionTrials``.  Mixin of ``optimizer.HessianFree``."""

from warnings import warn

import torch

from .preconditioners import diag_EF_preconditioner


class _SessionSteps:
    # ------------------------------------------------------------------------
    def _session_step(self, forward):
        """Start a step on the persistent engine session (session.py), creating it on first use.
        Runs the caller's ``forward()`` ONCE: its loss value is the step's initial loss and must be
        reproduced by the session's own forward pass; its autograd graph names the targets.  Returns
        ``(session, initial loss)`` or ``(None, None)`` -- then the generic path runs (and after
        repeated failures the session is not tried again)."""
        if self._session_off:
            return None, None
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
                if sess is not None:
                    self._session_decline = "another rank's session was refused (the ranks decide together)"
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
            self._session_decline = ("`forward()` does not return (loss with an autograd graph, model output tensor): "
                                     "the session reads targets and loss structure off that graph")
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
            why = []
            if getattr(outputs, "_hf_model", None) is not None:
                sess = EngineSession.try_create(*args, hessian=hessian, why=why)
            else:
                why.append("the model is not a prepared one (modelprep.prepare_model(model, channels_last=True) "
                           "installs the layers the fused engine reads)")
            spec = sess.accepts(*args) if sess is not None else None
            if spec is None:
                self._session_decline = ("; ".join(dict.fromkeys(why)) if why else
                                         "a freshly built session does not accept this step's own forward pass")
                self._session_failures += 1
                if self._session_failures >= 2:
                    self._session_off = True
                    self._session_decline += " (refused twice: not tried again)"
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
            self._session_decline = (f"the session's forward pass gave loss {b!r} where `forward()` gave {a!r}"
                                     " (session ended: not tried again)")
            return None, None, None
        self._session_failures = 0
        return sess, a, b

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
        from .engine import loss_spec_of
        from .modelprep import session_forward
        from .session import _quadratic_signature

        self._ensure_arena()
        with session_forward(sess):  # (the forward pass is one replay of the session's graph)
            out = model(inputs)
        if out is not getattr(sess, "_override_out", None):
            return None  # (the session did not answer this forward pass: another mode / shape)
        loss = loss_func(out, targets)
        spec = loss_spec_of(loss, out, check_values=False)
        if (spec is None or spec["kind"] != eng.loss_spec["kind"] or spec["reduction"] != reduction
                or spec["reduction"] != eng.loss_spec["reduction"]
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
