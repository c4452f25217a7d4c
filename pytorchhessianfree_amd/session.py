"""A persistent Newton-step session: ONE fused curvature engine and FOUR hipGraphs that serve every
``HessianFree.step()`` of a training run (reference ``optimizer.py:126-363``).

What the reference (and this package's generic path) redoes on every step:

* builds the autograd graph of ``forward()`` and differentiates it for the gradient
  (optimizer.py:216-234);
* sets up the curvature product from that graph (optimizer.py:237-247) -- here: engine construction,
  capture of the product graph, instantiation of the per-iteration PCG graph: ~13 ms per step;
* evaluates ``forward()`` eagerly for every trial point of the Levenberg-Marquardt rule,
  CG-backtracking and the line search (``tfunc``, optimizer.py:288-294: ~8 forward passes of
  ~2.4 ms of host time each on the ResNet-18 workload, a device->host sync each).

The engine's buffers are static and its own forward pass (``FusedGGNEngine.forward_own``) refills
them in place, so the session captures, ONCE:

    G_wT     the (I, H, W, O) weight copies of the data-gradient convolutions      (per step)
    G_fwd    W halves of all [W | v_W] operands <- theta, forward pass, softmax, loss   (per step
             and per trial point: ``theta = theta0 + alpha*step`` is one ``hf_axpy_out`` launch
             on the optimizer's flat arena, then ONE graph launch; losses stay on the device until
             a phase needs them)
    G_grad   the gradient by one adjoint sweep of the engine                        (per step)
    G_prod   the GGN product ``input_buffer -> output_buffer``                      (per PCG iteration,
             cloned by ``cg()`` into its one-launch iteration graph, which also survives)

Per step the host then issues a handful of graph launches plus the user's own ``forward()`` once:
its loss value must agree with the session's (a model / loss the session does not reproduce makes the
optimizer fall back to the generic path), and its autograd graph tells the session the targets.
Supported: models the engine covers (prepared ResNet families, NHWC fp32, eval-mode BatchNorm) with
a plain softmax cross-entropy loss.
"""

import os

import torch

from . import _lib
from .curvature import GraphedOperator
from .engine import FusedGGNEngine, loss_spec_of


def _runs_beside(cand, cur):
    """Whether work on stream ``cand`` executes while ``cur`` is busy: a few ms of streaming updates on ``cur``, a
    trivial kernel on ``cand``; concurrent iff the trivial one has finished while the updates have not.  The
    scratch vector is sized from the free memory (256 MiB at most, ~1/16 of what is free at least 4 MiB; more passes
    over a smaller vector keep ``cur`` busy for the same few ms) so that a nearly full GPU cannot fail a
    data-parallel step here; ``None`` if even that cannot be allocated (the caller then takes a plain stream)."""
    dev = torch.device("cuda", torch.cuda.current_device())
    long_done, short_done = torch.cuda.Event(), torch.cuda.Event()
    try:
        free = torch.cuda.mem_get_info(dev)[0]
        n = max(1 << 20, min(1 << 26, int(free // 64)))  # elements (x 4 bytes)
        big = torch.zeros(n, device=dev)
        scratch = torch.zeros(64, device=dev)
    except RuntimeError:  # (out of memory)
        return None
    torch.cuda.synchronize()
    with torch.cuda.stream(cur):
        for _ in range(max(24, 24 * (1 << 26) // n)):
            big.add_(1.0)
        long_done.record(cur)
    with torch.cuda.stream(cand):
        scratch.add_(1.0)
        short_done.record(cand)
    short_done.synchronize()
    beside = not long_done.query()
    long_done.synchronize()
    return beside


def _concurrent_stream(cur, candidates=8):
    """A side stream whose work runs BESIDE ``cur``'s.  HIP streams of one priority share four hardware queues; a
    side stream that lands on the compute stream's queue (one pool stream in four) runs in line with the sweep
    instead of beside it: its all-reduce hides nothing (stand-in kernels: scripts/experiments/stream_handover.hip,
    profiles/r05_stream_handover.jsonl; in the session: profiles/r05_two_phase_side_stream.jsonl).  Probed, not
    assumed: the first of a few pool streams that demonstrably overlaps with ``cur``; the last one tried if none
    does.  (On a 1-rank group, where the collective is the identity, an in-line side stream is the CHEAPER one --
    no cross-queue dependency, ~725 instead of ~790 us per iteration; it is not taken for that.)"""
    cand, beside = None, False
    for _ in range(candidates):
        cand = torch.cuda.Stream()
        beside = _runs_beside(cand, cur)
        if beside or beside is None:
            break
    return cand, bool(beside)


class _TwoPhaseProduct:
    """The data-parallel engine product as TWO hipGraphs with the all-reduce chunked by stage and overlapped
    with the sweep (the ``result += N * mb_result`` of optimizer.py:677-684 across GPUs, once per PCG iteration):

        G_a  tangent sweep, head, adjoint sweep of the late blocks, gather of their gradients
             -> the vector's suffix is final; its staged part (live taps of tensors with dead ones) is gathered
        side stream / second communicator:   all-reduce(tail pieces: the dense run in place + the staging
                                             vector, ~80 % of the bytes; ONE grouped launch)   <- overlaps G_b
        G_b  the rest of the adjoint sweep
        compute stream:                      all-reduce(head pieces: in place in the full vector)
        wait for the side stream; scatter the staged sums back into the full vector; K1-K3 graph

    Two plain hipGraphs and events BETWEEN graph launches (fork / join nodes inside one graph cost ~45 us
    each on this stack; so does every dependency between two hardware queues, whatever carries it -- events,
    stream-ordered value writes / waits, an event node inside a chained launch: DESIGN.md section 7).  On 2 ranks the
    result is bitwise the single all-reduce's (a + b in either
    order); on more ranks every rank still receives identical sums -- which is all the lockstep rule of
    ``cg()`` needs.  ResNet-18 on 28x28 inputs: layer3 + layer4 + fc are 14.2 of the 16.9 MB that travel and
    are final after ~60 % of the product, so ~0.3 ms of sweep remain to hide their all-reduce.

    Shared by ``EngineSession`` (what ``HessianFree.step(process_group=...)`` runs) and
    ``ChunkedEngineOperator`` (the bare operator).  Needs ``engine, input_buffer, output_buffer, stream,
    group`` on the instance."""

    split = None
    _side = None        # the stream the tail's all-reduce runs on (probed: work on it runs BESIDE the compute stream's)
    side_runs_beside = None

    @staticmethod
    def plan_phases(eng, tail_fraction=None):
        """``(cut block, first late parameter, flat offset)`` when the engine's product splits into an early
        part and a late suffix AND has a compact layout of the entries that travel; else ``None`` (the
        engine is left as it was: single graph, single compact / plain all-reduce)."""
        if tail_fraction is None:
            tail_fraction = 0.7
        if os.environ.get("HF_CHUNKED_ALLREDUCE", "auto") == "0" or not hasattr(eng, "phase_split"):
            return None
        if getattr(eng, "hessian", False) or getattr(eng, "train_bn", False):
            return None
        split = eng.phase_split(tail_fraction)
        if split is None:
            return None
        had = eng.__dict__.pop("_live_segs", None), getattr(eng, "_seg_break", None)
        eng._seg_break = split[1]
        if eng._live_segments() is None or eng._seg_cut is None:
            eng._seg_break = had[1]
            eng.__dict__.pop("_live_segs", None)
            return None
        return split

    def _phase_a(self):
        eng = self.engine
        eng.local_phase_a(self.input_buffer, self.output_buffer, self.split)
        eng._live_copy(self.output_buffer, False, part="tail")

    def _phase_b(self):
        eng = self.engine
        eng.local_phase_b(self.output_buffer, self.split)
        eng._live_copy(self.output_buffer, False, part="head")

    def _capture_phases(self):
        """(on ``self.stream``, warmed up) the two graphs of one product."""
        self.g_a = torch.cuda.CUDAGraph(keep_graph=True)
        with torch.cuda.graph(self.g_a, stream=self.stream):
            self._phase_a()
        self.g_b = torch.cuda.CUDAGraph(keep_graph=True)
        with torch.cuda.graph(self.g_b, stream=self.stream, pool=self.g_a.pool()):
            self._phase_b()
        self.g_a.instantiate()
        self.g_b.instantiate()

    def replay_phases(self):
        self.g_a.replay()
        self.g_b.replay()

    def reduce_phases(self):
        """One data-parallel product: reads ``input_buffer``, leaves the summed product in ``output_buffer``."""
        from . import distributed as hfdist

        eng, group = self.engine, self.group
        if group is None:
            return self.replay_phases()
        _inject_fault("two_phase", group)
        head = eng._reduce_pieces(self.output_buffer, "head")
        tail = eng._reduce_pieces(self.output_buffer, "tail")
        cur = torch.cuda.current_stream()
        side_comm = hfdist.side_comm(tail[0], group)
        if side_comm is not None and self._side is None:
            # (the probe's verdict is kept: ``side_runs_beside`` -- False: no pool stream was seen to overlap with the
            # compute stream, or the probe could not allocate its scratch; the tail's all-reduce then hides less)
            self._side, self.side_runs_beside = _concurrent_stream(cur)
            self._ev_a, self._ev_t = torch.cuda.Event(), torch.cuda.Event()
        works = []
        self.g_a.replay()
        if side_comm is not None:
            self._ev_a.record(cur)
            self._side.wait_event(self._ev_a)
            with torch.cuda.stream(self._side):
                side_comm.all_reduce_sum_multi(tail)
                self._ev_t.record(self._side)
        else:
            works = [torch.distributed.all_reduce(piece, group=group, async_op=True) for piece in tail]
        self.g_b.replay()
        hfdist.all_reduce_sum_multi(head, group)
        if side_comm is not None:
            cur.wait_event(self._ev_t)
        for work in works:
            work.wait()
        eng._live_copy(self.output_buffer, True)
        _inject_fault("two_phase", group, self.output_buffer)


class EngineSession(_TwoPhaseProduct):
    mode = ("persistent session: hipGraph replay of the " + FusedGGNEngine.mode
            + "; engine, product graph and PCG iteration graph kept across steps, forward pass / "
              "gradient / trial losses as graph replays on static buffers")

    # ------------------------------------------------------------------------------------
    @classmethod
    def try_create(cls, loss, outputs, params, weight=1.0, group=None, hessian=False, why=None):
        """A session for the model that produced ``outputs`` (``None`` if the engine does not cover
        it or the loss is not a plain softmax cross-entropy; ``why``, a list, then receives the reason)."""
        why = [] if why is None else why
        if os.environ.get("HF_SESSION", "1") == "0":
            why.append("the persistent session is switched off (HF_SESSION=0)")
            return None
        if not torch.cuda.is_available():
            why.append("no GPU")
            return None
        if loss_spec_of(loss, outputs) is None:
            why.append(_loss_decline(loss, outputs))
            return None
        holder = {}

        def builder():
            holder["eng"] = FusedGGNEngine.try_build(loss, outputs, list(params), weight=weight, group=group,
                                                     hessian=hessian, why=why)
            return holder["eng"]

        sess = cls.__new__(cls)
        try:
            sess._build(builder, params)
        except _NoEngine as exc:
            if not why:
                why.append(exc.reason)
            return None
        return sess

    def _build(self, builder, params):
        cur = torch.cuda.current_stream()
        dev = torch.cuda.current_device()
        if dev not in GraphedOperator._streams:
            GraphedOperator._streams[dev] = torch.cuda.Stream()
        self.stream = GraphedOperator._streams[dev]
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream), torch.no_grad():
            with torch.enable_grad():
                eng = builder()
            if eng is None or eng.loss_spec is None:
                cur.wait_stream(self.stream)
                raise _NoEngine("the fused engine does not cover this model" if eng is None else
                                "the engine's own forward pass does not reproduce this train-mode model, or the loss "
                                "is neither a plain softmax cross-entropy nor a mean-squared error")
            self.op = self.engine = eng
            self.n, self.group, self.params = eng.n, eng.group, eng.params
            self.weight = eng.weight
            f32 = dict(dtype=torch.float32, device=eng.dev)
            self.input_buffer = torch.zeros(self.n, **f32)
            self.output_buffer = torch.empty(self.n, **f32)
            self.grad_buffer = torch.empty(self.n, **f32)
            self.losses = torch.zeros(64, **f32)
            # data parallel: the product as two graphs, its all-reduce chunked by stage and overlapped
            self.split = self.plan_phases(eng) if eng.group is not None else None
            # warm-up of everything that will be captured (allocator, lazy initialisations)
            eng.refresh_weights(transposed=True)
            eng.forward_own(update_running=False)
            eng.gradient(self.grad_buffer)
            eng.local(self.input_buffer, out=self.output_buffer)
            if self.split is not None:
                for _ in range(2):
                    self._phase_a()
                    self._phase_b()
        self.stream.synchronize()
        with torch.no_grad():
            self.g_wT = self._capture(lambda: eng.refresh_weights(transposed=True))
            self.g_fwd = self._capture(lambda: eng.forward_own(refresh=True))
            # train-mode BatchNorm: a forward pass moves the running statistics.  The step that CREATES the
            # session has already run the model itself (the stock layers moved them): its refresh replays a
            # twin of the graph that leaves them alone
            self.g_fwd_still = (self._capture(lambda: eng.forward_own(refresh=True, update_running=False))
                                if eng.train_bn else self.g_fwd)
            self.g_grad = self._capture(lambda: eng.gradient(self.grad_buffer))
            # (always the single product graph; under data parallelism also the two-phase pair, and
            # ``choose_product_mode`` keeps whichever is faster with THIS communicator on THIS machine)
            self.graph = self._capture(lambda: eng.local(self.input_buffer, out=self.output_buffer), keep=True)
            self._split_plan = self.split
            if self.split is not None:
                self._capture_phases()
        # (data parallel: ``choose_product_mode`` validates the forms on the live communicator -- and, policy
        # ``auto``, times them -- before the first solve)
        self.mode_pending = eng.group is not None
        self.mode_validation = self.mode_timing = None
        cur.wait_stream(self.stream)
        torch.cuda.synchronize()
        self.calls = 0
        self.steps = 0
        self._cache = {}
        self._signature = self._signature_of(eng)
        self._layers = eng.layer_signature()
        self._diag_graphs = {}

    def _capture(self, fn, keep=False):
        g = torch.cuda.CUDAGraph(keep_graph=True) if keep else torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=self.stream):
            fn()
        if keep:
            g.instantiate()
        return g

    # ---- validity ------------------------------------------------------------------------
    @staticmethod
    def _signature_of(eng):
        return (id(eng.model_ref), tuple(id(p) for p in eng.params), tuple(eng.x_in.shape), eng.weight,
                eng.loss_spec["reduction"])

    def accepts(self, loss, outputs, params, weight, group):
        """The step that produced ``(loss, outputs)`` is one this session reproduces: same model
        object, parameters, input shape, rank weight and loss structure.  Returns the loss
        description (reduction, targets) or ``None``."""
        eng = self.engine
        ref = getattr(outputs, "_hf_model", None)
        model = ref() if ref is not None else None
        x = getattr(outputs, "_hf_input", None)
        if model is not eng.model_ref or x is None or group is not self.group:
            return None
        if eng.layer_signature() != self._layers:
            return None  # (a BatchNorm switched mode, eps / momentum changed, a layer was swapped)
        if tuple(x.shape) != tuple(eng.x_in.shape) or x.dtype != torch.float32 or x.device != eng.x_in.device:
            return None
        if len(params) != len(eng.params) or any(a is not b for a, b in zip(params, eng.params)):
            return None
        if float(weight) != eng.weight or eng._flat_params is None:
            return None
        if eng._flat_params.data_ptr() != eng.params[0].data_ptr():
            return None  # the parameters moved (arena rebuilt): a fresh session is needed
        spec = loss_spec_of(loss, outputs, check_values=False)
        if (spec is None or spec["kind"] != eng.loss_spec["kind"]
                or spec["reduction"] != eng.loss_spec["reduction"]):
            return None
        if _quadratic_signature(spec) != _quadratic_signature(eng.loss_spec):
            return None  # (another regulariser: coefficients or tensors differ)
        if tuple(spec["targets"].shape) != tuple(eng._targets.shape):
            return None
        return spec

    # ---- per step ------------------------------------------------------------------------
    def forward_override(self, model, x):
        """The model's forward pass answered by the session (``modelprep.session_forward``): new
        batch + current parameters -> every static buffer of the engine, ONE graph replay; returns
        the logits as a leaf that requires grad, or ``None`` when this call is not the session's
        (another model / shape / mode) and the model must run itself."""
        eng = self.engine
        if (model is not eng.model_ref or eng.layer_signature() != self._layers or not torch.is_grad_enabled()
                or not isinstance(x, torch.Tensor) or tuple(x.shape) != tuple(eng.x_in.shape)
                or x.dtype != torch.float32 or x.device != eng.x_in.device
                or eng._flat_params is None or eng._flat_params.data_ptr() != eng.params[0].data_ptr()):
            return None
        with torch.no_grad():
            eng.set_batch(x.detach())
            self.g_wT.replay()
            self.g_fwd.replay()
            out = eng.logits.clone()
        out.requires_grad_(True)
        self._override_out = out
        return out

    def begin_step(self, outputs, spec):
        """New batch + current parameters: refresh every static buffer of the engine (input, im2col,
        weights in kernel layout, activations, ReLU masks, pooling positions, probabilities) and
        return the session's own loss value as a 0-dim device tensor."""
        eng = self.engine
        with torch.no_grad():
            if outputs is getattr(self, "_override_out", None):
                # the forward pass of this very batch was the session's own replay: only the
                # targets are new
                eng.set_targets(spec["targets"])
                eng._loss_head()
            else:
                eng.set_batch(getattr(outputs, "_hf_input").detach(), spec["targets"])
                self.g_wT.replay()
                self.g_fwd_still.replay()  # (the model's own forward pass produced `outputs`)
        self._override_out = None
        self._cache = {}
        self.steps += 1
        self._first_order_fresh = False
        return eng.loss_buf

    _first_order_fresh = False  # (Hessian engines: the gradient sweep of this step has left its cotangents)

    def gradient(self):
        """``weight * grad`` at the parameters of the last forward replay (static buffer)."""
        self.g_grad.replay()
        self._first_order_fresh = True
        return self.grad_buffer

    def forward_loss(self, slot):
        """Forward pass at the CURRENT parameters; the loss goes to ``losses[slot]`` (device)."""
        self.g_fwd.replay()
        self.losses[slot].copy_(self.engine.loss_buf)
        return self.losses[slot]

    def diag_ef(self, reduction):
        """The diagonal empirical Fisher of the engine's current batch (``engine.diag_ef``: ~20 launches per sample)
        as ONE graph replay; captured on first use.  Returns a fresh vector."""
        key = ("diag", reduction)
        if key not in self._diag_graphs:
            eng = self.engine
            cur = torch.cuda.current_stream()
            self.stream.wait_stream(cur)
            with torch.cuda.stream(self.stream), torch.no_grad():
                buf = torch.empty(self.n, dtype=torch.float32, device=eng.dev)
                eng.diag_ef(reduction, out=buf)  # warm-up (allocations, lazy buffers)
            self.stream.synchronize()
            with torch.no_grad():
                g = self._capture(lambda: eng.diag_ef(reduction, out=buf))
            cur.wait_stream(self.stream)
            self._diag_graphs[key] = (g, buf)
        g, buf = self._diag_graphs[key]
        g.replay()
        return buf.clone()

    # ---- operator interface of cg() (see curvature.GraphedOperator) -------------------------
    def raw_graph(self):
        return self.graph.raw_cuda_graph()

    def replay_local(self):
        if self.split is None:
            self.graph.replay()
        else:
            self.replay_phases()

    def replay_and_reduce(self):
        """One product over all ranks: ``input_buffer`` -> summed product in ``output_buffer``."""
        if self.split is None:
            _inject_fault("single_graph", self.group)
            self.graph.replay()
            self.engine.reduce(self.output_buffer, self.group)
            _inject_fault("single_graph", self.group, self.output_buffer)
        else:
            self.reduce_phases()

    def choose_product_mode(self, reps=8):
        """Data parallel, once per session, COLLECTIVE (every rank of the group calls it at the same point --
        ``HessianFree`` does, after the ranks have agreed to use the session).

        1. VALIDATION on the live communicator (``mode_validation``): ONE product of a fixed pseudo-random vector in
           each form the session holds.  Every rank must hold the same bits (MIN / MAX all-reduce of two 64-bit digests
           of the result): the lockstep rule of ``cg()`` -- every rank takes the same branch without talking --
           rests on exactly that.  The single-graph form failing it is an error (nothing below it can work); the
           two-phase form failing it, or differing from the single-graph form by more than 1e-6 of its max-norm, is
           dropped with a warning and the session stays on the single graph.
        2. ``HF_CHUNKED_ALLREDUCE=auto`` (default) TIMING: a few products with the single graph + one compact
           all-reduce and with the two-phase chunked / overlapped all-reduce -- on THIS communicator, THIS machine --
           and keep the faster (two-phase only if it gains >= 3 %).  The chunked form hides most of the all-reduce
           behind the sweep but costs a graph launch, a stream hand-over and a gather more; which wins depends on the
           link (measured on one MI355X with a 1-rank RCCL group: single 1 438-1 452 matvecs/s, two-phase 1 296;
           projected at 8 GPUs over xGMI: two-phase).  One decision for all ranks (MAX
           all-reduce of the timings)."""
        import time
        from warnings import warn

        self.mode_pending = False
        if self.group is None:
            return self.split is not None
        dist = torch.distributed
        dev = self.engine.dev
        policy = os.environ.get("HF_CHUNKED_ALLREDUCE", "auto")
        cands = [("single_graph", None)]
        if self._split_plan is not None:
            cands.append(("two_phase", self._split_plan))
        # ---- 1. every form: identical on all ranks, the forms equal to rounding
        saved_input = self.input_buffer.clone()
        gen = torch.Generator(device=dev).manual_seed(20240229)  # (the same vector on every rank)
        self.input_buffer.copy_(torch.randn(self.n, device=dev, generator=gen))
        if getattr(self.engine, "hessian", False) and not self._first_order_fresh:
            self.g_grad.replay()
            self._first_order_fresh = True
        weights = torch.arange(self.n, device=dev, dtype=torch.int64) % 1021 + 1
        report, results = {}, {}
        for name, split in cands:
            self.split = split
            self.replay_and_reduce()
            out = self.output_buffer
            bits = out.view(torch.int32).to(torch.int64)
            digest = torch.stack([bits.sum(), (bits * weights).sum()])
            lo, hi = digest.clone(), digest.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
            report[name + "_identical_on_all_ranks"] = bool(torch.equal(lo, hi))
            results[name] = out.clone()
        del weights
        self.input_buffer.copy_(saved_input)
        if not report["single_graph_identical_on_all_ranks"]:
            self.mode_validation = report
            raise RuntimeError("data-parallel curvature product: the ranks hold DIFFERENT sums after the all-reduce "
                               "(single product graph + one all-reduce); the lockstep PCG cannot run on this communicator")
        two_phase_ok = len(cands) == 2
        if two_phase_ok:
            ref = results["single_graph"]
            rel = ((results["two_phase"] - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).reshape(1).double()
            dist.all_reduce(rel, op=dist.ReduceOp.MAX, group=self.group)
            report["two_phase_vs_single_graph"] = float(rel.item())
            two_phase_ok = report["two_phase_identical_on_all_ranks"] and float(rel.item()) <= 1e-6
            if not two_phase_ok:
                warn("data-parallel curvature product: the two-phase (chunked / overlapped all-reduce) form "
                     + ("gives different sums on different ranks" if not report["two_phase_identical_on_all_ranks"]
                        else f"differs from the single-graph form by {float(rel.item()):.1e} of its max-norm")
                     + " on this communicator; keeping the single product graph + one compact all-reduce")
        report["two_phase_kept_as_candidate"] = bool(two_phase_ok)
        self.mode_validation = report
        del results
        self.split = None
        if not two_phase_ok:
            return False
        if policy != "auto":  # (forced two-phase: "0" never gets a split plan)
            self.split = self._split_plan
            return True
        # ---- 2. timing
        sync = torch.zeros(2, dtype=torch.float64, device=dev)
        times = []
        for _name, split in cands:
            self.split = split
            for _ in range(3):
                self.replay_and_reduce()
            dist.all_reduce(sync, group=self.group)  # (every rank starts the timed replays together)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                self.replay_and_reduce()
            torch.cuda.synchronize()
            times.append((time.perf_counter() - t0) / reps)
        t = torch.tensor(times, dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        times = t.tolist()
        self.mode_timing = {name + "_ms": v * 1e3 for (name, _s), v in zip(cands, times)}
        self.split = self._split_plan if times[1] < 0.97 * times[0] else None
        return self.split is not None

    def reduce(self, t):
        return self.engine.reduce(t, self.group)

    @property
    def reduce_bytes(self):
        return self.engine.reduce_bytes

    def local(self, v, out=None):
        if v.data_ptr() != self.input_buffer.data_ptr():
            self.input_buffer.copy_(v)
        if not self._first_order_fresh and getattr(self.engine, "hessian", False):
            # a Hessian product reads the step's first-order cotangents: a caller who takes products right after
            # ``begin_step`` (``step`` itself takes the gradient first) gets them from a gradient replay here
            self.g_grad.replay()
            self._first_order_fresh = True
        self.replay_local()
        if out is not None:
            out.copy_(self.output_buffer)
            return out
        return self.output_buffer

    def __call__(self, v, out=None):
        self.calls += 1
        if self.group is None:
            return self.local(v, out)
        if v.data_ptr() != self.input_buffer.data_ptr():
            self.input_buffer.copy_(v)
        self.replay_and_reduce()
        if out is not None:
            out.copy_(self.output_buffer)
            return out
        return self.output_buffer


def _loss_decline(loss, outputs):
    """Why ``ce_loss_spec`` did not recognise the loss, in the user's terms."""
    fn = getattr(loss, "grad_fn", None)
    name = fn.name() if fn is not None else "no autograd node"
    if not isinstance(outputs, torch.Tensor) or outputs.dim() != 2:
        return "the model output is not a [batch, classes] tensor"
    return ("the loss is not a plain softmax cross-entropy on the model output (F.cross_entropy / nn.CrossEntropyLoss "
            "with class-index targets, no class weights, no label smoothing, no ignored target) nor a mean-squared "
            f"error against constant targets of the outputs' shape (the loss ends in {name}): the session evaluates "
            "loss, gradient and loss Hessian in closed form for these two only")


def _quadratic_signature(spec):
    return tuple((float(c), tuple(id(w) for w in ws)) for c, ws in (spec.get("quadratic") or ()))


class ChunkedEngineOperator(_TwoPhaseProduct):
    """The bare data-parallel engine operator with the chunked / overlapped all-reduce (``_TwoPhaseProduct``) for
    callers that drive ``cg()`` themselves; ``HessianFree.step(process_group=...)`` gets the same product from
    its ``EngineSession``."""

    mode = ("2 hipGraphs per product over the " + FusedGGNEngine.mode + "; all-reduce chunked by stage, the late "
            "layers' share overlapped with the rest of the adjoint sweep")

    def __init__(self, builder, params=None, tail_fraction=0.7):
        cur = torch.cuda.current_stream()
        dev = torch.cuda.current_device()
        if dev not in GraphedOperator._streams:
            GraphedOperator._streams[dev] = torch.cuda.Stream()
        self.stream = GraphedOperator._streams[dev]
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            eng = builder()
            if not isinstance(eng, FusedGGNEngine):
                raise TypeError("ChunkedEngineOperator needs the fused curvature engine")
            self.split = self.plan_phases(eng, tail_fraction)
            if self.split is None:
                raise TypeError("the model's parameters do not split into an early part and a late suffix with a "
                                "compact layout")
            self.op = self.engine = eng
            self.n, self.group, self.params = eng.n, eng.group, eng.params
            f32 = dict(dtype=torch.float32, device=eng.dev)
            self.input_buffer = torch.zeros(self.n, **f32)
            self.output_buffer = torch.zeros(self.n, **f32)
            with torch.no_grad():
                for _ in range(2):
                    self._phase_a()
                    self._phase_b()
        self.stream.synchronize()
        with torch.no_grad():
            self._capture_phases()
        cur.wait_stream(self.stream)
        torch.cuda.synchronize()
        self.calls = 0

    @property
    def reduce_bytes(self):
        return self.engine.reduce_bytes

    def raw_graph(self):  # (two graphs: cg() fuses K1-K3 only, which needs no product graph)
        return None

    def replay_local(self):
        self.replay_phases()

    def reduce(self, t):
        return self.engine.reduce(t, self.group)

    def replay_and_reduce(self):
        self.reduce_phases()

    def local(self, v, out=None):
        if v.data_ptr() != self.input_buffer.data_ptr():
            self.input_buffer.copy_(v)
        self.replay_local()
        if out is not None:
            out.copy_(self.output_buffer)
            return out
        return self.output_buffer

    def __call__(self, v, out=None):
        self.calls += 1
        if self.group is None:
            return self.local(v, out)
        if v.data_ptr() != self.input_buffer.data_ptr():
            self.input_buffer.copy_(v)
        self.replay_and_reduce()
        if out is not None:
            out.copy_(self.output_buffer)
            return out
        return self.output_buffer


class AccumulatedSession:
    """``HessianFree.acc_step()`` on the fused engine (reference optimizer.py:519-606, :608-684, :767-814): loss,
    gradient and every curvature product accumulated over lists of data chunks, without building a forward
    graph per chunk and product as the reference does (it says so itself, optimizer.py:537-540).

    One fused curvature engine per DISTINCT chunk of the three data lists (chunks that the lists share -- the
    default: one list for everything -- share their engine), every buffer static, and FOUR hipGraphs for the
    whole lists, kept across ``acc_step`` calls while the lists keep their shapes:

        G_wT    the (I, H, W, O) weight copies of every engine                              (per step)
        G_fwd   forward pass + loss of every chunk of the LOSS list (and of the chunks whose activations the
                gradient / curvature lists need), loss = sum_k N_k loss_k / sum_k N_k        (per step, per trial point)
        G_grad  one adjoint sweep per chunk of the GRADIENT list, summed                   (per step)
        G_prod  one product per chunk of the CURVATURE list, each already weighted N_k / sum N (mean) or 1
                (sum), summed by one gather launch -- cloned into ``cg()``'s one-launch-per-iteration graph

    The chunks' sweeps are independent until the final sum: they are captured on parallel branches of the
    graph (fork / join by events during capture), so that two latency-bound sweeps of half the batch overlap
    instead of queueing (measured: 1 288 matvecs/s against 855 one after the other; train-mode BatchNorm always runs the
    chunks in sequence -- they all move the same running statistics).  Everything is the package's own
    deterministic kernels: two ``acc_step`` calls on the same data are bitwise equal.

    Under data parallelism (``process_group``) every rank holds ITS lists; the counts are totals over all ranks
    and the summed product / gradient / losses are all-reduced once more (compact layout of the engine)."""

    @property
    def mode(self):
        if len(self.engines) == 1 and self.merged:
            return ("accumulated engine session: the chunks carry one per-sample weight and the model does not couple "
                    "samples, so they run as ONE batch on one fused curvature engine (" + FusedGGNEngine.mode + "); engine, "
                    "graphs and PCG iteration graph kept across acc_step calls")
        return ("accumulated engine session: one fused curvature engine per data chunk"
                + (" (chunks of equal per-sample weight merged)" if self.merged else "")
                + (" on parallel graph branches" if self.parallel else ", in sequence")
                + ", weighted sum by one gather launch; engine, graphs and PCG iteration graph kept across acc_step calls")

    # ------------------------------------------------------------------------------------
    @classmethod
    def try_create(cls, model, loss_func, lists, params, reduction, counts, hessian=False, group=None, why=None):
        """``lists = (loss_datalist, grad_datalist, mvp_datalist)`` on the device; ``counts`` their total sample
        counts (over all ranks).  ``None`` when the engine does not cover the model / loss (``why``, a list, then
        receives the reason)."""
        why = [] if why is None else why
        if os.environ.get("HF_ACC_SESSION", "1") == "0":
            why.append("the accumulated session is switched off (HF_ACC_SESSION=0)")
            return None
        if not torch.cuda.is_available():
            why.append("no GPU")
            return None
        if not getattr(model, "_hf_engine_hooks", False):
            why.append("the model is not a prepared one (modelprep.prepare_model(model, channels_last=True) installs "
                       "the layers the fused engine reads)")
            return None
        sess = cls.__new__(cls)
        sess._why = why
        try:
            sess._build(model, loss_func, lists, list(params), reduction, counts, hessian, group)
        except _NoEngine as exc:
            if not why:
                why.append(exc.reason)
            return None
        return sess

    @staticmethod
    def _plan(lists):
        """Distinct chunks (by identity of their tensors) and the slots each list uses."""
        slots, index, roles = [], {}, []
        for dl in lists:
            idx = []
            for inputs, targets in dl:
                key = (id(inputs), id(targets))
                if key not in index:
                    index[key] = len(slots)
                    slots.append((inputs, targets))
                idx.append(index[key])
            roles.append(tuple(idx))
        return slots, roles

    @staticmethod
    def _merge_groups(model, slots, roles):
        """Which distinct chunks may run as ONE batch on ONE engine.  The accumulated quantities are
        ``sum_k w_k q_k`` with ``q_k`` a mean / sum over the samples of chunk k (optimizer.py:677-684): chunks that
        appear in the same lists the same number of times carry the same weight PER SAMPLE (``1 / sum N`` resp. 1), so
        for a model that does not couple the samples of a batch their concatenation IS the accumulation -- the
        reference's own test states it (tests/test_optimizer_acc.py:116-175: [7, 8] chunks == one batch of 15).  Not
        merged: a model with a train-mode BatchNorm or an active dropout layer (per-chunk statistics / masks are part
        of the reference's result), chunks that differ in more than the batch size.  Returns lists of slot indices,
        in order of first appearance."""
        coupled = any((isinstance(m, torch.nn.modules.batchnorm._BatchNorm) and m.training)
                      or (isinstance(m, torch.nn.modules.dropout._DropoutNd) and m.training and m.p > 0)
                      for m in model.modules())
        groups, index = [], {}
        for k, (x, t) in enumerate(slots):
            key = (tuple(sum(1 for j in r if j == k) for r in roles), tuple(x.shape[1:]), x.dtype, tuple(t.shape[1:]),
                   t.dtype) if not coupled else k
            if key not in index:
                index[key] = len(groups)
                groups.append([])
            groups[index[key]].append(k)
        return groups

    def _merged(self, slots):
        """The data of the engines: per group of chunks their concatenation (a group of one: the chunk itself)."""
        return [slots[g[0]] if len(g) == 1 else
                (torch.cat([slots[k][0] for k in g]), torch.cat([slots[k][1] for k in g])) for g in self.groups]

    def _build(self, model, loss_func, lists, params, reduction, counts, hessian, group):
        slots, roles = self._plan(lists)
        if not slots or any(len(r) == 0 for r in roles):
            raise _NoEngine("an empty data list")
        self.model, self.loss_func, self.reduction, self.hessian = model, loss_func, reduction, bool(hessian)
        self.chunk_roles, self.chunk_shapes = roles, [tuple(x.shape) for x, _ in slots]
        for x, t in slots:
            if not (isinstance(x, torch.Tensor) and isinstance(t, torch.Tensor) and x.dim() >= 1 and t.dim() in (1, 2)
                    and t.shape[0] == x.shape[0]):
                raise _NoEngine("a data chunk is not (float32 inputs, class-index or [batch, outputs] targets)")
        # chunks that carry the same per-sample weight in every list run as ONE batch on ONE engine (round 6: the
        # default call -- one list for loss, gradient and curvature -- is then a single engine on the whole batch:
        # 1 450+ instead of 1 280 matvecs/s for chunks [16, 16], no graph branches)
        self.groups = self._merge_groups(model, slots, roles)
        self.merged = any(len(g) > 1 for g in self.groups)
        of_slot = {k: gi for gi, g in enumerate(self.groups) for k in g}
        roles = [tuple(gi for gi, g in enumerate(self.groups) for _ in range(sum(1 for j in r if j == g[0])))
                 for r in roles]
        slots = self._merged(slots)
        del of_slot
        self.roles, self.counts = roles, tuple(float(c) for c in counts)
        self.shapes = [tuple(x.shape) for x, _ in slots]
        self.group, self.params = group, params
        cur = torch.cuda.current_stream()
        dev = torch.cuda.current_device()
        if dev not in GraphedOperator._streams:
            GraphedOperator._streams[dev] = torch.cuda.Stream()
        self.stream = GraphedOperator._streams[dev]
        self.stream.wait_stream(cur)
        engines = []
        with torch.cuda.stream(self.stream), torch.no_grad():
            for x, t in slots:
                if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32 and t.dim() in (1, 2)):
                    cur.wait_stream(self.stream)
                    raise _NoEngine("a data chunk is not (float32 inputs, class-index or [batch, outputs] targets)")
                with torch.enable_grad():
                    out = model(x)
                    loss = loss_func(out, t)
                    spec = loss_spec_of(loss, out) if isinstance(out, torch.Tensor) and out.dim() == 2 else None
                    eng = None
                    if spec is None:
                        self._why.append(_loss_decline(loss, out))
                    elif spec["reduction"] != reduction:
                        self._why.append(f"the loss function reduces by '{spec['reduction']}', acc_step was asked for "
                                         f"reduction='{reduction}'")
                    else:
                        # (group=None: the ranks' sum is taken once, after the chunks' sum)
                        eng = FusedGGNEngine.try_build(loss, out, params, weight=1.0, group=None, hessian=hessian,
                                                       why=self._why)
                if eng is None or eng.loss_spec is None:
                    cur.wait_stream(self.stream)
                    raise _NoEngine("the fused engine does not cover this model / loss")
                engines.append(eng)
                del out, loss
            self.engines = engines
            self._layers = [e.layer_signature() for e in engines]
            # (the compact all-reduce layout -- which kernel taps can meet data -- is engine[0]'s: it depends on the
            # chunks' spatial shape, so all chunks must share everything but the batch size)
            if any(tuple(sh[1:]) != tuple(self.shapes[0][1:]) for sh in self.shapes):
                cur.wait_stream(self.stream)
                raise _NoEngine("the data chunks differ in more than their batch size: "
                                + ", ".join(str(sh) for sh in self.shapes))
            e0 = engines[0]
            self.engine = e0
            self.n, self.dev = e0.n, e0.dev
            self.train_bn = any(e.train_bn for e in engines)
            self.parallel = len(engines) > 1 and not self.train_bn
            if self.parallel:  # (one level of graph branches: the chunks'; no fork inside a forked branch)
                for e in engines:
                    e._extras_allowed = False
            f32 = dict(dtype=torch.float32, device=self.dev)
            k_all = len(engines)
            self.input_buffer = torch.zeros(self.n, **f32)
            self.output_buffer = torch.empty(self.n, **f32)
            self.grad_buffer = torch.empty(self.n, **f32)
            self._parts = torch.empty((k_all, self.n), **f32)   # per-chunk partial products / gradients
            self._loss_ks = list(dict.fromkeys(roles[0]))  # (trial points: only these engines run)
            self._lossvec = torch.zeros(len(self._loss_ks), **f32)
            w = torch.zeros(len(self._loss_ks), **f32)
            for k in roles[0]:
                w[self._loss_ks.index(k)] += float(self.shapes[k][0]) if reduction == "mean" else 1.0
            self._loss_w = w
            self.loss_buf = torch.zeros((), **f32)
            self.losses = torch.zeros(64, **f32)
            self._side = [torch.cuda.Stream() for _ in range(k_all - 1)] if self.parallel else []
            self._events = [torch.cuda.Event() for _ in range(2 * k_all)]
            # which engines each graph touches
            grad_set = list(dict.fromkeys(roles[1]))
            if hessian:  # (a Hessian engine's products read the first-order cotangents its gradient sweep keeps)
                grad_set += [k for k in dict.fromkeys(roles[2]) if k not in grad_set]
            self._grad_set, self._mvp_set = grad_set, list(dict.fromkeys(roles[2]))
            self._fwd_set = list(range(k_all))

            def role_weights(role):  # (a chunk listed twice in one list counts twice)
                out = {}
                for k in roles[role]:
                    wk = float(self.shapes[k][0]) / self.counts[role] if reduction == "mean" else 1.0
                    out[k] = out.get(k, 0.0) + wk
                return out

            self._w_grad, self._w_mvp = role_weights(1), role_weights(2)
            # warm-up of everything that will be captured
            self._refresh()
            self._forward(self._fwd_set, update_running=False)
            self._gradient()
            self._product()
        self.stream.synchronize()
        with torch.no_grad():
            self.g_wT = self._capture(self._refresh)
            # every engine (a step's linearisation point) / the loss list's engines only (trial points)
            self.g_fwd_all = self._capture(lambda: self._forward(self._fwd_set, update_running=True))
            self.g_fwd = (self.g_fwd_all if len(self._loss_ks) == k_all
                          else self._capture(lambda: self._forward(self._loss_ks, update_running=True)))
            self.g_fwd_still = (self._capture(lambda: self._forward(self._fwd_set, update_running=False))
                                if self.train_bn else self.g_fwd_all)
            self.g_grad = self._capture(self._gradient)
            self.graph = self._capture(self._product, keep=True)
        cur.wait_stream(self.stream)
        torch.cuda.synchronize()
        self.calls = 0
        self.steps = 0
        self.base_loss = None
        self._fresh = True  # (the model's own forward passes of the creating step have moved the running statistics)

    def _capture(self, fn, keep=False):
        g = torch.cuda.CUDAGraph(keep_graph=True) if keep else torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=self.stream):
            fn()
        if keep:
            g.instantiate()
        return g

    # ---- the four bodies ---------------------------------------------------------------------
    def _fork_join(self, ks, fn):
        """``fn(k)`` for every engine index of ``ks``: on parallel branches (the current stream + side streams,
        forked and joined by events -- inside a capture these become the graph's branches) or in sequence."""
        ks = list(ks)
        if not self.parallel or len(ks) < 2:
            for k in ks:
                fn(k)
            return
        cur = torch.cuda.current_stream()
        fork = self._events[0]
        fork.record(cur)
        for j, k in enumerate(ks[1:]):
            st = self._side[j]
            st.wait_event(fork)
            with torch.cuda.stream(st):
                fn(k)
                self._events[1 + j].record(st)
        fn(ks[0])
        for j in range(len(ks) - 1):
            cur.wait_event(self._events[1 + j])

    def _refresh(self):
        self._fork_join(self._fwd_set, lambda k: self.engines[k].refresh_weights(transposed=True))

    def _forward(self, ks, update_running=True):
        self._fork_join(ks, lambda k: self.engines[k].forward_own(refresh=True, update_running=update_running))
        torch.stack([self.engines[k].loss_buf for k in self._loss_ks], out=self._lossvec)
        val = torch.dot(self._lossvec, self._loss_w)
        if self.reduction == "mean":
            val = val / self.counts[0]
        self.loss_buf.copy_(val)

    def _sum_parts(self, ks, out):
        """``out = sum of the leading rows of the parts``: one gather launch (fixed order: repeatable)."""
        _lib.pack_ex(out, [self._parts[0]], {}, {0: (len(list(ks)), self.n)}, scale=1.0)

    def _gradient(self):
        order = self._grad_set  # (gradient-list chunks first: their parts are the leading rows)

        def one(j):
            eng = self.engines[order[j]]
            eng.weight = self._w_grad.get(order[j], 1.0)
            eng.gradient(self._parts[j])

        n_sum = len(dict.fromkeys(self.roles[1]))
        if n_sum == 1:  # (one chunk: straight into the result)
            eng = self.engines[order[0]]
            eng.weight = self._w_grad.get(order[0], 1.0)
            eng.gradient(self.grad_buffer)
            for j in range(1, len(order)):
                one(j)
            return
        self._fork_join(range(len(order)), one)
        self._sum_parts(range(n_sum), self.grad_buffer)

    def _product(self):
        order = self._mvp_set

        def one(j):
            eng = self.engines[order[j]]
            eng.weight = self._w_mvp[order[j]]
            eng.local(self.input_buffer, out=self._parts[j])

        if len(order) == 1:
            eng = self.engines[order[0]]
            eng.weight = self._w_mvp[order[0]]
            eng.local(self.input_buffer, out=self.output_buffer)
            return
        self._fork_join(range(len(order)), one)
        self._sum_parts(range(len(order)), self.output_buffer)

    # ---- validity ------------------------------------------------------------------------------
    def accepts(self, model, loss_func, lists, params, reduction, counts, hessian, group):
        slots, roles = self._plan(lists)
        if (model is not self.model or loss_func is not self.loss_func or reduction != self.reduction
                or bool(hessian) != self.hessian or group is not self.group or roles != self.chunk_roles
                or tuple(float(c) for c in counts) != self.counts):
            return None
        # (what the captured graphs bake in about the layers -- module identities, every BatchNorm's mode / eps /
        # momentum, the model's mode where it matters -- as EngineSession compares it; not model.training itself: an
        # eval() model with one train-mode BatchNorm is a train_bn session)
        if any(eng.layer_signature() != sig for eng, sig in zip(self.engines, self._layers)):
            return None
        if [tuple(x.shape) for x, _ in slots] != self.chunk_shapes:
            return None
        if self._merge_groups(model, slots, roles) != self.groups:
            return None  # (a dropout / BatchNorm layer changed mode: the chunks must no longer / may now be merged)
        if len(params) != len(self.params) or any(a is not b for a, b in zip(params, self.params)):
            return None
        e0 = self.engines[0]
        if e0._flat_params is None or e0._flat_params.data_ptr() != e0.params[0].data_ptr():
            return None
        for x, t in slots:
            if not x.is_cuda or x.dtype != torch.float32 or t.dim() not in (1, 2) or t.shape[0] != x.shape[0]:
                return None
        return slots

    # ---- per step ----------------------------------------------------------------------------
    def begin_step(self, slots, verify=False, reduce=True):
        """New data + current parameters into every engine; returns the accumulated loss (``reduce``: summed over
        the ranks here; else the caller does it -- after the ranks have agreed to use the session at all)."""
        with torch.no_grad():
            for (x, t), eng in zip(self._merged(slots), self.engines):
                eng.set_batch(x.detach(), t)
            self.g_wT.replay()
            (self.g_fwd_still if self._fresh else self.g_fwd_all).replay()
            self._fresh = False
            bad = torch.stack([eng.bad_targets.float().reshape(()) for eng in self.engines]).sum()
            vals = torch.stack([self.loss_buf.float().reshape(()), bad]).tolist()
        if vals[1]:
            raise _NoEngine("a target is outside the classes")
        if verify:
            self._verify(slots)
        self.steps += 1
        self._first_order_fresh = False
        self.base_loss = self.reduce_losses(self.loss_buf.reshape(1)).tolist()[0] if reduce else vals[0]
        return self.base_loss

    def _verify(self, slots):
        """The captured graphs still describe the model: its STOCK forward pass on the first chunk against the
        engine's logits (1e-4); raises ``_NoEngine`` otherwise."""
        x, _ = slots[0]
        with torch.no_grad():
            want = self.model._hf_stock_model_forward(x)
        got = self.engines[0].logits[: want.shape[0]]  # (chunk 0 leads the first engine's batch)
        err = float((got - want).abs().max() / want.abs().max().clamp_min(1e-30))
        if not err < 1e-4:
            raise _NoEngine(f"the captured graphs no longer reproduce the model's own forward pass (logits differ by "
                            f"{err:.1e})")

    def reduce_losses(self, vals):
        """Sum over ranks of (already count-weighted) loss values, as float64."""
        vals = vals.double()
        if self.group is not None:
            torch.distributed.all_reduce(vals, group=self.group)
        return vals

    _first_order_fresh = False  # (Hessian engines: the gradient sweep of this step has left its cotangents)

    def gradient(self):
        self.g_grad.replay()
        self._first_order_fresh = True
        if self.group is not None:
            self.engine.reduce(self.grad_buffer, self.group)
        return self.grad_buffer

    def forward_loss(self, slot):
        self.g_fwd.replay()
        self.losses[slot].copy_(self.loss_buf)
        return self.losses[slot]

    # ---- operator interface of cg() ------------------------------------------------------------
    def raw_graph(self):
        return self.graph.raw_cuda_graph()

    def replay_local(self):
        self.graph.replay()

    def reduce(self, t):
        return self.engine.reduce(t, self.group)

    @property
    def reduce_bytes(self):
        return self.engine.reduce_bytes

    def local(self, v, out=None):
        if v.data_ptr() != self.input_buffer.data_ptr():
            self.input_buffer.copy_(v)
        if not self._first_order_fresh and self.hessian:
            # (as EngineSession.local: products taken right after ``begin_step`` -- ``acc_linearise`` hands the
            # gradient to ``step`` unevaluated -- need this step's first-order cotangents)
            self.g_grad.replay()
            self._first_order_fresh = True
        self.graph.replay()
        if out is not None:
            out.copy_(self.output_buffer)
            return out
        return self.output_buffer

    def __call__(self, v, out=None):
        self.calls += 1
        return self.reduce(self.local(v, out))


class _NoEngine(Exception):
    """The session does not serve this model / loss / data; ``str(exc)`` says why (``HessianFree.path_report()``)."""

    def __init__(self, reason="the fused engine does not cover this model / loss"):
        super().__init__(reason)
        self.reason = reason


def _inject_fault(site, group, buf=None):
    """TEST HOOK ``HF_TEST_DP_FAULT=<kind>[:<where>]`` (tests/test_distributed_gpu.py: the data-parallel fallback
    ladder of ``bench.py --gpus N``): makes the LAST rank of the group misbehave in a data-parallel session product.
    ``kind``: ``hang`` (sleeps for good before the product), ``raise`` (RuntimeError), ``mismatch`` (its copy of the
    summed product is off by one in entry 0 -- the ranks no longer hold identical sums).  ``where``: ``two_phase``
    (default: only the two-phase form) or ``not_plain`` (either form, unless the run is the plainest configuration:
    ``HF_CHUNKED_ALLREDUCE=0`` and ``HF_DIRECT_RCCL=0``)."""
    spec = os.environ.get("HF_TEST_DP_FAULT", "")
    if not spec or group is None:
        return
    kind, _, where = spec.partition(":")
    where = where or "two_phase"
    if where == "two_phase":
        active = site == "two_phase"
    else:
        active = not (os.environ.get("HF_CHUNKED_ALLREDUCE", "auto") == "0" and os.environ.get("HF_DIRECT_RCCL", "1") == "0")
    dist = torch.distributed
    if not active or dist.get_rank(group) != dist.get_world_size(group) - 1:
        return
    if buf is None:
        if kind == "hang":
            import time

            time.sleep(3600.0)
        elif kind == "raise":
            raise RuntimeError(f"HF_TEST_DP_FAULT: injected failure in the {site} product")
    elif kind == "mismatch":
        buf[:1] += 1.0
