"""Curvature-vector products ``v -> B v`` on flat parameter vectors (the ``mvp`` of
the reference's ``optimizer.py:237-247, :450-462``) and their data-parallel sum.

The reference delegates to BackPACK, which re-derives everything on every call:
``R_op`` builds a double-backward graph through a dummy cotangent, runs it, and
``parameters_to_vector`` concatenates the result (optimizer.py:457-462).  Here

* the linear map ``u -> J^T u`` (``J = d outputs / d params``) is recorded ONCE per
  step as an autograd graph; ``J v`` is then one backward sweep through that
  recorded graph per matvec (MIOpen/rocBLAS kernels on PyTorch-ROCm -- this is
  not a dense contraction, so no custom MFMA kernel);
* the loss Hessian ``H_L`` is likewise recorded once (``d loss / d outputs`` with
  graph) and applied with one tiny backward sweep;
* ``J^T (H_L J v)`` is one ordinary reverse pass through the step's forward graph;
* the per-parameter results are gathered into ONE contiguous HBM vector by the
  multi-tensor ``hf_pack`` kernel (8 N bytes instead of ``torch.cat``), with the
  data-parallel weight ``N_k / sum N`` folded into the same pass;
* across ranks the partial products are summed with ONE all-reduce per matvec
  (the ``result += N * mb_result`` of optimizer.py:677-684 turned sideways).

On CPU tensors (host-logic tests, gloo tests) the gather is ``torch.cat``; on a
GPU the HIP kernel is mandatory (``_lib`` raises if the library is missing).
"""

import torch

from . import _lib
from .modelprep import consumed_at_once, first_order_only, tangent_owner
from .utils import vector_to_parameter_list


def flatten_into(tensors, like_params, out=None, scale=1.0):
    """``parameters_to_vector`` replacement: gather ``tensors`` (``None`` -> zeros
    shaped like the matching parameter) into one flat vector, times ``scale``."""
    ref = like_params[0]
    n = sum(p.numel() for p in like_params)
    if out is None:
        out = torch.empty(n, dtype=ref.dtype, device=ref.device)
    dense = [
        torch.zeros_like(p) if t is None else t.detach() for t, p in zip(tensors, like_params)
    ]
    if out.is_cuda:
        _lib.pack(out, dense, scale=scale, mode=0)
    else:
        torch.cat([t.reshape(-1) for t in dense], out=out)
        if scale != 1.0:
            out.mul_(scale)
    return out


def _all_reduce_sum(t, group):
    from .distributed import all_reduce_sum

    return all_reduce_sum(t, group)


class _Operator:
    """Common part: flat-vector in, flat-vector out, optional rank weight and
    process group.  ``group=None`` means single process (reference behaviour)."""

    def __init__(self, params, weight=1.0, group=None):
        self.params = list(params)
        self.weight = float(weight)
        self.group = group
        self.n = sum(p.numel() for p in self.params)
        self.calls = 0

    mode = "eager autograd"

    def _finish(self, per_param, out):
        return flatten_into(per_param, self.params, out=out, scale=self.weight)

    def local(self, v, out=None):
        """This rank's weighted partial product (no communication)."""
        raise NotImplementedError

    def reduce(self, t, group=None):
        """In-place sum of a local product over the ranks (``group`` defaults to the operator's
        own); operators that know more about the vector may move fewer bytes (the engine)."""
        return _all_reduce_sum(t, self.group if group is None else group)

    def __call__(self, v, out=None):
        self.calls += 1
        return self.reduce(self.local(v, out))


class GGNOperator(_Operator):
    """``v -> J^T H_L J v`` for the mini-batch behind ``(loss, outputs)``
    (math contract of BackPACK's ``ggn_vector_product_from_plist``,
    optimizer.py:457-462)."""

    def __init__(self, loss, outputs, params, weight=1.0, group=None):
        super().__init__(params, weight, group)
        self.outputs = outputs
        # u -> J^T u, recorded once (u is a dummy cotangent)
        self._u = torch.zeros_like(outputs, requires_grad=True)
        with first_order_only():  # only d/du of this map is ever taken
            JTu = torch.autograd.grad(
                outputs, self.params, grad_outputs=self._u, create_graph=True,
                retain_graph=True, allow_unused=True,
            )
        self._used = [i for i, g in enumerate(JTu) if g is not None]
        self._JTu = [JTu[i] for i in self._used]
        # d loss / d outputs with graph -> H_L by one more sweep
        (self._dl,) = torch.autograd.grad(loss, outputs, create_graph=True, retain_graph=True)
        self._ce = self._closed_form_loss_hessian(loss, outputs)

    def _closed_form_loss_hessian(self, loss, outputs):
        """``(p, scale)`` if the loss is a softmax cross-entropy of ``outputs`` whose
        Hessian ``scale * (diag(p) - p p^T)`` (per row) reproduces the autograd sweep
        through ``d loss / d outputs`` on a random vector, else ``None``.  The structural
        test only nominates; the numerical check decides (class weights, ignored targets,
        label smoothing etc. fail it and keep the generic sweep).  GPU tensors only."""
        fn = loss.grad_fn
        try:
            if not outputs.is_cuda or outputs.dim() != 2 or fn is None or fn.name() != "NllLossBackward0":
                return None
            lsm = fn.next_functions[0][0]
            if lsm is None or lsm.name() != "LogSoftmaxBackward0" or lsm._saved_dim not in (1, -1):
                return None
            if lsm.next_functions[0][0] is not outputs.grad_fn or not self._dl.requires_grad:
                return None
            scale = {1: 1.0 / outputs.shape[0], 2: 1.0}.get(fn._saved_reduction)
        except AttributeError:
            return None
        if scale is None:
            return None
        p = torch.softmax(outputs.detach(), dim=1)
        gen = torch.Generator(device=outputs.device).manual_seed(99)
        probe = torch.randn(outputs.shape, dtype=outputs.dtype, device=outputs.device, generator=gen)
        (want,) = torch.autograd.grad(self._dl, outputs, grad_outputs=probe, retain_graph=True)
        got = _lib.softmax_ce_hvp(p, probe, scale)
        if not float((got - want).abs().max()) <= 1e-5 * float(want.abs().max()):
            return None
        return p, scale

    def _loss_hessian(self, Jv):
        if self._ce is not None:
            return _lib.softmax_ce_hvp(self._ce[0], Jv, self._ce[1])
        if not self._dl.requires_grad:  # loss linear in the outputs: H_L = 0
            return torch.zeros_like(Jv)
        (HJv,) = torch.autograd.grad(
            self._dl, self.outputs, grad_outputs=Jv, retain_graph=True, allow_unused=True
        )
        return torch.zeros_like(Jv) if HJv is None else HJv

    def local(self, v, out=None):
        vs = vector_to_parameter_list(v, self.params)
        with tangent_owner(self, v):  # conv layers' v_W operands: one scatter launch
            (Jv,) = torch.autograd.grad(
                self._JTu, self._u, grad_outputs=[vs[i] for i in self._used], retain_graph=True
            )
        HJv = self._loss_hessian(Jv)
        with consumed_at_once():  # gathered by hf_pack right below
            JTHJv = torch.autograd.grad(
                self.outputs, self.params, grad_outputs=HJv, retain_graph=True, allow_unused=True
            )
            return self._finish(JTHJv, out)

    # -- the same product in two phases, for overlapping the all-reduce with compute --
    def split_point(self, tail_fraction=0.75):
        """Index ``c`` such that ``params[c:]`` (the LAST layers, whose gradients the
        adjoint sweep produces FIRST) hold at most ``tail_fraction`` of the entries;
        returns ``(c, flat offset of params[c])``."""
        total, acc, c = self.n, 0, len(self.params)
        while c > 1 and acc + self.params[c - 1].numel() <= tail_fraction * total:
            c -= 1
            acc += self.params[c].numel()
        return c, total - acc

    def _H_J(self, v):
        vs = vector_to_parameter_list(v, self.params)
        with tangent_owner(self, v):
            (Jv,) = torch.autograd.grad(
                self._JTu, self._u, grad_outputs=[vs[i] for i in self._used], retain_graph=True
            )
        return self._loss_hessian(Jv)

    def phase_tail(self, v, out, cut, offset):
        """``out[offset:] = weight * (J^T H_L J v)[offset:]`` -- tangent sweep, loss
        Hessian, and the adjoint sweep only as far back as ``params[cut]``."""
        self._hjv = self._H_J(v)
        tail = self.params[cut:]
        with consumed_at_once():
            g = torch.autograd.grad(self.outputs, tail, grad_outputs=self._hjv, retain_graph=True,
                                    allow_unused=True)
            flatten_into(g, tail, out=out[offset:], scale=self.weight)

    def phase_head(self, out, cut, offset):
        """``out[:offset]``: the rest of the adjoint sweep (re-walks the tail's data
        gradients, which is cheap next to the all-reduce it hides)."""
        head = self.params[:cut]
        with consumed_at_once():
            g = torch.autograd.grad(self.outputs, head, grad_outputs=self._hjv, retain_graph=True,
                                    allow_unused=True)
            flatten_into(g, head, out=out[:offset], scale=self.weight)


def ggn_operator(loss, outputs, params, weight=1.0, group=None, why=None):
    """The GGN operator for ``(loss, outputs)``: the fused curvature engine (engine.py) when
    ``outputs`` comes from a prepared model of a family it knows (conv - eval-BatchNorm - ReLU
    units with residual connections, NHWC fp32), else the autograd operator above."""
    if isinstance(outputs, torch.Tensor) and getattr(outputs, "_hf_model", None) is not None:
        from .engine import FusedGGNEngine

        eng = FusedGGNEngine.try_build(loss, outputs, list(params), weight=weight, group=group, why=why)
        if eng is not None:
            return eng
    elif why is not None:
        why.append("the model is not a prepared one (modelprep.prepare_model(model, channels_last=True) installs the "
                   "layers the fused engine reads)")
    return GGNOperator(loss, outputs, params, weight=weight, group=group)


def hessian_operator(loss, outputs, params, grad_with_graph=None, weight=1.0, group=None, why=None):
    """The Hessian operator for ``loss``: the fused curvature engine in Hessian mode (engine.py:
    forward-over-reverse on the package's own kernels) when ``outputs`` comes from a prepared model of
    a family it covers (plain conv-ReLU stacks with a softmax cross-entropy, optionally plus a tagged
    L2 term), else the autograd operator below."""
    if isinstance(outputs, torch.Tensor) and getattr(outputs, "_hf_model", None) is not None:
        from .engine import FusedGGNEngine

        eng = FusedGGNEngine.try_build(loss, outputs, list(params), weight=weight, group=group, hessian=True,
                                       why=why)
        if eng is not None:
            return eng
    elif why is not None:
        why.append("the model is not a prepared one (modelprep.prepare_model(model, channels_last=True) installs the "
                   "layers the fused engine reads)")
    return HessianOperator(loss, params, grad_with_graph=grad_with_graph, weight=weight, group=group)


class HessianOperator(_Operator):
    """``v -> (d^2 loss / d params^2) v`` (BackPACK's ``hessian_vector_product``,
    optimizer.py:450-455).  ``grad_with_graph`` are the per-parameter gradients
    computed with ``create_graph=True`` (the step needs them anyway,
    optimizer.py:231-233); H is symmetric, so one reverse sweep through that
    graph with cotangent ``v`` is ``H v``."""

    def __init__(self, loss, params, grad_with_graph=None, weight=1.0, group=None):
        super().__init__(params, weight, group)
        if grad_with_graph is None:
            grad_with_graph = torch.autograd.grad(
                loss, self.params, create_graph=True, retain_graph=True
            )
        self._used = [i for i, g in enumerate(grad_with_graph) if g.requires_grad]
        self._g = [grad_with_graph[i] for i in self._used]

    def local(self, v, out=None):
        vs = vector_to_parameter_list(v, self.params)
        if self._g:
            Hv = torch.autograd.grad(
                self._g, self.params, grad_outputs=[vs[i] for i in self._used],
                retain_graph=True, allow_unused=True,
            )
        else:  # loss linear in the parameters
            Hv = [None] * len(self.params)
        return self._finish(Hv, out)


class GraphedOperator:
    """The local part of a curvature operator (autograd sweeps + ``hf_pack``)
    captured ONCE into a hipGraph and replayed per matvec.

    On a ResNet-18 one GGN matvec is ~700 small kernels; issued eagerly the host
    needs longer to launch them than the MI355X needs to run them.  A graph
    replay costs one launch.  The graph reads its input from ``input_buffer`` and
    writes ``output_buffer``; :func:`~pytorchhessianfree_amd.cg.cg` adopts
    ``input_buffer`` as its search-direction vector ``p`` (the update kernel writes
    the next direction straight into the graph's input: no copy), any other input
    vector is copied in first.  The result tensor is overwritten by the next call.
    The data-parallel all-reduce stays outside the graph.

    ``builder()`` must run the forward pass and return the eager operator; it is
    executed on the capture stream because autograd issues backward kernels on
    the stream their forward ran on.

    Precondition (PyTorch-ROCm 2.10 / HIP 7.0 crash the process otherwise, measured
    in scripts/experiments/graph_repro*.py): while capturing, no autograd state that ties the
    parameters to ANOTHER stream may be alive -- i.e. no ``.grad`` on them (stashed
    and restored here when ``params`` is given) and no live autograd graph built
    on another stream that reaches them (their ``AccumulateGrad`` nodes remember
    that stream and the engine then touches it during capture).
    """

    @property
    def mode(self):
        inner = getattr(getattr(self, "op", None), "mode", "")
        if "engine" in inner:
            return "hipGraph replay of the " + inner
        return "hipGraph replay of autograd sweeps + hf_pack"

    _captured_before = False  # later captures in a process need a single warm-up run
    _streams = {}             # one capture stream per device, shared by all instances

    def __init__(self, builder, warmup=None, params=None):
        if warmup is None:
            warmup = 1 if GraphedOperator._captured_before else 3
        if not torch.cuda.is_available():
            raise RuntimeError("GraphedOperator needs a GPU")
        import gc

        first = not GraphedOperator._captured_before
        GraphedOperator._captured_before = True
        stash = [(p, p.grad) for p in (params or []) if p.grad is not None]
        for p, _ in stash:
            p.grad = None
        if first or stash:
            # let dead graphs (and their AccumulateGrad nodes) go.  Later captures
            # skip this 40 ms sweep: every instance records on the SAME stream, so
            # a lingering earlier graph of ours ties the parameters to that stream only
            gc.collect()
        try:
            self._capture(builder, warmup)
        finally:
            for p, g in stash:
                p.grad = g

    def _capture(self, builder, warmup):
        cur = torch.cuda.current_stream()
        dev = torch.cuda.current_device()
        if dev not in GraphedOperator._streams:
            GraphedOperator._streams[dev] = torch.cuda.Stream()
        self.stream = GraphedOperator._streams[dev]
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            self.op = builder()
            ref = self.op.params[0]
            self.n, self.group = self.op.n, self.op.group
            self.params = self.op.params
            self.input_buffer = torch.zeros(self.n, dtype=ref.dtype, device=ref.device)
            self.output_buffer = torch.empty(self.n, dtype=ref.dtype, device=ref.device)
            for _ in range(warmup):
                self.op.local(self.input_buffer, out=self.output_buffer)
        self.stream.synchronize()
        # keep_graph: the raw hipGraph_t stays available, so that cg() can clone it into
        # its one-launch-per-iteration graph (product -> K1 -> K2 -> K3, hf_pcg_graph_*)
        self.graph = torch.cuda.CUDAGraph(keep_graph=True)
        with torch.cuda.graph(self.graph, stream=self.stream):
            self.op.local(self.input_buffer, out=self.output_buffer)
        self.graph.instantiate()
        cur.wait_stream(self.stream)
        torch.cuda.synchronize()
        self.calls = 0
        self._verify_replay()

    def raw_graph(self):
        """The captured product as a raw ``hipGraph_t`` (an int), owned by ``self.graph``."""
        return self.graph.raw_cuda_graph()

    def replay_local(self):
        """Replay the captured local product: reads ``input_buffer``, writes ``output_buffer``."""
        self.graph.replay()

    def reduce(self, t):
        inner = getattr(self.op, "reduce", None)  # (the captured operator's own rule, if it has one)
        if inner is not None:
            return inner(t, self.group)
        return _all_reduce_sum(t, self.group)

    _verified = set()  # signatures whose first capture in this process was checked

    def _verify_replay(self):
        """Replay twice on a random vector and compare with the eager product.

        Not paranoia: on this stack (PyTorch-ROCm 2.10, HIP 7.0, MIOpen 3.5) library
        routines that zero a scratch buffer and then accumulate into it (MIOpen's CK
        split-K weight gradient, its backward-bias, PyTorch's multi-block reductions of
        NHWC tensors) were measured to be right when issued eagerly and wrong when
        replayed from a hipGraph, the second replay differently from the first.  The
        package routes around the instances it met; a model that brings a new one
        must fail here, loudly, instead of producing wrong Newton steps.  By default
        the first capture of every signature in the process -- vector size, parameter
        shapes, shape of the network output (batch size!), operator type, device: a
        different batch or input shape selects different library kernels -- is checked
        (~2 products); ``HF_GRAPH_VERIFY=always|never`` overrides."""
        import os

        policy = os.environ.get("HF_GRAPH_VERIFY", "first")
        outputs = getattr(self.op, "outputs", None)
        key = (self.n, tuple(tuple(p.shape) for p in self.params), str(self.input_buffer.device),
               self.input_buffer.dtype, type(self.op).__name__,
               None if outputs is None else tuple(outputs.shape),
               getattr(self.op, "input_signature", None))
        if policy == "never" or (policy != "always" and key in GraphedOperator._verified):
            return
        gen = torch.Generator(device=self.input_buffer.device).manual_seed(1234)
        with torch.cuda.stream(self.stream):
            self.input_buffer.copy_(torch.randn(self.n, dtype=self.input_buffer.dtype,
                                                device=self.input_buffer.device, generator=gen))
            want = torch.empty_like(self.output_buffer)
            self.op.local(self.input_buffer, out=want)
            errs = []
            for _ in range(2):
                self._replay()
                errs.append(float((self.output_buffer - want).abs().max() / want.abs().max().clamp_min(1e-30)))
            self.input_buffer.zero_()
        self.stream.synchronize()
        torch.cuda.current_stream().wait_stream(self.stream)
        if not all(e < 1e-3 for e in errs):
            raise RuntimeError(
                "hipGraph replay of the curvature product does not reproduce the eager product "
                f"(relative max-norm error of two replays: {errs[0]:.2e}, {errs[1]:.2e}). A library "
                "routine inside the product is not capture-safe on this stack; run with "
                "graph_matvec=False (eager) or prepare the model with modelprep.prepare_model.")
        GraphedOperator._verified.add(key)

    def _replay(self):
        self.graph.replay()

    def local(self, v, out=None):
        if v.data_ptr() != self.input_buffer.data_ptr():
            self.input_buffer.copy_(v)
        self.graph.replay()
        if out is not None:
            out.copy_(self.output_buffer)
            return out
        return self.output_buffer

    def __call__(self, v, out=None):
        self.calls += 1
        return self.reduce(self.local(v, out))


class OverlappedGraphedOperator(GraphedOperator):
    """Data-parallel variant: the local product is captured as TWO hipGraphs so that
    the all-reduce of the last layers' part of the vector (the adjoint sweep
    produces it first; on conv nets it is most of the bytes) runs while the rest
    of the sweep is still computing:

        replay G1 (tangent sweep, H_L, adjoint of the tail, pack tail)
        all-reduce(out[offset:])   -- asynchronous, on the collective's stream
        replay G2 (adjoint of the head, pack head)          <- overlaps
        all-reduce(out[:offset])   -- asynchronous
        wait for both

    The collectives stay outside the graphs.  Results are those of the unsplit
    product up to fp32 summation order.

    EXPERIMENTAL, off by default.  Measured on one MI355X (ResNet-18, batch 32):
    the split costs +0.19 ms per product (the head's sweep re-walks the tail's
    data gradients; second graph launch), and two asynchronous collectives cost
    a further ~0.4 ms of stream hand-offs even on a 1-rank RCCL group -- more than
    the ~0.3 ms of all-reduce it can hide on 44.7 MB.  It pays only where the
    all-reduce is slow (many ranks over few links, much larger vectors)."""

    mode = "2 hipGraphs per product, all-reduce of the tail overlapped with the head's adjoint sweep"
    raw_graph = None  # two graphs: cg() keeps its separate K1-K3 launches

    def __init__(self, builder, params=None, tail_fraction=0.75):
        self._tail_fraction = tail_fraction
        super().__init__(builder, params=params)

    def _capture(self, builder, warmup):
        cur = torch.cuda.current_stream()
        dev = torch.cuda.current_device()
        if dev not in GraphedOperator._streams:
            GraphedOperator._streams[dev] = torch.cuda.Stream()
        self.stream = GraphedOperator._streams[dev]
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            self.op = builder()
            if not isinstance(self.op, GGNOperator):
                raise TypeError("OverlappedGraphedOperator needs a GGNOperator")
            ref = self.op.params[0]
            self.n, self.group, self.params = self.op.n, self.op.group, self.op.params
            self.cut, self.offset = self.op.split_point(self._tail_fraction)
            self.input_buffer = torch.zeros(self.n, dtype=ref.dtype, device=ref.device)
            self.output_buffer = torch.empty(self.n, dtype=ref.dtype, device=ref.device)
            for _ in range(warmup):
                self.op.phase_tail(self.input_buffer, self.output_buffer, self.cut, self.offset)
                self.op.phase_head(self.output_buffer, self.cut, self.offset)
        self.stream.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=self.stream):
            self.op.phase_tail(self.input_buffer, self.output_buffer, self.cut, self.offset)
        self.graph_head = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph_head, stream=self.stream, pool=self.graph.pool()):
            self.op.phase_head(self.output_buffer, self.cut, self.offset)
        cur.wait_stream(self.stream)
        torch.cuda.synchronize()
        self.calls = 0
        self._verify_replay()

    def _replay(self):
        self.graph.replay()
        self.graph_head.replay()

    def local(self, v, out=None):
        if v.data_ptr() != self.input_buffer.data_ptr():
            self.input_buffer.copy_(v)
        self.graph.replay()
        self.graph_head.replay()
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
        from . import distributed as hfdist

        dist, buf = torch.distributed, self.output_buffer
        tail, head = buf[self.offset:], buf[: self.offset]
        side_comm = hfdist.side_comm(tail, self.group)
        if side_comm is not None:
            # direct RCCL: the tail's all-reduce runs on a side stream with its own
            # communicator while the head's adjoint sweep replays; the head's all-reduce
            # follows on the compute stream -- one event hand-off in, one out
            cur = torch.cuda.current_stream()
            if getattr(self, "_side", None) is None:
                self._side = torch.cuda.Stream()
            self.graph.replay()
            self._side.wait_stream(cur)
            with torch.cuda.stream(self._side):
                side_comm.all_reduce_sum(tail)
            self.graph_head.replay()
            hfdist.all_reduce_sum(head, self.group)
            cur.wait_stream(self._side)
        else:
            self.graph.replay()
            w_tail = dist.all_reduce(tail, group=self.group, async_op=True)
            self.graph_head.replay()
            w_head = dist.all_reduce(head, group=self.group, async_op=True)
            w_tail.wait()
            w_head.wait()
        if out is not None:
            out.copy_(buf)
            return out
        return buf


def maybe_graphed(builder, enable=True, params=None):
    """``GraphedOperator(builder)`` if capture works on this stack, else the eager
    operator (still HIP: only the launch mechanism differs)."""
    if enable and torch.cuda.is_available():
        try:
            return GraphedOperator(builder, params=params)
        except Exception as exc:  # capture not supported for some op on this stack
            import warnings

            warnings.warn(f"hipGraph capture of the matvec failed ({exc!r}); running it eagerly")
            torch.cuda.synchronize()
    return builder()
