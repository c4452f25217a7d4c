"""CPU: host logic added around the persistent engine session -- batched / cached trial-loss
evaluation (``optimizer._SessionTrials``), the prefetch hooks of CG-backtracking and the line search
(reference ``hessianfree/cg_backtracking.py:53-112``, ``hessianfree/linesearch.py:8-103``: same results, same
early exits, fewer device->host reads), and the structural loss detection (``engine.ce_loss_spec``)."""

import torch

from pytorchhessianfree_amd.cg_backtracking import cg_efficient_backtracking
from pytorchhessianfree_amd.engine import ce_loss_spec
from pytorchhessianfree_amd.linesearch import simple_linesearch
from pytorchhessianfree_amd.optimizer import _SessionTrials
from pytorchhessianfree_amd.utils import ParameterArena


class _FakeSession:
    """Stands in for ``session.EngineSession``: ``forward_loss(slot)`` evaluates a fixed quadratic of the
    arena's current parameters into ``losses[slot]`` (no host synchronisation in the real one)."""

    def __init__(self, arena, target):
        self.arena, self.target = arena, target
        self.losses = torch.zeros(8)
        self.forwards = 0
        self.base_loss = float(((arena.theta - target) ** 2).sum())

    def forward_loss(self, slot):
        self.forwards += 1
        self.losses[slot] = ((self.arena.theta - self.target) ** 2).sum()
        return self.losses[slot]


class _FakeOpt:
    process_group, shard_weight = None, 1.0


def _setup(n=12, seed=0):
    g = torch.Generator().manual_seed(seed)
    params = [torch.nn.Parameter(torch.randn(n, generator=g))]
    arena = ParameterArena(params)
    base = arena.snapshot()
    target = torch.randn(n, generator=g)
    sess = _FakeSession(arena, target)
    trials = _SessionTrials(_FakeOpt(), sess, arena, base)
    flushes = [0]
    real_flush = trials.flush

    def counting_flush():
        if trials.pending:
            flushes[0] += 1
        real_flush()

    trials.flush = counting_flush
    return arena, base, target, sess, trials, flushes


def _direct(base, target, step, alpha):
    return float(((base + alpha * step - target) ** 2).sum())


def test_trial_values_are_exact_cached_and_batched():
    arena, base, target, sess, trials, flushes = _setup()
    steps = [torch.randn(12, generator=torch.Generator().manual_seed(i)) for i in range(4)]
    trials.prefetch([(steps[0], 1.0), (steps[1], 1.0)])  # LM damping's pair: one read-back
    assert flushes[0] == 0 and sess.forwards == 2
    a, b = trials.value(steps[0], 1.0), trials.value(steps[1], 1.0)
    assert flushes[0] == 1
    assert abs(a - _direct(base, target, steps[0], 1.0)) < 1e-5 * abs(a)
    assert abs(b - _direct(base, target, steps[1], 1.0)) < 1e-5 * abs(b)
    # asked again (the last CG iterate in back-tracking, the chosen step at alpha = 1): no forward, no read
    assert trials.value(steps[1], 1.0) == b and sess.forwards == 2 and flushes[0] == 1
    # alpha = 0 is the base point of the step: the loss the step's own forward pass produced
    assert trials.value(steps[2], 0.0) == sess.base_loss and sess.forwards == 2
    # a scaled step is another point
    c = trials.value(steps[1], 0.5)
    assert sess.forwards == 3 and abs(c - _direct(base, target, steps[1], 0.5)) < 1e-5 * abs(c)
    # more pending points than loss slots: flushed in between, all values right
    many = [(steps[3], 0.1 * k) for k in range(1, 12)]
    trials.prefetch(many)
    for s, al in many:
        v = trials.value(s, al)
        assert abs(v - _direct(base, target, s, al)) < 1e-5 * max(1.0, abs(v))


def test_train_mode_session_evaluates_no_speculative_points():
    """Train-mode BatchNorm: every evaluated trial point moves the running statistics, as every ``forward()`` of
    the reference does (optimizer.py:288-294) -- so the speculative second candidate of a back-tracking / Armijo
    pair (``needed=1``) is NOT evaluated there, while an eval-mode session evaluates both with one read-back."""
    for train, forwards in ((False, 2), (True, 1)):
        arena, base, target, sess, trials, flushes = _setup()

        class _Eng:
            train_bn = train

        sess.engine = _Eng()
        steps = [torch.randn(12, generator=torch.Generator().manual_seed(i)) for i in range(2)]
        trials.prefetch([(steps[0], 1.0), (steps[1], 1.0)], needed=1)
        assert sess.forwards == forwards
        v = trials.value(steps[0], 1.0)
        assert abs(v - _direct(base, target, steps[0], 1.0)) < 1e-5 * abs(v)
        # the LM pair (both values certainly consumed) is evaluated in full either way
        trials.prefetch([(steps[0], 0.5), (steps[1], 0.5)])
        assert sess.forwards == forwards + 2


def test_backtracking_with_prefetch_equals_reference_walk():
    """The reference's toy sequence (tests/test_cg_backtracking.py:8-44): walk from the last stored step
    backwards, stop at the first non-improvement.  With the prefetch hook the same index / value come out, at
    most one candidate beyond the stopping point is evaluated, and a pair of values costs one read-back."""
    values = [2.0, 1.0, None, 2.7, 2.4, None, None, 7.3]
    steps = [None if v is None else torch.full((3,), float(i)) for i, v in enumerate(values)]
    evaluated, reads, pending = [], [0], []

    def f(step):
        idx = int(step[0])
        if idx in pending:
            reads[0] += 1
            pending.clear()
        elif idx not in evaluated:
            evaluated.append(idx)
            reads[0] += 1
        return values[idx]

    def prefetch(points, needed=None):
        for step, alpha in points:
            idx = int(step[0])
            if idx not in evaluated:
                evaluated.append(idx)
                pending.append(idx)

    plain = cg_efficient_backtracking(f, steps)
    n_plain = len(evaluated)
    evaluated.clear(); pending.clear(); reads[0] = 0
    f.prefetch = prefetch
    hooked = cg_efficient_backtracking(f, steps)
    assert hooked == plain == (4, 2.4)
    assert len(evaluated) <= n_plain + 1  # the early exit is the reference's, one speculative evaluation at most
    assert reads[0] <= (len(evaluated) + 1) // 2 + 1


def test_linesearch_with_prefetch_equals_plain():
    arena, base, target, sess, trials, flushes = _setup(seed=3)
    step = 6.0 * (target - base)  # overshoots: the Armijo rule has to shrink alpha a few times
    grad0 = 2.0 * (base - target)

    def f_plain(s):
        return _direct(base, target, s, 1.0)

    want = simple_linesearch(f_plain, grad0, step, init_alpha=1.0)

    def tfunc(s):
        return trials.value(s, 1.0)

    tfunc.scaled = trials.value
    tfunc.prefetch = trials.prefetch
    got = simple_linesearch(tfunc, grad0, step, init_alpha=1.0)
    assert got[0] == want[0] and abs(got[1] - want[1]) <= 1e-5 * abs(want[1])
    assert got[0] < 1.0  # (the search really iterated)
    assert flushes[0] < sess.forwards  # fewer read-backs than evaluations


def test_ce_loss_spec_reads_the_loss_structure():
    torch.manual_seed(0)
    w = torch.randn(5, 4, requires_grad=True)
    x, t = torch.randn(6, 4), torch.randint(0, 5, (6,))
    out = x @ w.t()
    ce = torch.nn.functional.cross_entropy
    spec = ce_loss_spec(ce(out, t), out)
    assert spec is not None and spec["reduction"] == "mean" and torch.equal(spec["targets"], t)
    assert ce_loss_spec(ce(out, t, reduction="sum"), out)["reduction"] == "sum"
    assert ce_loss_spec(ce(out, t, label_smoothing=0.1), out) is None
    assert ce_loss_spec(ce(out, t, weight=torch.rand(5)), out) is None
    assert ce_loss_spec(ce(2.0 * out, t), out) is None  # not the cross-entropy OF `outputs`
    assert ce_loss_spec(torch.nn.functional.mse_loss(out, torch.randn(6, 5)), out) is None
    t_ignored = t.clone()
    t_ignored[0] = -100
    assert ce_loss_spec(ce(out, t_ignored), out) is None
    # cross-entropy + a tagged quadratic regulariser (testproblems.l2_regularized)
    total = ce(out, t) + 0.5 * 1e-3 * (w * w).sum()
    assert ce_loss_spec(total, out) is None  # untagged: unknown structure
    total._hf_quadratic = ((1e-3, (w,)),)
    spec = ce_loss_spec(total, out)
    assert spec is not None and spec["quadratic"][0][0] == 1e-3 and spec["quadratic"][0][1][0] is w
    # logits handed out as a leaf (what a persistent session's forward pass returns)
    leaf = out.detach().clone().requires_grad_(True)
    assert ce_loss_spec(ce(leaf, t), leaf) is not None


def test_accumulated_session_plans_engines_by_chunk_identity():
    """``acc_step`` on the engine keys one engine per DISTINCT chunk (identity of its tensors): lists that share
    chunks -- the default, one list for loss, gradient and curvature (optimizer.py:519-606) -- share engines; distinct
    lists get their own; a chunk listed twice in one list is one engine (its weight counts twice)."""
    from pytorchhessianfree_amd.session import AccumulatedSession

    a, b, c = [(torch.zeros(4, 3), torch.zeros(4)) for _ in range(3)]
    one = [a, b]
    slots, roles = AccumulatedSession._plan((one, one, one))
    assert len(slots) == 2 and roles == [(0, 1), (0, 1), (0, 1)]
    slots, roles = AccumulatedSession._plan(([a, b], [b, c], [c]))
    assert [s[0] is t[0] and s[1] is t[1] for s, t in zip(slots, (a, b, c))] == [True, True, True]
    assert roles == [(0, 1), (1, 2), (2,)]
    slots, roles = AccumulatedSession._plan(([a, a], [a], [a]))
    assert len(slots) == 1 and roles == [(0, 0), (0,), (0,)]


def test_accumulated_session_merges_chunks_of_equal_per_sample_weight():
    """Chunks that appear in the same lists the same number of times carry one per-sample weight (optimizer.py:
    677-684: ``N_k / sum N`` per chunk = ``1 / sum N`` per sample): for a model that does not couple the samples of
    a batch they run as ONE batch (the reference's own statement, tests/test_optimizer_acc.py:116-175).  Not merged:
    a train-mode BatchNorm / active dropout anywhere in the model, chunks of different shape, chunks with different
    list membership."""
    from pytorchhessianfree_amd.session import AccumulatedSession

    net = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3), torch.nn.BatchNorm2d(4), torch.nn.Dropout(0.5)).eval()
    mk = lambda n, hw=8: (torch.zeros(n, 3, hw, hw), torch.zeros(n, dtype=torch.int64))  # noqa: E731
    a, b, c, d = mk(20), mk(12), mk(8), mk(8, hw=6)
    slots, roles = AccumulatedSession._plan(([a, b], [a, b], [a, b]))
    assert AccumulatedSession._merge_groups(net, slots, roles) == [[0, 1]]
    # distinct lists: one group per list
    slots, roles = AccumulatedSession._plan(([a, b], [b, c], [c]))
    assert AccumulatedSession._merge_groups(net, slots, roles) == [[0], [1], [2]]
    slots, roles = AccumulatedSession._plan(([a, b], [c, mk(4)], [a, b]))
    assert AccumulatedSession._merge_groups(net, slots, roles) == [[0, 1], [2, 3]]
    # a chunk listed twice weighs twice per sample: its own group
    slots, roles = AccumulatedSession._plan(([a, a, b], [a, b], [a, b]))
    assert AccumulatedSession._merge_groups(net, slots, roles) == [[0], [1]]
    # another image size
    slots, roles = AccumulatedSession._plan(([c, d],) * 3)
    assert AccumulatedSession._merge_groups(net, slots, roles) == [[0], [1]]
    # batch-coupled layers: per-chunk statistics / masks are part of the reference's result
    slots, roles = AccumulatedSession._plan(([a, b],) * 3)
    net[1].train()
    assert AccumulatedSession._merge_groups(net, slots, roles) == [[0], [1]]
    net[1].eval()
    net[2].train()
    assert AccumulatedSession._merge_groups(net, slots, roles) == [[0], [1]]


def test_mse_loss_spec_reads_the_loss_structure():
    """``engine.mse_loss_spec`` / ``loss_spec_of``: the loss of the reference's own examples and tests (``nn.MSELoss``,
    examples/run_mwe.py:19, tests/test_utils.py:47) is recognised structurally -- mean / sum, constant targets of the
    outputs' shape, OF the given outputs -- and nothing else is."""
    from pytorchhessianfree_amd.engine import loss_spec_of, mse_loss_spec

    lin = torch.nn.Linear(4, 5)
    out = lin(torch.randn(6, 4))
    t = torch.randn(6, 5)
    spec = mse_loss_spec(torch.nn.MSELoss()(out, t), out)
    assert spec is not None and spec["kind"] == "mse" and spec["reduction"] == "mean" and torch.equal(spec["targets"], t)
    assert mse_loss_spec(torch.nn.functional.mse_loss(out, t, reduction="sum"), out)["reduction"] == "sum"
    assert loss_spec_of(torch.nn.MSELoss()(out, t), out)["kind"] == "mse"
    assert loss_spec_of(torch.nn.functional.cross_entropy(out, torch.randint(0, 5, (6,))), out)["kind"] == "ce"
    assert mse_loss_spec(torch.nn.functional.mse_loss(2.0 * out, t), out) is None       # not the error OF `outputs`
    assert mse_loss_spec(torch.nn.functional.mse_loss(out, lin(torch.randn(6, 4))), out) is None  # differentiated targets
    assert mse_loss_spec(torch.nn.functional.l1_loss(out, t), out) is None
    assert mse_loss_spec(torch.nn.functional.mse_loss(out, t, reduction="none").mean(), out) is None
    assert mse_loss_spec(torch.nn.functional.mse_loss(out, t) + 0.1 * out.sum(), out) is None
    leaf = out.detach().clone().requires_grad_(True)  # (logits handed out as a leaf by a persistent session)
    assert mse_loss_spec(torch.nn.functional.mse_loss(leaf, t), leaf) is not None
