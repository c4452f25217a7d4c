"""GPU: the fused curvature engine (``pytorchhessianfree_amd.engine``) -- explicit tangent /
adjoint sweeps over conv-BatchNorm units on the package's own convolution kernels, split-K
partial results summed by the consumer kernels -- against float64 autograd products of the
STOCK model (the contract of BackPACK's ``ggn_vector_product_from_plist``,
``/root/reference/hessianfree/optimizer.py:457-462``).

Stated fp32 tolerance: 1e-6 max-norm relative (2.5e-7 measured) on the ResNet-18 workload (stock fp32 autograd
itself sits at 2e-7); on the badly conditioned Bottleneck net no worse than 3x stock fp32 autograd; products
bitwise repeatable."""

import warnings

import pytest
import torch
from tol import within

import pytorchhessianfree_amd as hf
from pytorchhessianfree_amd import curvature, modelprep
from pytorchhessianfree_amd import testproblems as tp
from pytorchhessianfree_amd.engine import FusedGGNEngine

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _float64_product(make, v, masks=None, **kw):
    """float64 autograd product of the STOCK model; ``masks``: the ReLU sign decisions to take instead of the
    float64 network's own, in call order (the product of the same piecewise-linear network an fp32 operator
    linearised: one ReLU input within fp32 rounding of zero, decided the other way, moves a product of a
    deep net by ~1e-4 of its max-norm -- no error of either side)."""
    model, (x, t), lossf = make(device=DEV, **kw)
    model, x = model.double(), x.double()
    if masks is not None:
        _replay_relu_decisions(model, masks)
    params = [p for p in model.parameters() if p.requires_grad]
    out = model(x)
    return curvature.GGNOperator(lossf(out, t), out, params)(v.double())


def _replay_relu_decisions(model, masks, max_flips=200, margin=1e-5):
    """Make every ``nn.ReLU`` call of ``model`` take the sign decisions ``masks`` (in call order) -- and BOUND what is
    being replayed: a decision may differ from the model's own (``input > 0``) only where the model's own input lies
    within ``margin`` x the layer's max-norm of zero (an fp32 forward pass is good to ~1e-6 of that), and in at most
    ``max_flips`` entries over the whole network.  A wrong mask -- a kernel bug -- would flip decisions of inputs far
    from zero, or many of them, and is not inherited by the reference product."""
    import types

    cursor, flips = [0], [0]

    def relu_forward(self, inp):
        m = masks[cursor[0]].to(device=inp.device)
        cursor[0] += 1
        own = inp > 0
        diff = own != (m > 0)
        n = int(diff.sum())
        if n:
            flips[0] += n
            worst = float(inp[diff].abs().max() / inp.abs().max())
            within(worst, margin, note=("replayed ReLU decision far from zero", cursor[0] - 1, n))
            within(flips[0], max_flips, strict=False, note="replayed ReLU decisions that differ from the model's own")
        return inp * m.to(dtype=inp.dtype)

    for mod in model.modules():
        if isinstance(mod, torch.nn.ReLU):
            mod.forward = types.MethodType(relu_forward, mod)
    return flips


@pytest.mark.parametrize("batch", [32, 5])
def test_resnet18_engine_product_matches_float64(batch):
    model, (x, t), lossf = tp.resnet18_mnist(batch_size=batch, device=DEV)
    modelprep.prepare_model(model, channels_last=True)
    params = [p for p in model.parameters() if p.requires_grad]
    out = model(x)
    op = curvature.ggn_operator(lossf(out, t), out, params)
    assert isinstance(op, FusedGGNEngine)
    v = torch.randn(op.n, device=DEV, generator=torch.Generator(device=DEV).manual_seed(3))
    got = op(v).clone()
    for _ in range(3):
        assert torch.equal(op(v), got)  # the reference's _test_mvp_deterministic, bitwise
    # float64 autograd of the STOCK model on the engine's own ReLU decisions: no dependence on data seeds whose
    # pre-activations happen to stay clear of zero (the plain float64 product is checked too where it applies)
    masks = [(u.y > 0) for u in op.units if u.relu]
    want = _float64_product(tp.resnet18_mnist, v, masks=masks, batch_size=batch)
    within(float((got.double() - want).abs().max() / want.abs().max()), 1e-6)  # (2.5e-7 measured)
    # the stem's launch carries the v_W scatter (hf_conv2d_nhwc_slabs_unpack): same bits as the two launches
    assert op._carry_ok
    op._carry_ok = False
    assert torch.equal(op(v), got)
    op._carry_ok = True
    # linear in v, symmetric operator: <u, G v> == <v, G u>
    u = torch.randn(op.n, device=DEV, generator=torch.Generator(device=DEV).manual_seed(4))
    a, b = float(u.double() @ got.double()), float(v.double() @ op(u).double())
    within(abs(a - b), 1e-5 * abs(a), strict=False)


def test_bottleneck_net_engine_product_matches_float64():
    """Bottleneck (ResNet-50) blocks: the engine's product against float64 autograd of the stock model ON THE
    ENGINE'S OWN ReLU DECISIONS, fixed bound 5e-6 (max-norm relative).  Against the float64 network's own
    decisions this deep random-init net gives 1e-6 ... 3e-4 from run to run for every fp32 implementation,
    stock autograd included: a handful of its ~10^6 pre-activations lie within fp32 rounding of zero."""
    model, (x, t), lossf = tp.resnet50_small_images(batch_size=4, device=DEV, image=32)
    modelprep.prepare_model(model, channels_last=True)
    params = [p for p in model.parameters() if p.requires_grad]
    out = model(x)
    op = curvature.ggn_operator(lossf(out, t), out, params)
    assert isinstance(op, FusedGGNEngine)  # (a regression that makes the engine decline must fail here)
    v = torch.randn(op.n, device=DEV, generator=torch.Generator(device=DEV).manual_seed(5))
    got = op(v)
    masks = [(u.y > 0) for u in op.units if u.relu]
    want = _float64_product(tp.resnet50_small_images, v, masks=masks, batch_size=4, image=32)
    err = float((got.double() - want).abs().max() / want.abs().max())
    worst, off = [], 0
    for name, p in model.named_parameters():
        a, b = got[off:off + p.numel()].double(), want[off:off + p.numel()]
        worst.append((float((a - b).abs().max() / want.abs().max()), name))
        off += p.numel()
    within(err, 5e-6, note=(err, sorted(worst, reverse=True)[:5]))
    assert torch.equal(op(v), got)


def test_engine_declines_what_it_does_not_know(monkeypatch):
    """NCHW-prepared models, nets of other families and other losses keep the autograd path."""
    # (no MIOpen find step for the NCHW shapes this test touches once: it cost 35 s of the suite)
    monkeypatch.setattr(torch.backends.cudnn, "benchmark", False)
    model, (x, t), lossf = tp.resnet18_mnist(batch_size=4, device=DEV)
    modelprep.prepare_model(model)  # NCHW
    out = model(x)
    params = list(model.parameters())
    assert type(curvature.ggn_operator(lossf(out, t), out, params)) is curvature.GGNOperator
    # a conv stack the plain-stack engine does not cover (a grouped convolution; no global pooling)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.ReLU(),
                              torch.nn.Conv2d(8, 8, 3, padding=1, groups=2), torch.nn.ReLU(),
                              torch.nn.AdaptiveAvgPool2d(1), torch.nn.Flatten()).to(DEV).eval()
    modelprep.prepare_model(net, channels_last=True)
    x, t = torch.rand(4, 3, 8, 8, device=DEV), torch.randint(0, 8, (4,), device=DEV)
    out = net(x)
    lossf = torch.nn.CrossEntropyLoss()
    assert type(curvature.ggn_operator(lossf(out, t), out, list(net.parameters()))) is curvature.GGNOperator
    # All-CNN-C with a loss that is not a plain cross-entropy: the plain-stack engine needs the closed form
    net, (x, t), _ = tp.allcnnc_cifar100(batch_size=4, device=DEV)
    modelprep.prepare_model(net, channels_last=True)
    out = net(x)
    mse = torch.nn.MSELoss()
    onehot = torch.nn.functional.one_hot(t, 100).float()
    assert type(curvature.ggn_operator(mse(out, onehot), out, list(net.parameters()))) is curvature.GGNOperator
    # HF_ENGINE_DEBUG=1: the reason a model was not taken is said aloud (quiet by default: not being covered is normal)
    monkeypatch.setenv("HF_ENGINE_DEBUG", "1")
    with pytest.warns(UserWarning, match="fused curvature engine .* not used"):
        curvature.ggn_operator(mse(out, onehot), out, list(net.parameters()))
    monkeypatch.delenv("HF_ENGINE_DEBUG")
    # HF_ENGINE=0: a model the engine covers stays on the autograd operator
    ce = torch.nn.CrossEntropyLoss()
    assert isinstance(curvature.ggn_operator(ce(out, t), out, list(net.parameters())), FusedGGNEngine)
    monkeypatch.setenv("HF_ENGINE", "0")
    out = net(x)
    assert type(curvature.ggn_operator(ce(out, t), out, list(net.parameters()))) is curvature.GGNOperator


@pytest.mark.parametrize("batch", [32, 3])
def test_allcnnc_plain_stack_engine_product_matches_float64_and_cpu_oracle(batch):
    """BASELINE.json configs[3]'s topology (All-CNN-C, examples/example_utils.py:59-83) on the
    plain-stack engine: GGN product against float64 autograd of the STOCK model (1e-6 max-norm
    relative) and, at batch 32, against the CPU oracle (BackPACK's published algorithm restated,
    oracle/backpack_restated.py: 1e-5), bitwise repeatable, symmetric."""
    from pytorchhessianfree_amd.engine import PlainStackEngine

    model, (x, t), lossf = tp.allcnnc_cifar100(batch_size=batch, device=DEV)
    modelprep.prepare_model(model, channels_last=True)
    params = [p for p in model.parameters() if p.requires_grad]
    out = model(x)
    op = curvature.ggn_operator(lossf(out, t), out, params)
    assert isinstance(op, PlainStackEngine)
    v = torch.randn(op.n, device=DEV, generator=torch.Generator(device=DEV).manual_seed(3))
    got = op(v).clone()
    for _ in range(3):
        assert torch.equal(op(v), got)
    want = _float64_product(tp.allcnnc_cifar100, v, batch_size=batch)
    within(float((got.double() - want).abs().max() / want.abs().max()), 1e-6)  # (2.2e-7 measured)
    u = torch.randn(op.n, device=DEV, generator=torch.Generator(device=DEV).manual_seed(4))
    a, b = float(u.double() @ got.double()), float(v.double() @ op(u).double())
    within(abs(a - b), 1e-5 * abs(a), strict=False)
    if batch == 32:
        from oracle import backpack_restated as bp
        from pytorchhessianfree_amd.utils import vector_to_parameter_list

        cm, (cx, ct), cl = tp.allcnnc_cifar100(batch_size=batch, device="cpu")
        cp = [p for p in cm.parameters() if p.requires_grad]
        co = cm(cx)
        ref = torch.cat([g.reshape(-1) for g in bp.ggn_vector_product_from_plist(
            cl(co, ct), co, cp, vector_to_parameter_list(v.cpu(), cp))])
        within(float((got.cpu() - ref).abs().max() / ref.abs().max()), 1e-5)
        # the engine's own forward pass and one-sweep gradient against CPU autograd in FLOAT64 (fp32 CPU autograd of
        # this net is itself 2e-6 ... 1e-5 from it, depending on the host's thread count: 5.4e-6 with 16 threads)
        import copy

        cm64 = copy.deepcopy(cm).double()
        cp64 = [p for p in cm64.parameters() if p.requires_grad]
        co64 = cm64(cx.double())
        grad = torch.cat([g.reshape(-1) for g in torch.autograd.grad(cl(co64, ct), cp64)])
        within(float((op.gradient().cpu().double() - grad).abs().max() / grad.abs().max()), 2e-6)
        within(float((op.logits.cpu().double() - co64.detach()).abs().max() / co64.detach().abs().max()), 2e-6)
        # ... and against what the REAL reference computed in the build container (golden ``products``: ``_Gv``
        # through the BackPACK restatement on the stock CPU model) -- in float64 (product / gradient 2e-6: what its
        # fp32 results are rounded from) and in fp32 (the same + three times the reference's OWN fp32 distance to float64,
        # which is 1.0e-5 for this net's gradient on 8 CPU threads and moves by 9e-6 with the thread count)
        from helpers import RefTrace

        ref_p = RefTrace("allcnnc", "products/ggn")
        RefTrace("allcnnc", "products").check_inputs(cp, cx)
        got_p = op(ref_p.probe().to(DEV))
        within(ref_p.vec_err64("", got_p), 2e-6)
        within(ref_p.vec_err("", got_p), ref_p.envelope("", 2e-6))
        ref_g = RefTrace("allcnnc", "products")
        within(ref_g.vec_err64("grad", op.gradient()), 2e-6)
        within(ref_g.vec_err("grad", op.gradient()), ref_g.envelope("grad", 2e-6))
        want_logits = torch.from_numpy(ref_g.array("logits"))
        within(float((op.logits.cpu() - want_logits).abs().max() / want_logits.abs().max()), 5e-6)


@pytest.mark.parametrize("l2", [0.0, 5e-4])
def test_allcnnc_engine_hessian_product_matches_float64_and_cpu_oracle(l2):
    """BASELINE.json configs[3]: the HESSIAN product of All-CNN-C (cross-entropy, optionally + the L2 term
    of examples/example_utils.py:77-81) on the plain-stack engine -- forward-over-reverse on the own
    kernels -- against float64 autograd double-backward of the STOCK model (2e-6 max-norm relative), against
    the CPU oracle (BackPACK's ``hessian_vector_product`` restated, reference call site
    optimizer.py:450-455: 1e-5), bitwise repeatable, symmetric."""
    from oracle import backpack_restated as bp
    from pytorchhessianfree_amd.engine import PlainStackEngine
    from pytorchhessianfree_amd.utils import vector_to_parameter_list

    def problem(device, dtype=torch.float32):
        model, (x, t), lossf = tp.allcnnc_cifar100(batch_size=32, device=device)
        model, x = model.to(dtype), x.to(dtype)
        if l2 > 0:
            lossf = tp.l2_regularized(lossf, model, l2)
        return model, x, t, lossf

    model, x, t, lossf = problem(DEV)
    modelprep.prepare_model(model, channels_last=True)
    params = [p for p in model.parameters() if p.requires_grad]
    out = model(x)
    op = curvature.hessian_operator(lossf(out, t), out, params)
    assert isinstance(op, PlainStackEngine) and op.hessian
    v = torch.randn(op.n, device=DEV, generator=torch.Generator(device=DEV).manual_seed(8))
    got = op(v).clone()
    for _ in range(2):
        assert torch.equal(op(v), got)
    m64, x64, t64, l64 = problem(DEV, torch.float64)
    p64 = [p for p in m64.parameters() if p.requires_grad]
    want = curvature.HessianOperator(l64(m64(x64), t64), p64)(v.double())
    within(float((got.double() - want).abs().max() / want.abs().max()), 2e-6)
    u = torch.randn(op.n, device=DEV, generator=torch.Generator(device=DEV).manual_seed(9))
    a, b = float(u.double() @ got.double()), float(v.double() @ op(u).double())
    within(abs(a - b), 1e-5 * abs(a), strict=False)
    cm, cx, ct, cl = problem("cpu")
    cp = [p for p in cm.parameters() if p.requires_grad]
    closs = cl(cm(cx), ct)
    ref = torch.cat([g.reshape(-1) for g in bp.hessian_vector_product(closs, cp, vector_to_parameter_list(v.cpu(), cp))])
    within(float((got.cpu() - ref).abs().max() / ref.abs().max()), 1e-5)
    # (gradient and loss against CPU autograd in FLOAT64: the fp32 CPU gradient of this net moves by 5e-6 with the
    # host's thread count)
    cm64, cx64, ct64, cl64 = problem("cpu", torch.float64)
    cp64 = [p for p in cm64.parameters() if p.requires_grad]
    closs64 = cl64(cm64(cx64), ct64)
    grad = torch.cat([g.reshape(-1) for g in torch.autograd.grad(closs64, cp64)])
    within(float((op.gradient().cpu().double() - grad).abs().max() / grad.abs().max()), 2e-6)
    within(abs(float(op.loss_buf) - float(closs64)), 1e-6 * abs(float(closs64)), strict=False)
    # the REAL reference's ``_Hv`` on the stock CPU model (golden): 1e-5
    from helpers import RefTrace

    ref_p = RefTrace("allcnnc", "hessian_l2_product" if l2 > 0 else "products/hessian")
    RefTrace("allcnnc", "products").check_inputs(cp, cx)
    got_p = op(ref_p.probe().to(DEV))
    within(ref_p.vec_err64("", got_p), 6e-6)  # (float64: the engine's Hessian products are 2e-6 from it)
    within(ref_p.vec_err("", got_p), ref_p.envelope("", 6e-6))


@pytest.mark.parametrize("batch", [32, 6])
def test_resnet18_engine_hessian_product_matches_float64_and_cpu_oracle(batch):
    """The HESSIAN product (``curvature_opt="hessian"``, reference optimizer.py:450-455) of the ResNet-18 of
    examples/run_resnet18_mnist.py with eval-mode BatchNorm on the fused engine: forward-over-reverse on the own
    kernels -- per unit the plain stack's ``conv_D(g, V)`` / ``conv_W(t_x, g)`` plus the BatchNorm scale's own
    second-order terms, residual adds, the max-pool and the linear head.  Against float64 autograd double-backward
    of the STOCK model on the engine's ReLU / pooling decisions (2e-6 max-norm relative), against the CPU oracle
    (BackPACK's ``hessian_vector_product`` restated: 1e-5), bitwise repeatable, symmetric."""
    from oracle import backpack_restated as bp
    from pytorchhessianfree_amd.utils import vector_to_parameter_list

    seed = tp.RESNET18_B32_SEPARATED_SEEDS[0]
    model, (x, t), lossf = tp.resnet18_mnist(batch_size=batch, device=DEV, data_seed=seed)
    modelprep.prepare_model(model, channels_last=True)
    params = [p for p in model.parameters() if p.requires_grad]
    out = model(x)
    op = curvature.hessian_operator(lossf(out, t), out, params)
    assert isinstance(op, FusedGGNEngine) and op.hessian and type(op) is FusedGGNEngine
    v = torch.randn(op.n, device=DEV, generator=torch.Generator(device=DEV).manual_seed(13))
    got = op(v).clone()
    for _ in range(2):
        assert torch.equal(op(v), got)
    masks = [(u.y > 0) for u in op.units if u.relu]
    m64, (x64, t64), l64 = tp.resnet18_mnist(batch_size=batch, device=DEV, data_seed=seed)
    m64 = m64.double()
    _replay_relu_decisions(m64, masks)
    p64 = [p for p in m64.parameters() if p.requires_grad]
    want = curvature.HessianOperator(l64(m64(x64.double()), t64), p64)(v.double())
    err = float((got.double() - want).abs().max() / want.abs().max())
    worst, off = [], 0
    for name, p in model.named_parameters():
        a, b = got[off:off + p.numel()].double(), want[off:off + p.numel()]
        worst.append((float((a - b).abs().max() / want.abs().max()), name))
        off += p.numel()
    within(err, 2e-6, note=(err, sorted(worst, reverse=True)[:6]))
    u = torch.randn(op.n, device=DEV, generator=torch.Generator(device=DEV).manual_seed(14))
    a, b = float(u.double() @ got.double()), float(v.double() @ op(u).double())
    within(abs(a - b), 1e-5 * abs(a), strict=False)
    cm, (cx, ct), cl = tp.resnet18_mnist(batch_size=batch, device="cpu", data_seed=seed)
    cp = [p for p in cm.parameters() if p.requires_grad]
    closs = cl(cm(cx), ct)
    ref = torch.cat([g.reshape(-1) for g in bp.hessian_vector_product(closs, cp, vector_to_parameter_list(v.cpu(), cp))])
    within(float((got.cpu() - ref).abs().max() / ref.abs().max()), 1e-5)
    grad = torch.cat([g.reshape(-1) for g in torch.autograd.grad(closs, cp)])
    within(float((op.gradient().cpu() - grad).abs().max() / grad.abs().max()), 2e-6)
    if batch == 32:  # the REAL reference's ``_Hv`` on the stock CPU model (golden ``hessian_product``): 1e-5
        from helpers import RefTrace

        ref_p = RefTrace("resnet18", "hessian_product")
        RefTrace("resnet18", "solve_martens").check_inputs(cp, cx)
        got_p = op(ref_p.probe().to(DEV))
        within(ref_p.vec_err64("", got_p), 6e-6)
        within(ref_p.vec_err("", got_p), ref_p.envelope("", 6e-6))
    # the GGN product of the same engine family is untouched by the Hessian bookkeeping
    out2 = model(x)
    ggn = curvature.ggn_operator(lossf(out2, t), out2, params)
    assert isinstance(ggn, FusedGGNEngine) and not ggn.hessian


def test_resnet18_train_mode_hessian_product_and_solve_on_the_engine():
    """``curvature_opt="hessian"`` on the TRAIN-mode ResNet-18 -- the model of examples/run_resnet18_mnist.py:19-35, which
    never calls ``model.eval()`` -- on the fused engine: the batch statistics' second-order terms by
    ``hf_bn_train_hessian_coeffs`` / ``hf_bn_train_hessian_apply`` (reference optimizer.py:450-455 through BackPACK's
    double backward).  Against (i) float64 double backward of the STOCK train-mode model on the engine's own ReLU
    decisions: 2e-5 max-norm relative (6.6e-6 measured; batch 32: 3.1e-6, stock fp32 autograd there: 6.1e-6); bitwise
    repeatable, symmetric;
    (ii) the REAL reference's ``_Hv`` on the stock CPU model (golden ``train_hessian_product``, batch 16): the
    envelope rule; gradient likewise; (iii) a six-iteration damped solve against the reference's ``cg`` on its own
    product (golden ``train_hessian_solve``): same termination reason and iteration count, the non-positive-curvature
    warnings at the same iterations, iterates rel-l2."""
    import warnings

    from helpers import RefTrace

    ref = RefTrace("resnet18_train_hessian", "train_hessian_solve")
    prod = RefTrace("resnet18_train_hessian", "train_hessian_product")
    cm, (cx, ct), lossf = tp.resnet18_mnist(batch_size=16, device="cpu", data_seed=5)
    ref.check_inputs(list(cm.parameters()), cx)
    model, x, t = cm.to(DEV), cx.to(DEV), ct.to(DEV)
    model.train()
    modelprep.prepare_model(model, channels_last=True)
    params = [p for p in model.parameters() if p.requires_grad]
    out = model(x)
    op = curvature.hessian_operator(lossf(out, t), out, params)
    assert type(op) is FusedGGNEngine and op.hessian and op.train_bn and op.train_own
    v = torch.randn(op.n, device=DEV, generator=torch.Generator(device=DEV).manual_seed(13))
    got = op(v).clone()
    for _ in range(2):
        assert torch.equal(op(v), got)
    masks = [(u.y > 0) for u in op.units if u.relu]
    m64, (x64, t64), l64 = tp.resnet18_mnist(batch_size=16, device=DEV, data_seed=5)
    m64 = m64.double().train()
    _replay_relu_decisions(m64, masks)
    p64 = [p for p in m64.parameters() if p.requires_grad]
    want = curvature.HessianOperator(l64(m64(x64.double()), t64), p64)(v.double())
    err = float((got.double() - want).abs().max() / want.abs().max())
    worst, off = [], 0
    for name, p in model.named_parameters():
        a, b = got[off:off + p.numel()].double(), want[off:off + p.numel()]
        worst.append((float((a - b).abs().max() / want.abs().max()), name))
        off += p.numel()
    within(err, 2e-5, note=(err, sorted(worst, reverse=True)[:6]))  # (batch 16: 6.6e-6 measured; batch 32: 3.1e-6)
    u = torch.randn(op.n, device=DEV, generator=torch.Generator(device=DEV).manual_seed(14))
    a, b = float(u.double() @ got.double()), float(v.double() @ op(u).double())
    within(abs(a - b), 1e-4 * abs(a), strict=False)
    # (ii) the reference's own numbers
    got_p = op(prod.probe().to(DEV))
    within(prod.vec_err64("", got_p), 1e-5)
    within(prod.vec_err("", got_p), prod.envelope("", 1e-5))
    grad = op.gradient()
    within(ref.vec_err64("grad", grad), 1e-5)
    within(ref.vec_err("grad", grad), ref.envelope("grad", 1e-5))
    # (iii) the solve
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        xs, ms, reason = hf.cg(hf.DampedCurvature(op, 1.0), -grad, max_iter=6, martens_conv_crit=True,
                               store_x_at_iters=list(range(7)))
    assert reason == str(ref.array("reason")) and len(xs) - 1 == int(ref.scalar("n_iters"))
    nonpos = sorted(int(str(w.message).split("iteration ")[1].split(".")[0]) for w in rec
                    if "Directional curvature" in str(w.message))
    assert nonpos == ref.array("nonpos_iters").tolist(), (nonpos, ref.array("nonpos_iters"))
    for i in range(1, len(xs)):
        rel = ref.vec_rel_l2(f"x/{i}", xs[i])
        within(rel, 1e-3, note=(i, rel))


def test_train_mode_hessian_step_through_the_session_equals_the_autograd_path():
    """One default ``HessianFree.step()`` with ``curvature_opt="hessian"`` on the train-mode ResNet-18: the persistent
    session over the Hessian engine against this package's autograd path (``curvature.HessianOperator``: double
    backward through the stock train-mode layers on MIOpen) -- same damping schedule and termination reason,
    iteration counts +-2, initial loss 1e-5, final loss 1e-3."""
    import warnings

    def run(engine):
        model, (x, t), lossf = tp.resnet18_mnist(batch_size=16, device=DEV, data_seed=5)
        model.train()
        modelprep.prepare_model(model, channels_last=True)
        opt = hf.HessianFree(model.parameters(), curvature_opt="hessian", graph_matvec=engine, cg_max_iter=20)
        if not engine:
            opt._session_off = True

        def forward():
            out = model(x)
            return lossf(out, t), out

        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            final = opt.step(forward)
        return opt, final

    a, fa = run(True)
    assert a._session is not None and a._session.engine.hessian and a._session.engine.train_bn
    b, fb = run(False)
    assert b._session is None
    within(abs(a.state["init_losses"][0] - b.state["init_losses"][0]), 1e-5 * abs(b.state["init_losses"][0]), strict=False)
    assert a.state["dampings"] == b.state["dampings"] and a.state["cg_reasons"] == b.state["cg_reasons"]
    within(abs(a.state["num_cg_iters"][0] - b.state["num_cg_iters"][0]), 2, strict=False)
    within(abs(fa - fb), 1e-3 * abs(fb), strict=False)
    assert fa < a.state["init_losses"][0]


def test_resnet18_hessian_step_through_the_session_matches_reference_trace():
    """One default ``HessianFree.step()`` with ``curvature_opt="hessian"`` on the ResNet-18 workload through the
    persistent session over the Hessian engine, against the reference's own step (golden ``hessian_step``: stock
    model, double backward through the BackPACK restatement, ``hessianfree.cg.cg``): initial loss 1e-5, damping /
    learning rate / reason identical, iterations +-2.  The Hessian of this random-init ReLU net is INDEFINITE at
    damping 1.0: CG meets directions of negative curvature, its fp32 iterates blow up and recover (cg.py:133-139), and
    back-tracking then picks between stored iterates whose losses differ in the third digit -- measured 2.1963 (GPU)
    against 2.2035 / 2.2079 (CPU runs with 128 / 8 threads: the reference side scatters as much), products themselves
    equal to 1e-5 (test above): final loss 3e-2, and the run
    must reduce the loss."""
    from helpers import RefTrace

    ref = RefTrace("resnet18", "hessian_step")
    sc, fc = ref.state, ref.finals[0]
    seed = tp.RESNET18_B32_SEPARATED_SEEDS[0]
    model, (x, t), lossf = tp.resnet18_mnist(batch_size=32, device="cpu", data_seed=seed)
    ref.check_inputs(list(model.parameters()), x, step=0)
    model, x, t = model.to(DEV), x.to(DEV), t.to(DEV)
    modelprep.prepare_model(model, channels_last=True)
    opt = hf.HessianFree(model.parameters(), curvature_opt="hessian", graph_matvec=True)

    def forward():
        o = model(x)
        return lossf(o, t), o

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        fg = opt.step(forward)
    assert opt._session is not None and opt._session.engine.hessian
    sg = opt.state
    within(abs(sg["init_losses"][0] - sc["init_losses"][0]), 1e-5 * abs(sc["init_losses"][0]), strict=False)
    assert sg["dampings"] == sc["dampings"] and sg["learning_rates"] == sc["learning_rates"]
    assert sg["cg_reasons"] == sc["cg_reasons"]
    within(abs(sg["num_cg_iters"][0] - sc["num_cg_iters"][0]), 2, strict=False)
    within(abs(fg - fc), 3e-2 * abs(fc), strict=False)
    assert fg < sg["init_losses"][0] and fc < sc["init_losses"][0]


def test_train_mode_batchnorm_engine_product_matches_cpu_oracle_and_float64():
    """What ``examples/run_resnet18_mnist.py:19-35`` actually runs -- no ``model.eval()``: BatchNorm normalises
    with BATCH statistics and the GGN couples the samples.  The engine takes such a model (single GPU): tangent
    and adjoint of every BatchNorm carry the statistics' dependence on the layer input (two per-channel sums per
    sweep, folded into the elementwise kernels).  Product against the CPU oracle (BackPACK's algorithm restated
    on the stock train-mode model: 5e-5, the tolerance of test_optimizer_gpu's stock-layer test) and against
    float64 autograd on the GPU (1e-5), both on the engine's ReLU decisions (batch normalisation centres the
    pre-activations at zero: with ~10^6 of them some always lie within fp32 rounding of it, and the plain
    float64 product is 2e-3 away from EVERY fp32 product, stock autograd included); bitwise repeatable."""
    from oracle import backpack_restated as bp
    from pytorchhessianfree_amd.utils import vector_to_parameter_list

    seed = tp.RESNET18_B32_SEPARATED_SEEDS[1]
    model, (x, t), lossf = tp.resnet18_mnist(batch_size=16, device=DEV, data_seed=seed)
    model.train()
    modelprep.prepare_model(model, channels_last=True)
    params = [p for p in model.parameters() if p.requires_grad]
    out = model(x)
    op = curvature.ggn_operator(lossf(out, t), out, params)
    assert isinstance(op, FusedGGNEngine) and op.train_bn
    v = torch.randn(op.n, device=DEV, generator=torch.Generator(device=DEV).manual_seed(31))
    got = op(v).clone()
    assert torch.equal(op(v), got)
    masks = [(u.y > 0) for u in op.units if u.relu]
    # float64 stock model in train mode, same data
    m64, (x64, t64), l64 = tp.resnet18_mnist(batch_size=16, device=DEV, data_seed=seed)
    m64 = m64.double().train()
    _replay_relu_decisions(m64, masks)
    p64 = [p for p in m64.parameters() if p.requires_grad]
    o64 = m64(x64.double())
    want = curvature.GGNOperator(l64(o64, t64), o64, p64)(v.double())
    within(float((got.double() - want).abs().max() / want.abs().max()), 1e-5)
    cm, (cx, ct), cl = tp.resnet18_mnist(batch_size=16, device="cpu", data_seed=seed)
    cm.train()
    _replay_relu_decisions(cm, [m.cpu() for m in masks])
    cp = [p for p in cm.parameters() if p.requires_grad]
    co = cm(cx)
    ref = torch.cat([g.reshape(-1) for g in bp.ggn_vector_product_from_plist(
        cl(co, ct), co, cp, vector_to_parameter_list(v.cpu(), cp))])
    within(float((got.cpu() - ref).abs().max() / ref.abs().max()), 5e-5)
    # a full step through the hipGraph-replayed engine decreases the loss
    opt = hf.HessianFree(model.parameters(), graph_matvec=True, cg_max_iter=30)

    def forward():
        o = model(x)
        return lossf(o, t), o

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        final = opt.step(forward)
    assert final < opt.state["init_losses"][0]


def test_train_mode_prologue_form_variants_agree_and_state_is_independent_of_the_first_use_check(monkeypatch):
    """Train-mode BatchNorm inside the product: the per-channel finalisation runs in the prologue of the elementwise
    pass (``hf_chan_affine_train``); the tangent's partial sums come from the convolution's own epilogue
    (``hf_conv2d_nhwc_group_slabs_bnsum``) or -- ``HF_BN_EPILOGUE=0`` -- from the reduction launch; the downsample
    blocks' two units share launches (``hf_chan_affine_train_pair``) or not (``HF_BN_TRAIN_PAIR=0``).  Same workgroup
    code with and without the pair launches: the same bits.  Epilogue vs reduction launch: the epilogue multiplies
    every SPLIT's partial tile by xhat and adds the products up in fp64, the reduction launch multiplies the fp32 sum
    of the slabs -- 0.8e-6 ... 1.1e-6 of the product's max-norm measured over ten data seeds on two leases
    (gpurun_out/r5a/diag2.jsonl): bound 5e-6.  Each form is bitwise repeatable.

    GPUTEST_r04's red test, root cause: the engine's linearisation point after construction
    depended on whether its first-use check ran -- skipped, a train-mode engine stayed at the model's recorded
    activations with its OWN batch statistics -- and the check's registry was keyed by ``id(model)``, an address CPython
    reuses.  So: an engine built with ``HF_ENGINE_VERIFY=never`` and one built with ``always`` must hold bitwise the
    same activations, statistics and masks, and give bitwise the same product."""
    seed = tp.RESNET18_B32_SEPARATED_SEEDS[2]
    v = None
    products, states = {}, {}

    def build():
        model, (x, t), lossf = tp.resnet18_mnist(batch_size=16, device=DEV, data_seed=seed)
        model.train()
        modelprep.prepare_model(model, channels_last=True)
        params = [p for p in model.parameters() if p.requires_grad]
        out = model(x)
        op = curvature.ggn_operator(lossf(out, t), out, params)
        assert isinstance(op, FusedGGNEngine) and op.train_bn and op.train_own
        return op

    for form in ("default", "no-pair", "no-epilogue", "verify-never", "verify-always"):
        monkeypatch.setenv("HF_BN_EPILOGUE", "0" if form == "no-epilogue" else "1")
        monkeypatch.setenv("HF_BN_TRAIN_PAIR", "0" if form == "no-pair" else "1")
        monkeypatch.setenv("HF_ENGINE_VERIFY", form.split("-")[1] if form.startswith("verify") else "first")
        op = build()
        # the tangent's partial sums by the convolution's own epilogue: every unit but the im2col'd stem, unless off
        assert sum(u.epi for u in op.units) == (0 if form == "no-epilogue" else len(op.units) - 1)
        if v is None:
            v = torch.randn(op.n, device=DEV, generator=torch.Generator(device=DEV).manual_seed(41))
        got = op(v).clone()
        for _ in range(10):
            assert torch.equal(op(v), got)
        assert all(u.tsum == u.epi for u in op.units)
        products[form] = got
        states[form] = [t.clone() for u in op.units for t in (u.a, u.y, u.mean_t, u.rstd)]
    assert torch.equal(products["default"], products["no-pair"])
    ref = products["no-epilogue"]
    within(float((products["default"] - ref).abs().max() / ref.abs().max()), 5e-6)
    for form in ("verify-never", "verify-always"):
        assert torch.equal(products[form], products["default"]), form
        assert all(torch.equal(a, b) for a, b in zip(states[form], states["default"])), form
    # forward pass: the one-pass statistics (E[a^2] - mean^2 in fp64) against torch's batch_norm on the same summed
    # convolution output, per unit: batch means 1e-6, rstd 5e-6 (max-norm relative); running statistics moved once
    saved = [(u.bn.running_mean.clone(), u.bn.running_var.clone(), u.bn.num_batches_tracked.clone()) for u in op.units]

    def rel(a, b):
        return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))

    op.forward_own(update_running=True)
    logits = op.logits.clone()
    for u, (rm, rv, nb) in zip(op.units, saved):
        a64 = u.a.double()
        mean = a64.mean(dim=(0, 2, 3))
        var = a64.var(dim=(0, 2, 3), unbiased=False)
        cnt = a64.numel() / a64.shape[1]
        assert rel(u.mean_t.double(), mean) < 1e-6 and rel(u.rstd.double(), (var + u.bn.eps).rsqrt()) < 5e-6
        mom = u.bn.momentum
        within(rel(u.bn.running_mean.double(), (1 - mom) * rm.double() + mom * mean), 1e-6)
        within(rel(u.bn.running_var.double(), (1 - mom) * rv.double() + mom * var * cnt / (cnt - 1)), 5e-6)
        assert int(u.bn.num_batches_tracked) == int(nb) + 1
    for u, (rm, rv, nb) in zip(op.units, saved):
        u.bn.running_mean.copy_(rm)
        u.bn.running_var.copy_(rv)
        u.bn.num_batches_tracked.copy_(nb)
    op.forward_own(update_running=True)
    assert torch.equal(op.logits, logits)  # (bitwise repeatable)


@pytest.mark.parametrize("case", ["allcnnc_l2", "allcnnc_hessian", "resnet18", "resnet18_sum", "resnet18_frozen"])
def test_engine_diag_ef_matches_per_sample_autograd(case):
    """The diagonal of the empirical Fisher ``(1/N) sum_i g_i^2`` (reference preconditioners.py:11-105: one backward
    pass per sample, or BackPACK's ``SumGradSquared``) from ONE adjoint sweep of the engine + per-sample weight-gradient
    launches + squaring gathers (``engine.diag_ef``) against the per-sample autograd loop on the same model
    (``diag_EF_autograd``): 1e-5 max-norm relative -- All-CNN-C with the tagged L2 term (each per-sample loss carries it
    whole: closed form), the Hessian engine (cotangents kept in ``g1`` / ``ga1``), ResNet-18 (BatchNorm parameters per
    sample, linear head in closed form), reduction ``sum``."""
    reduction = "sum" if case.endswith("_sum") else "mean"
    if case.startswith("allcnnc"):
        model, (x, t), lossf = tp.allcnnc_cifar100(batch_size=8, device=DEV, data_seed=3)
        lossf = tp.l2_regularized(lossf, model, 5e-4)
    else:
        model, (x, t), _ = tp.resnet18_mnist(batch_size=8, device=DEV, data_seed=tp.RESNET18_B32_SEPARATED_SEEDS[0])
        lossf = torch.nn.CrossEntropyLoss(reduction=reduction)
        if case == "resnet18_frozen":  # (round 6: stem + layer1 frozen -- dead units are skipped, no entries for them)
            tp.freeze_stem_and_layer1(model)
            list(model.layers)[4].bn2.weight.requires_grad_(False)  # ... and one frozen scale inside the live region
    modelprep.prepare_model(model, channels_last=True)
    params = [p for p in model.parameters() if p.requires_grad]
    out = model(x)
    hessian = case == "allcnnc_hessian"
    make = curvature.hessian_operator if hessian else curvature.ggn_operator
    eng = make(lossf(out, t), out, params)
    assert isinstance(eng, FusedGGNEngine) and eng.hessian == hessian
    got = eng.diag_ef(reduction).clone()
    # (the reference's loop, preconditioners.py:91-99, on batches of ONE sample: BatchNorm2d wants 4-D inputs)
    want = torch.zeros_like(got)
    for i in range(x.shape[0]):
        g_i = torch.autograd.grad(lossf(model(x[i:i + 1]), t[i:i + 1]), params)
        want += torch.cat([g.reshape(-1) for g in g_i]) ** 2
    if reduction == "mean":
        want /= x.shape[0]
    if case.startswith("allcnnc"):
        ref = hf.diag_EF_autograd(model, lossf, x, t, reduction)
        within(float((ref - want).abs().max() / want.abs().max()), 1e-6)
    within(float((got - want).abs().max() / want.abs().max()), 1e-5)
    assert torch.equal(eng.diag_ef(reduction), got)  # repeatable
    # the products of the same engine are untouched by the per-sample bookkeeping
    v = torch.randn(eng.n, device=DEV, generator=torch.Generator(device=DEV).manual_seed(5))
    a = eng(v).clone()
    eng.diag_ef(reduction)
    assert torch.equal(eng(v), a)


def test_get_preconditioner_uses_the_sessions_engine_from_the_second_step(monkeypatch):
    """``HessianFree.get_preconditioner`` (optimizer.py:928-952) with a persistent session for the model: the diagonal
    comes from the engine (no per-sample backward passes) and equals the autograd construction to 1e-5; before the
    first step (no session yet) and for another input shape the autograd construction runs."""
    from pytorchhessianfree_amd import optimizer_session as hfopt  # (where get_preconditioner lives)

    model, (x, t), lossf0 = tp.allcnnc_cifar100(batch_size=8, device=DEV, data_seed=4)
    lossf = tp.l2_regularized(lossf0, model, 5e-4)
    modelprep.prepare_model(model, channels_last=True)
    opt = hf.HessianFree(model.parameters(), curvature_opt="hessian", graph_matvec=True, cg_max_iter=5)
    calls = []
    real = hfopt.diag_EF_preconditioner
    monkeypatch.setattr(hfopt, "diag_EF_preconditioner", lambda *a, **k: calls.append(1) or real(*a, **k))

    def forward():
        o = model(x)
        return lossf(o, t), o

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        M0 = opt.get_preconditioner(model, lossf, x, t, "mean", use_backpack=False)   # no session yet
        assert len(calls) == 1
        opt.step(forward, M_func=M0)
        assert opt._session is not None
        M1 = opt.get_preconditioner(model, lossf, x, t, "mean", use_backpack=False)   # the engine's sweep
        assert len(calls) == 1
        want = hf.diag_EF_autograd(model, lossf, x, t, "mean")
        within(float((M1.diag - want).abs().max() / want.abs().max()), 1e-5)
        assert M1.damping == opt.param_groups[0]["damping"]
        final = opt.step(forward, M_func=M1)
        assert final <= opt.state["init_losses"][-1]
        opt.get_preconditioner(model, lossf, x[:4], t[:4], "mean", use_backpack=False)  # another shape: autograd
        assert len(calls) == 2


def test_resnet18_engine_product_matches_cpu_oracle_at_batch_32():
    """The engine's product DIRECTLY against the CPU oracle (oracle/backpack_restated.py: BackPACK's
    ``ggn_vector_product_from_plist`` restated; reference call site optimizer.py:457-462) at
    BASELINE.json config 2's batch 32: 1e-5 max-norm relative."""
    from oracle import backpack_restated as bp
    from pytorchhessianfree_amd.utils import vector_to_parameter_list

    seed = tp.RESNET18_B32_SEPARATED_SEEDS[0]
    model, (x, t), lossf = tp.resnet18_mnist(batch_size=32, device=DEV, data_seed=seed)
    modelprep.prepare_model(model, channels_last=True)
    params = [p for p in model.parameters() if p.requires_grad]
    out = model(x)
    op = curvature.ggn_operator(lossf(out, t), out, params)
    assert isinstance(op, FusedGGNEngine)
    v = torch.randn(op.n, device=DEV, generator=torch.Generator(device=DEV).manual_seed(21))
    got = op(v).cpu()
    cm, (cx, ct), cl = tp.resnet18_mnist(batch_size=32, device="cpu", data_seed=seed)
    cp = [p for p in cm.parameters() if p.requires_grad]
    co = cm(cx)
    ref = torch.cat([g.reshape(-1) for g in bp.ggn_vector_product_from_plist(
        cl(co, ct), co, cp, vector_to_parameter_list(v.cpu(), cp))])
    within(float((got - ref).abs().max() / ref.abs().max()), 1e-5)
    # the REAL reference's ``_Gv`` on the stock CPU model (golden ``ggn_product``): 1e-5
    from helpers import RefTrace

    ref_p = RefTrace("resnet18", "ggn_product")
    RefTrace("resnet18", "solve_martens").check_inputs(cp, cx)
    got_p = op(ref_p.probe().to(DEV))
    within(ref_p.vec_err64("", got_p), 2e-6)
    within(ref_p.vec_err("", got_p), ref_p.envelope("", 2e-6))


def test_step_with_engine_graph_and_data_parallel_weight():
    """The engine behind ``HessianFree.step(graph_matvec=True)``: one hipGraph per PCG iteration
    whose product is the engine's launches; loss decreases, result equals the eager-engine step."""
    results = []
    for graph in (False, True):
        model, (x, t), lossf = tp.resnet18_mnist(batch_size=16, device=DEV)
        modelprep.prepare_model(model, channels_last=True)
        opt = hf.HessianFree(model.parameters(), cg_max_iter=20, graph_matvec=graph)

        def forward():
            o = model(x)
            return lossf(o, t), o

        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            final = opt.step(forward)
        results.append((opt.state["init_losses"][0], final, opt.state["num_cg_iters"][0],
                        torch.cat([p.detach().reshape(-1) for p in model.parameters()])))
    (i0, f0, n0, p0), (i1, f1, n1, p1) = results
    assert f0 < i0 and f1 < i1
    assert n0 == n1 and abs(f0 - f1) <= 1e-5 * abs(f0)
    within(float((p0 - p1).abs().max()), 1e-5 * float(p0.abs().max()), strict=False)


def _cl(t):
    return t.contiguous(memory_format=torch.channels_last)


@pytest.mark.parametrize("geom", [(32, 64, 14, 14, 3, 2, 1), (3, 8, 9, 7, 2, 2, 0), (2, 12, 10, 10, 3, 1, 1),
                                  (2, 4, 7, 7, 3, 3, 1)])
def test_maxpool_kernels_match_autograd(geom):
    """``hf_maxpool_tangent_nhwc`` / ``hf_maxpool_adjoint_nhwc`` (one launch each, gather form, slab
    sums of two cotangents) against ATen's max-pool JVP / backward in float64."""
    from pytorchhessianfree_amd import _lib

    n, c, h, w, k, s, p = geom
    gen = torch.Generator(device=DEV).manual_seed(sum(geom))
    x = _cl(torch.randn(n, c, h, w, device=DEV, generator=gen))
    y, idx = torch.nn.functional.max_pool2d(x, k, s, p, return_indices=True)
    oh, ow = y.shape[2], y.shape[3]
    idx32 = idx.permute(0, 2, 3, 1).contiguous().to(torch.int32)
    lib, st = _lib.load(), _lib.current_stream_ptr(x.device)
    P = _lib.c_void_p

    t = _cl(torch.randn(n, c, h, w, device=DEV, generator=gen))
    wide = _cl(torch.zeros(n, 2 * c, oh, ow, device=DEV))
    _lib.check(lib.hf_maxpool_tangent_nhwc(P(wide.data_ptr()), P(t.data_ptr()), P(idx32.data_ptr()), n, h, w, oh, ow,
                                           c, 2 * c, _lib.HF_F32, st), "tangent")
    want = t.flatten(2).gather(2, idx.flatten(2)).view_as(y)
    assert torch.equal(wide[:, :c], want) and float(wide[:, c:].abs().max()) == 0.0

    # adjoint: 3 slabs of one cotangent + 1 of the other
    a = torch.randn(3, n, oh, ow, c, device=DEV, generator=gen)
    b = torch.randn(1, n, oh, ow, c, device=DEV, generator=gen)
    g = _cl(torch.empty(n, c, h, w, device=DEV))
    _lib.check(lib.hf_maxpool_adjoint_nhwc(
        P(g.data_ptr()), P(a.data_ptr()), 3, a[0].numel(), P(b.data_ptr()), 1, 0, P(idx32.data_ptr()), n, h, w, oh, ow,
        c, k, k, s, s, p, p, _lib.HF_F32, st), "adjoint")
    gy = (a.double().sum(0) + b[0].double()).permute(0, 3, 1, 2)
    x64 = x.double().requires_grad_(True)
    (want,) = torch.autograd.grad(torch.nn.functional.max_pool2d(x64, k, s, p), x64, gy)
    within(float((g.double() - want).abs().max()), 1e-6 * float(want.abs().max()), strict=False)
    g2 = torch.empty_like(g)
    _lib.check(lib.hf_maxpool_adjoint_nhwc(
        P(g2.data_ptr()), P(a.data_ptr()), 3, a[0].numel(), P(b.data_ptr()), 1, 0, P(idx32.data_ptr()), n, h, w, oh, ow,
        c, k, k, s, s, p, p, _lib.HF_F32, st), "adjoint")
    assert torch.equal(g, g2)


@pytest.mark.parametrize("shape", [(32, 512, 10, True), (5, 64, 3, False), (16, 256, 28, True), (7, 260, 11, True), (64, 128, 5, True), (130, 64, 12, False)])
def test_linear_ce_head_kernel_matches_float64(shape):
    """``hf_linear_ce_head``: logits' tangent, softmax-CE Hessian and the head's three gradients in
    one launch, against the same chain in float64 (stated tolerance 2e-6 of each result's max)."""
    from pytorchhessianfree_amd import _lib

    b, f, k, bias = shape
    gen = torch.Generator(device=DEV).manual_seed(b + f + k)
    r = lambda *s: torch.randn(*s, device=DEV, generator=gen)
    t_feat, feat, w, v_w, v_b, logits = r(b, f), r(b, f).abs(), r(k, f) / f**0.5, r(k, f), r(k), r(b, k)
    p = torch.softmax(logits, 1)
    scale = 1.0 / b
    slabs = _lib.load().hf_linear_ce_head_slabs(b)  # g_w / g_b arrive as per-workgroup partial sums
    assert slabs == (b + 3) // 4
    g_feat = torch.empty(b, f, device=DEV)
    g_w, g_b = torch.empty(slabs, k, f, device=DEV), torch.empty(slabs, k, device=DEV)
    P = _lib.c_void_p
    rc = _lib.load().hf_linear_ce_head(
        P(g_feat.data_ptr()), P(g_w.data_ptr()), P(g_b.data_ptr()) if bias else None, P(t_feat.data_ptr()),
        P(feat.data_ptr()), P(w.data_ptr()), P(v_w.data_ptr()), P(v_b.data_ptr()) if bias else None, P(p.data_ptr()),
        scale, b, f, k, _lib.HF_F32, _lib.current_stream_ptr(feat.device))
    assert rc == 0
    d = lambda t: t.double()
    jv = d(t_feat) @ d(w).t() + d(feat) @ d(v_w).t() + (d(v_b) if bias else 0.0)
    hjv = scale * d(p) * (jv - (d(p) * jv).sum(1, keepdim=True))
    for got, want in ((g_feat, hjv @ d(w)), (g_w.sum(0), hjv.t() @ d(feat))) + (((g_b.sum(0), hjv.sum(0)),) if bias else ()):
        within(float((got.double() - want).abs().max()), 2e-6 * float(want.abs().max()), strict=False)


def test_linear_ce_head_refuses_large_heads():
    from pytorchhessianfree_amd import _lib

    x = torch.zeros(16, device=DEV)
    P = _lib.c_void_p
    a = P(x.data_ptr())
    lib = _lib.load()
    assert lib.hf_linear_ce_head(a, a, a, a, a, a, a, a, a, 1.0, 4, 2048, 10, _lib.HF_F32, None) == -1   # features
    assert lib.hf_linear_ce_head(a, a, a, a, a, a, a, a, a, 1.0, 4, 512, 1000, _lib.HF_F32, None) == -1  # classes
    assert lib.hf_linear_ce_head(a, a, a, a, a, a, a, a, a, 1.0, 64, 512, 64, _lib.HF_F32, None) == -1    # LDS


def test_pack_and_unpack_skip_structurally_zero_taps():
    """``live`` masks of ``hf_pack_ex`` / ``hf_unpack_tangent_ex``: the slices of kernel taps that
    never meet data (3x3 kernel on a 1x1 map: 8 of 9) are not read -- same vector as without the
    mask when those gradient entries are zero, which they structurally are."""
    from pytorchhessianfree_amd import _lib
    from pytorchhessianfree_amd.engine import _live_taps

    assert _live_taps(1, 1, 3, 3, (1, 1), (1, 1)) == 1 << 4           # centre tap only
    assert _live_taps(2, 2, 3, 3, (2, 2), (1, 1)) == 0b110110000       # layer4.0.conv1: taps (1..2, 1..2)
    assert _live_taps(2, 2, 3, 3, (1, 1), (1, 1)) == 0                 # every tap meets data somewhere
    assert _live_taps(7, 7, 3, 3, (1, 1), (1, 1)) == 0
    gen = torch.Generator(device=DEV).manual_seed(11)
    k, c = 24, 16
    masks = {0: 1 << 4, 2: 0b110110000}
    # three weight gradients stored (O, H, W, I) = channels_last [O, I, 3, 3]; #1 has no dead taps
    grads, splits = [], {}
    for i in range(3):
        nsp = (1, 3, 2)[i]
        g = torch.randn(nsp, k, 3, 3, c, device=DEV, generator=gen)
        m = masks.get(i)
        if m is not None:
            dead = torch.tensor([not (m >> t) & 1 for t in range(9)], device=DEV).view(3, 3)
            g[:, :, dead] = 0.0
        grads.append(g)
        if nsp > 1:
            splits[i] = (nsp, g[0].numel())
    n = sum(g[0].numel() for g in grads)
    perms = {i: (c, 9) for i in range(3)}
    firsts = [g[0].reshape(-1) for g in grads]
    plain = _lib.pack_ex(torch.empty(n, device=DEV), firsts, perms, splits, scale=0.5)
    # poison the dead slices: with the mask they must not be read
    for i, m in masks.items():
        dead = torch.tensor([not (m >> t) & 1 for t in range(9)], device=DEV).view(3, 3)
        grads[i][:, :, dead] = float("nan")
    masked = _lib.pack_ex(torch.empty(n, device=DEV), firsts, perms, splits, scale=0.5, live=masks)
    assert torch.equal(plain, masked)
    want = torch.cat([0.5 * g.sum(0).permute(0, 3, 1, 2).reshape(-1) for g in
                      [torch.nan_to_num(g, nan=0.0) for g in grads]])
    within(float((masked - want).abs().max()), 1e-6, strict=False)

    # unpack: only live slices are written
    v = torch.randn(2 * k * c * 9 + 5, device=DEV, generator=gen)
    bufs = [torch.full((k, 2 * c, 3, 3), 7.0, device=DEV).contiguous(memory_format=torch.channels_last)
            for _ in range(2)]
    _lib.unpack_tangent(v, [(5, bufs[0], c, 1 << 4), (5 + k * c * 9, bufs[1], c, 0)])
    src0 = v[5:5 + k * c * 9].view(k, c, 3, 3)
    assert torch.equal(bufs[0][:, c:, 1, 1], src0[:, :, 1, 1])
    rest = bufs[0][:, c:].clone()
    rest[:, :, 1, 1] = 7.0
    assert float((rest - 7.0).abs().max()) == 0.0 and float((bufs[0][:, :c] - 7.0).abs().max()) == 0.0
    assert torch.equal(bufs[1][:, c:], v[5 + k * c * 9:].view(k, c, 3, 3))


def test_solve_as_one_graph_per_iteration_equals_separate_launches(monkeypatch):
    """The PCG loop on the ResNet-18 engine's captured product as ONE hipGraph launch per iteration (product -> K1 -> K2
    -> K3, default) and with product, K1, K2, K3 as separate launches (``HF_FUSE_ITERATION=0``): the same kernels on
    the same data in the same order -- identical termination, bitwise equal iterates and m_k."""
    model, (x, t), lossf = tp.resnet18_mnist(batch_size=8, device=DEV, data_seed=tp.RESNET18_B32_SEPARATED_SEEDS[0])
    modelprep.prepare_model(model, channels_last=True)
    params = [p for p in model.parameters() if p.requires_grad]
    b = None
    runs = {}
    for name, env in (("graph", {}), ("separate", {"HF_FUSE_ITERATION": "0"})):
        monkeypatch.delenv("HF_FUSE_ITERATION", raising=False)
        for k, val in env.items():
            monkeypatch.setenv(k, val)

        def builder():
            o = model(x)
            return curvature.ggn_operator(lossf(o, t), o, params)

        op = curvature.maybe_graphed(builder, params=params)
        assert isinstance(op.op, FusedGGNEngine)
        if b is None:
            b = torch.randn(op.n, device=DEV, generator=torch.Generator(device=DEV).manual_seed(3))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            xs, ms, reason = hf.cg(hf.DampedCurvature(op, 0.05), b, max_iter=12, martens_conv_crit=True,
                                   store_x_at_iters=list(range(13)))
        assert bool(getattr(op, "_iteration_graphs", {})) == (name == "graph")
        runs[name] = (xs, ms, reason)
    (xs0, ms0, reason0), (xs, ms, reason) = runs["graph"], runs["separate"]
    assert reason == reason0 and len(xs) == len(xs0)
    for i in range(1, len(xs)):
        assert torch.equal(xs[i], xs0[i]) and float(ms[i]) == float(ms0[i]), i


def test_live_copy_gathers_and_scatters_the_live_entries():
    """``hf_live_copy``: dense segments and conv-weight segments with a live-tap mask, against
    indexing with an explicit boolean mask; scatter restores exactly the gathered entries."""
    from pytorchhessianfree_amd import _lib

    _check_live_copy([(0, 37, 0, 0), (37, 270, 9, 1 << 4), (307, 11, 0, 0), (318, 108, 9, 0b110110000), (426, 5, 0, 0)],
                     431)
    # several blocks per segment (2048 compact entries each), a tap count that does not divide the block's
    # share (6 of 9), unaligned dense runs
    big, off = [], 3
    for cnt, per, mask in [(70001, 0, 0), (300 * 40 * 9, 9, 1 << 4), (4099, 0, 0), (128 * 64 * 9, 9, 0b110110110),
                           (96 * 48 * 9, 9, 0b110110000), (8192, 0, 0)]:
        big.append((off, cnt, per, mask))
        off += cnt + (5 if per == 0 else 0)  # (gaps: entries that belong to no segment stay untouched)
    _check_live_copy(big, off + 2)


def _check_live_copy(segs, n):
    from pytorchhessianfree_amd import _lib

    gen = torch.Generator(device=DEV).manual_seed(21)
    # [dense 37 | weight 6x5x(3x3), centre tap | dense 11 | weight 4x3x(3x3), 4 taps | dense 5]
    keep = torch.zeros(n, dtype=torch.bool, device=DEV)
    for off, cnt, per, mask in segs:
        keep[off:off + cnt] = True
        if per:
            taps = torch.tensor([(mask >> t) & 1 for t in range(per)], dtype=torch.bool, device=DEV)
            keep[off:off + cnt] = taps.repeat(cnt // per)
    full = torch.randn(n, device=DEV, generator=gen)
    m = int(keep.sum())
    comp = torch.full((m,), -1.0, device=DEV)
    arr = lambda c: (_lib.c_int64 * len(segs))(*[s[c] for s in segs])
    P = _lib.c_void_p
    st = _lib.current_stream_ptr(full.device)
    lib = _lib.load()
    assert lib.hf_live_copy(P(full.data_ptr()), P(comp.data_ptr()), 0, arr(0), arr(1), arr(2), arr(3), len(segs),
                            _lib.HF_F32, st) == 0
    assert torch.equal(comp, full[keep])
    target, doubled = torch.zeros(n, device=DEV), 2 * comp
    assert lib.hf_live_copy(P(target.data_ptr()), P(doubled.data_ptr()), 1, arr(0), arr(1), arr(2), arr(3),
                            len(segs), _lib.HF_F32, st) == 0
    assert torch.equal(target[keep], 2 * full[keep]) and float(target[~keep].abs().max()) == 0.0


def test_grouped_launches_equal_the_single_ones_bitwise():
    """``hf_conv2d_nhwc_group_slabs`` (4 convolutions of different directions in one launch),
    ``hf_chan_affine_pair`` and ``hf_chan_affine_bwd_pair`` write exactly what the corresponding
    single launches write."""
    from pytorchhessianfree_amd import _lib

    gen = torch.Generator(device=DEV).manual_seed(77)
    r_ = lambda *s: torch.randn(*s, device=DEV, generator=gen)
    n, h, w_, c, k = 8, 7, 7, 64, 128
    x = _cl(r_(n, c, h, w_))
    w3 = _cl(r_(k, c, 3, 3))
    w1 = _cl(r_(k, c, 1, 1))
    geo3 = (n, h, w_, c, k, 3, 3, (2, 2), (1, 1))   # a block's first convolution ...
    geo1 = (n, h, w_, c, k, 1, 1, (2, 2), (0, 0))   # ... and its downsample branch: both 7 -> 4
    gy = _cl(r_(n, k, 4, 4))
    w3T, w1T = w3.permute(1, 2, 3, 0).contiguous(), w1.permute(1, 2, 3, 0).contiguous()
    probs = [(0, geo3, x, w3, n * 16 * k), (0, geo1, x, w1, n * 16 * k), (1, geo3, gy, w3T, x.numel()),
             (2, geo1, x, gy, w1.numel())]
    singles, grouped, group_args = [], [], []
    for d, geo, act, mat, numel in probs:
        sp = _lib.conv_plan(d, *geo[:7], geo[7], geo[8])
        a, b = torch.zeros(sp, numel, device=DEV), torch.zeros(sp, numel, device=DEV)
        _lib.conv2d_nhwc_slabs(d, a, act, mat, *geo[:7], geo[7], geo[8], sp)
        singles.append(a)
        grouped.append(b)
        group_args.append((d, b, act, mat, geo, sp, 0, 0))
    _lib.conv_group_slabs(group_args, x.device)
    for a, b in zip(singles, grouped):
        assert torch.equal(a, b)

    # BatchNorm tangent / adjoint pairs against the single launches
    lib, st, P = _lib.load(), _lib.current_stream_ptr(x.device), _lib.c_void_p
    rows, chans = (n * 16, n * 16), (128, 64)
    aff, adj = (_lib.AffineProblem * 2)(), (_lib.BnAdjointProblem * 2)()
    keep, want_t, want_a = [], [], []
    for q, qa, rws, ch in zip(aff, adj, rows, chans):
        slabs = r_(3, rws * ch)
        xx, yy, mean, rstd, wt, vq, vr = r_(rws, ch), r_(rws, ch), r_(ch), r_(ch).abs() + 0.5, r_(ch), r_(ch), r_(ch)
        out1, out2 = torch.empty(rws, ch, device=DEV), torch.empty(rws, ch, device=DEV)
        keep += [slabs, xx, yy, mean, rstd, wt, vq, vr, out1, out2]
        _lib.check(lib.hf_chan_affine_ex(P(out1.data_ptr()), P(slabs.data_ptr()), P(xx.data_ptr()), P(mean.data_ptr()),
                                         P(rstd.data_ptr()), P(wt.data_ptr()), P(vq.data_ptr()), P(vr.data_ptr()), None,
                                         P(yy.data_ptr()), 0, rws, ch, 1, 1, 0, 0, 3, rws * ch, _lib.HF_F32, st), "single")
        q.out, q.a, q.x, q.mean, q.rstd, q.w = (t.data_ptr() for t in (out2, slabs, xx, mean, rstd, wt))
        q.q, q.r, q.add, q.mask_src, q.relu_self = vq.data_ptr(), vr.data_ptr(), None, yy.data_ptr(), 0
        q.n, q.c, q.hw, q.out_ld, q.add_ld, q.a_splits, q.a_slab = rws, ch, 1, 0, 0, 3, rws * ch
        want_t.append((out1, out2))
        rb = 8
        outs1 = [torch.empty(rws, ch, device=DEV), torch.empty(rb, ch, device=DEV), torch.empty(rb, ch, device=DEV),
                 torch.empty(rws, ch, device=DEV)]
        outs2 = [torch.empty_like(t) for t in outs1]
        g2 = r_(rws, ch)
        keep += outs1 + outs2 + [g2]
        _lib.check(lib.hf_chan_affine_bwd_ex(*(P(t.data_ptr()) for t in outs1), P(slabs.data_ptr()), 3, rws * ch,
                                             P(g2.data_ptr()), 1, 0, P(xx.data_ptr()), P(mean.data_ptr()),
                                             P(rstd.data_ptr()), P(wt.data_ptr()), P(yy.data_ptr()), rws, ch, 1, 1, rb,
                                             _lib.HF_F32, st), "single bwd")
        qa.gx, qa.gw, qa.gb, qa.gres = (t.data_ptr() for t in outs2)
        qa.gy, qa.gy_splits, qa.gy_slab = slabs.data_ptr(), 3, rws * ch
        qa.gy2, qa.gy2_splits, qa.gy2_slab = g2.data_ptr(), 1, 0
        qa.x, qa.mean, qa.rstd, qa.w, qa.mask_src = (t.data_ptr() for t in (xx, mean, rstd, wt, yy))
        qa.n, qa.c, qa.hw, qa.row_blocks = rws, ch, 1, rb
        want_a.append((outs1, outs2))
    _lib.check(lib.hf_chan_affine_pair(_lib.ctypes.cast(aff, P), _lib.HF_F32, st), "pair")
    _lib.check(lib.hf_chan_affine_bwd_pair(_lib.ctypes.cast(adj, P), _lib.HF_F32, st), "pair bwd")
    for a, b in want_t:
        assert torch.equal(a, b)
    for o1, o2 in want_a:
        for a, b in zip(o1, o2):
            assert torch.equal(a, b)


@pytest.mark.parametrize("rows,ch,splits,rb", [(1568, 64, 5, 63), (512, 128, 9, 64), (128, 256, 17, 32), (32, 512, 3, 1),
                                               (200, 96, 1, 7)])
def test_train_mode_batchnorm_launches_against_float64_formulas(rows, ch, splits, rb):
    """The launches of a train-mode BatchNorm unit, each against the float64 formula it implements (C ABI, no engine):
    ``hf_chan_affine_bwd_ex`` (partial rows) + ``hf_chan_affine_train`` (adds them up in its prologue) =
    ``mask * (w*rstd * [t - mean(t) - xhat*mean(xhat*t)] + xhat*vq + vr + add)`` with ``t`` the sum of the slabs;
    ``hf_bn_stats_rows`` + ``hf_bn_forward_train`` = ``F.batch_norm(training=True)`` + residual + ReLU,
    batch statistics and moved running statistics included.  Tolerances: 2e-6 of the output's max-norm."""
    from pytorchhessianfree_amd import _lib

    gen = torch.Generator(device=DEV).manual_seed(rows + ch)

    def r_(*shape):
        return torch.randn(*shape, device=DEV, generator=gen)

    lib, st, P = _lib.load(), _lib.current_stream_ptr(torch.device(DEV)), _lib.c_void_p

    def p(t):
        return P(t.data_ptr()) if t is not None else None

    slabs = r_(splits, rows * ch)
    a, y, add = r_(rows, ch) * 2 + 0.5, r_(rows, ch), r_(rows, 2 * ch)      # conv output, ReLU mask source, residual
    w, vq, vr = r_(ch), r_(ch), r_(ch)
    mean = a.mean(0).contiguous()
    rstd = (1.0 / (a.var(0, unbiased=False) + 1e-5).sqrt()).contiguous()
    t64 = slabs.double().sum(0).view(rows, ch)
    xhat = (a.double() - mean.double()) * rstd.double()
    core = w.double() * rstd.double() * (t64 - t64.mean(0) - xhat * (xhat * t64).mean(0))
    want = core + xhat * vq.double() + vr.double() + add[:, :ch].double()
    want = torch.where(y > 0, want, torch.zeros_like(want))
    tol = 2e-6 * float(want.abs().max())
    # reduction launch + elementwise pass with the prologue
    gw, gb = torch.empty(rb, ch, device=DEV), torch.empty(rb, ch, device=DEV)
    _lib.check(lib.hf_chan_affine_bwd_ex(None, p(gw), p(gb), None, p(slabs), splits, rows * ch, None, 1, 0, p(a), p(mean),
                                         p(rstd), None, None, rows, ch, 1, 1, rb, _lib.HF_F32, st), "reduction")
    out = torch.full((rows, 2 * ch), float("nan"), device=DEV)
    _lib.check(lib.hf_chan_affine_train(p(out), p(slabs), p(a), p(mean), p(rstd), p(w), p(gw), p(gb), rb, p(vq), p(vr),
                                        float(rows), p(add), p(y), rows, ch, 1, 2 * ch, 2 * ch, splits, rows * ch,
                                        _lib.HF_F32, st), "hf_chan_affine_train")
    assert float((out[:, :ch].double() - want).abs().max()) < tol
    assert torch.isnan(out[:, ch:]).all()  # (the other half of the wider buffer is not touched)
    again = torch.empty_like(out)
    _lib.check(lib.hf_chan_affine_train(p(again), p(slabs), p(a), p(mean), p(rstd), p(w), p(gw), p(gb), rb, p(vq), p(vr),
                                        float(rows), p(add), p(y), rows, ch, 1, 2 * ch, 2 * ch, splits, rows * ch,
                                        _lib.HF_F32, st), "hf_chan_affine_train")
    assert torch.equal(again[:, :ch], out[:, :ch])
    # forward: one-pass statistics' partial rows + the normalising launch that finalises them
    part = torch.empty(rb, 2, ch, dtype=torch.float64, device=DEV)
    asum = torch.empty(rows, ch, device=DEV)
    bn_w, bn_b, res = r_(ch), r_(ch), r_(rows, ch)
    rm, rv = r_(ch), r_(ch).abs() + 0.5
    rm0, rv0 = rm.clone(), rv.clone()
    _lib.check(lib.hf_bn_stats_rows(p(asum), p(slabs), splits, rows * ch, p(part), rows, ch, rb, _lib.HF_F32, st),
               "hf_bn_stats_rows")
    assert torch.equal(asum.view(-1), slabs.sum(0)) or float((asum.view(-1) - slabs.sum(0)).abs().max()) < 1e-5
    m_out, r_out = torch.empty(ch, device=DEV), torch.empty(ch, device=DEV)
    yy, y2 = torch.empty(rows, ch, device=DEV), torch.full((rows, 2 * ch), float("nan"), device=DEV)
    _lib.check(lib.hf_bn_forward_train(p(yy), p(y2[:, ch:]), 2 * ch, p(asum), p(part), rb, p(m_out), p(r_out), p(rm),
                                       p(rv), float(rows), 1e-5, 0.1, p(bn_w), p(bn_b), p(res), 0, 1, rows, ch,
                                       _lib.HF_F32, st), "hf_bn_forward_train")
    a64 = asum.double()
    ref = torch.nn.functional.batch_norm(a64.t().reshape(1, ch, rows), rm0.double().clone(), rv0.double().clone(),
                                         bn_w.double(), bn_b.double(), True, 0.1, 1e-5).reshape(ch, rows).t()
    ref = torch.relu(ref + res.double())
    within(float((yy.double() - ref).abs().max()), 2e-6 * float(ref.abs().max()))
    assert torch.equal(y2[:, ch:], yy) and torch.isnan(y2[:, :ch]).all()
    within(float((m_out.double() - a64.mean(0)).abs().max()), 1e-6 * float(a64.mean(0).abs().max()))
    within(float((r_out.double() * (a64.var(0, unbiased=False) + 1e-5).sqrt() - 1).abs().max()), 2e-6)
    within(float((rm.double() - (0.9 * rm0.double() + 0.1 * a64.mean(0))).abs().max()), 1e-6)
    within(float((rv.double() - (0.9 * rv0.double() + 0.1 * a64.var(0, unbiased=True))).abs().max()), 2e-6 * float(rv.abs().max()))


def _freeze(model, pattern):
    blocks = list(model.layers)
    if pattern == "stem_conv_only":           # frozen weight under a trainable BatchNorm: conv(x, 0) = 0, not dead
        model.conv1.weight.requires_grad_(False)
    elif pattern == "mid_block_conv2":        # a frozen weight INSIDE the live region (tangent flows through it)
        blocks[3].conv2.weight.requires_grad_(False)
    elif pattern == "all_batchnorm":          # every scale / shift frozen, every convolution trained
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.requires_grad_(False)
                m.bias.requires_grad_(False)
    elif pattern == "stem_and_first_block":   # dead prefix that ends INSIDE layer1: an identity block reads a tangent-free input
        for mod in (model.conv1, model.bn1, blocks[0]):
            for p in mod.parameters():
                p.requires_grad_(False)
    elif pattern == "up_to_layer3":           # dead prefix of six blocks: the first live block has a downsample branch
        for mod in (model.conv1, model.bn1, *blocks[:6]):
            for p in mod.parameters():
                p.requires_grad_(False)
    elif pattern == "stem_layer1_train":
        tp.freeze_stem_and_layer1(model)
        model.train()
    else:
        raise ValueError(pattern)
    return model


@pytest.mark.parametrize("pattern", ["mid_block_conv2", "all_batchnorm", "stem_and_first_block", "up_to_layer3",
                                     "stem_layer1_train"])
def test_engine_hessian_products_with_frozen_parameter_patterns_match_float64(pattern):
    """The same patterns under ``curvature_opt="hessian"`` (optimizer.py:450-455): forward-over-reverse with the
    second-order terms of dead units skipped, ``V = 0`` for a frozen weight, ``v_gamma = 0`` for a frozen scale.  Against
    float64 double backward of the stock model with the same tensors frozen, on the engine's ReLU decisions: 2e-6
    (train mode 1.5e-5), bitwise repeatable; the one-sweep gradient 2e-6."""
    def make(device=DEV, **kw):
        model, data, lossf = tp.resnet18_mnist(device=device, **kw)
        return _freeze(model, pattern), data, lossf

    model, (x, t), lossf = make(batch_size=16, data_seed=3)
    modelprep.prepare_model(model, channels_last=True)
    params = [p for p in model.parameters() if p.requires_grad]
    out = model(x)
    why = []
    op = FusedGGNEngine.try_build(lossf(out, t), out, params, hessian=True, why=why)
    assert isinstance(op, FusedGGNEngine) and op.hessian and op.frozen_any, why
    v = torch.randn(op.n, device=DEV, generator=torch.Generator(device=DEV).manual_seed(6))
    got = op(v).clone()
    assert torch.equal(op(v), got)
    ref, (rx, rt), rl = make(batch_size=16, data_seed=3)
    ref, rx = ref.double(), rx.double()
    _replay_relu_decisions(ref, [(u.y > 0) for u in op.units if u.relu])
    rp = [p for p in ref.parameters() if p.requires_grad]
    rloss = rl(ref(rx), rt)
    want = curvature.HessianOperator(rloss, rp)(v.double())
    train = pattern == "stem_layer1_train"
    # (train mode through batch statistics behind a frozen prefix: 4.1e-6 measured; stated 1.5e-5)
    within(float((got.double() - want).abs().max() / want.abs().max()), 1.5e-5 if train else 2e-6)
    grad = torch.cat([g.reshape(-1) for g in torch.autograd.grad(rloss, rp)])
    # (the one-sweep gradient through batch statistics: 4.1e-6 measured; stated 1.5e-5)
    within(float((op.gradient().double() - grad).abs().max() / grad.abs().max()), 1.5e-5 if train else 2e-6)


@pytest.mark.parametrize("pattern", ["stem_conv_only", "mid_block_conv2", "all_batchnorm", "stem_and_first_block",
                                     "up_to_layer3", "stem_layer1_train"])
def test_engine_products_with_frozen_parameter_patterns_match_float64(pattern):
    """The engine "in the subspace of trainable parameters" (reference optimizer.py:121-123, utils.py:31-32) for frozen
    patterns beyond the fixture's stem + layer1: frozen tensors inside the live region (no tangent term, no gather
    entry), dead prefixes that end inside a stage or in front of a downsample block, train-mode BatchNorm behind a
    frozen prefix.  Each against float64 autograd of the stock model with the same tensors frozen, on the engine's own
    ReLU decisions: 1e-6 (train mode 1.5e-5), bitwise repeatable."""
    def make(device=DEV, **kw):
        model, data, lossf = tp.resnet18_mnist(device=device, **kw)
        return _freeze(model, pattern), data, lossf

    model, (x, t), lossf = make(batch_size=16, data_seed=3)
    modelprep.prepare_model(model, channels_last=True)
    params = [p for p in model.parameters() if p.requires_grad]
    out = model(x)
    why = []
    op = FusedGGNEngine.try_build(lossf(out, t), out, params, why=why)
    assert isinstance(op, FusedGGNEngine), why
    assert op.frozen_any and op.n == sum(p.numel() for p in params)
    expect_dead = {"stem_and_first_block": 1, "up_to_layer3": 6, "stem_layer1_train": 2}.get(pattern, 0)
    assert op.dead_blocks == expect_dead and op.stem.dead == (expect_dead > 0)
    v = torch.randn(op.n, device=DEV, generator=torch.Generator(device=DEV).manual_seed(5))
    got = op(v).clone()
    assert torch.equal(op(v), got)
    if pattern == "stem_layer1_train":
        ref, (rx, rt), rl = make(batch_size=16, data_seed=3)
        ref, rx = ref.double(), rx.double()
        _replay_relu_decisions(ref, [(u.y > 0) for u in op.units if u.relu])
        rp = [p for p in ref.parameters() if p.requires_grad]
        ro = ref(rx)
        want = curvature.GGNOperator(rl(ro, rt), ro, rp)(v.double())
        within(float((got.double() - want).abs().max() / want.abs().max()), 1.5e-5)  # (3.8e-6 measured)
    else:
        want = _float64_product(make, v, masks=[(u.y > 0) for u in op.units if u.relu], batch_size=16, data_seed=3)
        within(float((got.double() - want).abs().max() / want.abs().max()), 1e-6)


@pytest.mark.parametrize("hessian", [False, True], ids=["ggn", "hessian"])
def test_allcnnc_plain_stack_engine_with_frozen_layers_matches_float64(hessian):
    """The plain-stack engine (All-CNN-C, BASELINE configs[3]'s topology) on a trainable subset: the first two
    convolution layers frozen (dead for both sweeps: the tangent sweep starts at the third layer with a tangent-free
    input, the adjoint sweep ends there without a data gradient), plus a frozen bias and a frozen weight inside the live
    region.  GGN and Hessian products against float64 autograd of the stock model with the same tensors frozen (1e-6 /
    2e-6; All-CNN-C's ReLU inputs at this seed stay clear of zero: the plain float64 product), bitwise repeatable;
    gradient 2e-6; the diagonal empirical Fisher against the per-sample autograd loop 1e-5."""
    from pytorchhessianfree_amd.engine import PlainStackEngine

    def make(device=DEV, **kw):
        model, data, lossf = tp.allcnnc_cifar100(device=device, **kw)
        convs = [m for m in model.modules() if isinstance(m, torch.nn.Conv2d)]
        for m in convs[:2]:
            for p in m.parameters():
                p.requires_grad_(False)
        convs[4].bias.requires_grad_(False)
        convs[6].weight.requires_grad_(False)
        return model, data, lossf

    model, (x, t), lossf = make(batch_size=8, data_seed=3)
    modelprep.prepare_model(model, channels_last=True)
    params = [p for p in model.parameters() if p.requires_grad]
    out = model(x)
    why = []
    op = FusedGGNEngine.try_build(lossf(out, t), out, params, hessian=hessian, why=why)
    assert isinstance(op, PlainStackEngine) and op.frozen_any and op.dead_units == 2, why
    v = torch.randn(op.n, device=DEV, generator=torch.Generator(device=DEV).manual_seed(8))
    got = op(v).clone()
    assert torch.equal(op(v), got)
    ref, (rx, rt), rl = make(batch_size=8, data_seed=3)
    ref, rx = ref.double(), rx.double()
    rp = [p for p in ref.parameters() if p.requires_grad]
    ro = ref(rx)
    rloss = rl(ro, rt)
    want = (curvature.HessianOperator(rloss, rp) if hessian else curvature.GGNOperator(rloss, ro, rp))(v.double())
    within(float((got.double() - want).abs().max() / want.abs().max()), 2e-6 if hessian else 1e-6)
    grad = torch.cat([g.reshape(-1) for g in torch.autograd.grad(rloss, rp)])
    within(float((op.gradient().double() - grad).abs().max() / grad.abs().max()), 2e-6)
    got_d = op.diag_ef("mean").clone()
    want_d = torch.zeros_like(got_d)
    for i in range(x.shape[0]):
        g_i = torch.autograd.grad(lossf(model(x[i:i + 1]), t[i:i + 1]), params)
        want_d += torch.cat([g.reshape(-1) for g in g_i]) ** 2
    want_d /= x.shape[0]
    within(float((got_d - want_d).abs().max() / want_d.abs().max()), 1e-5)
    assert torch.equal(op(v), got)


@pytest.mark.parametrize("case", ["ggn", "hessian", "ggn_train", "hessian_train", "ggn_frozen", "ggn_sum"])
def test_mse_head_products_match_float64(case):
    """The mean-squared-error head (``nn.MSELoss``: the loss of the reference's examples and tests, examples/run_mwe.py:19;
    ``_Gv`` / ``_Hv`` take any loss, optimizer.py:450-462) under the engine's variants: GGN and Hessian products, eval- and
    train-mode BatchNorm, frozen stem + layer1, reduction ``sum`` -- against float64 autograd of the stock model on the
    engine's ReLU decisions (eval 1e-6 / Hessian 2e-6, train mode 1.5e-5); gradient 2e-6 / 1.5e-5; bitwise repeatable."""
    hessian, train = case.startswith("hessian"), case.endswith("_train")
    reduction = "sum" if case.endswith("_sum") else "mean"

    def make(device=DEV, **kw):
        model, data, _ = tp.resnet18_mnist_mse(device=device, **kw)
        if case.endswith("_frozen"):
            tp.freeze_stem_and_layer1(model)
        if train:
            model.train()
        return model, data, torch.nn.MSELoss(reduction=reduction)

    model, (x, t), lossf = make(batch_size=16, data_seed=3)
    modelprep.prepare_model(model, channels_last=True)
    params = [p for p in model.parameters() if p.requires_grad]
    out = model(x)
    why = []
    op = FusedGGNEngine.try_build(lossf(out, t), out, params, hessian=hessian, why=why)
    assert isinstance(op, FusedGGNEngine) and op.hessian == hessian, why
    assert op.loss_spec is not None and op.loss_spec["kind"] == "mse" and op.loss_spec["reduction"] == reduction
    v = torch.randn(op.n, device=DEV, generator=torch.Generator(device=DEV).manual_seed(9))
    got = op(v).clone()
    assert torch.equal(op(v), got)
    ref, (rx, rt), rl = make(batch_size=16, data_seed=3)
    ref, rx, rt = ref.double(), rx.double(), rt.double()
    _replay_relu_decisions(ref, [(u.y > 0) for u in op.units if u.relu])
    rp = [p for p in ref.parameters() if p.requires_grad]
    ro = ref(rx)
    rloss = rl(ro, rt)
    want = (curvature.HessianOperator(rloss, rp) if hessian else curvature.GGNOperator(rloss, ro, rp))(v.double())
    tol = 1.5e-5 if train else (2e-6 if hessian else 1e-6)
    within(float((got.double() - want).abs().max() / want.abs().max()), tol)
    grad = torch.cat([g.reshape(-1) for g in torch.autograd.grad(rloss, rp)])
    within(float((op.gradient().double() - grad).abs().max() / grad.abs().max()), 1.5e-5 if train else 2e-6)
    within(abs(float(op.loss_buf) - float(rloss)), 2e-6 * abs(float(rloss)), strict=False)
