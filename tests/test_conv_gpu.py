"""GPU: the package's implicit-GEMM convolution kernels (``hf_conv2d_nhwc``: fp32 MFMA,
deterministic split-K, dead taps skipped) against float64 convolutions, through the C ABI.

These kernels replace, inside the curvature product, the three MIOpen calls per conv layer
that BackPACK's R-op / L-op issue through PyTorch (``/root/reference/hessianfree/
optimizer.py:461``).  Stated tolerance: fp32 fma-chain accuracy, ``3e-6 * sum|a*b|`` scale
(max-norm error below 2e-5 of the result's max-norm at the longest reductions here), and
BITWISE equality of two launches on the same inputs (what MIOpen's split-K kernels with
atomic accumulation do not give)."""

import pytest
import torch
from tol import within

from pytorchhessianfree_amd import _lib

pytestmark = pytest.mark.gpu
DEV = "cuda"

# (N, H, W, C, K, R, S, stride, padding)
GEOMS = [
    (32, 7, 7, 64, 64, 3, 3, (1, 1), (1, 1)),      # ResNet-18 layer1
    (32, 7, 7, 64, 128, 3, 3, (2, 2), (1, 1)),     # layer2.0.conv1 (7 -> 4)
    (32, 7, 7, 64, 128, 1, 1, (2, 2), (0, 0)),     # layer2.0.downsample
    (32, 4, 4, 128, 128, 3, 3, (1, 1), (1, 1)),    # layer2
    (32, 2, 2, 256, 256, 3, 3, (1, 1), (1, 1)),    # layer3: every tap meets data somewhere
    (32, 2, 2, 256, 512, 3, 3, (2, 2), (1, 1)),    # layer4.0.conv1: 4 of 9 taps live
    (32, 1, 1, 512, 512, 3, 3, (1, 1), (1, 1)),    # layer4: centre tap only
    (32, 7, 7, 128, 64, 3, 3, (1, 1), (1, 1)),     # tangent operands: 2*Cin channels
    (4, 5, 6, 12, 20, 3, 2, (2, 1), (1, 0)),       # ragged everything
    (2, 9, 9, 8, 100, 5, 5, (1, 1), (2, 2)),       # K not a multiple of the tile
    (3, 6, 5, 36, 8, 1, 1, (1, 1), (0, 0)),
    (8, 16, 16, 96, 96, 3, 3, (1, 1), (1, 1)),     # All-CNN-C-like: many tiles, no split needed
    (1, 3, 3, 4, 4, 3, 3, (1, 1), (0, 0)),         # one output pixel
    (3, 8, 8, 3, 6, 3, 3, (1, 1), (1, 1)),         # channel counts not multiples of 4: scalar gathers
    (32 * 196, 1, 1, 49, 64, 1, 1, (1, 1), (0, 0)),  # the stem as a 1x1 product over its im2col
    # large-M problems: the 128x128-tile configuration (four accumulators per wave)
    (32, 32, 32, 96, 96, 3, 3, (1, 1), (1, 1)),    # All-CNN-C conv2: 32768 rows
    (32, 16, 16, 96, 192, 3, 3, (2, 2), (1, 1)),   # strided, 96 -> 192
    (8, 16, 16, 192, 192, 3, 3, (1, 1), (0, 0)),   # no padding, ragged row tile (8*14*14 = 1568 rows: small config)
    (32, 16, 16, 192, 192, 3, 3, (1, 1), (1, 1)),  # All-CNN-C conv5: weight gradient on 96 x 128 tiles, columns flat over (tap, c)
    (32, 12, 12, 192, 100, 1, 1, (1, 1), (0, 0)),  # 1x1, 100 output channels (ragged column tile)
    (16, 16, 16, 256, 128, 1, 1, (1, 1), (0, 0)),  # ResNet-50-like 1x1 reduction
]


def _cl(t):
    return t.contiguous(memory_format=torch.channels_last)


def _reference(x, w, gy, stride, padding):
    x64, w64, gy64 = x.double().requires_grad_(True), w.double().requires_grad_(True), gy.double()
    y = torch.nn.functional.conv2d(x64, w64, None, stride, padding)
    gx, gw = torch.autograd.grad(y, (x64, w64), gy64)
    return y.detach(), gx, gw


@pytest.mark.parametrize("geom", GEOMS, ids=[str(g[:7]) for g in GEOMS])
def test_three_directions_match_float64(geom):
    n, h, w_, c, k, r, s, stride, padding = geom
    gen = torch.Generator(device=DEV).manual_seed(hash(geom) % 2**31)
    x = _cl(torch.randn(n, c, h, w_, device=DEV, generator=gen))
    w = _cl(torch.randn(k, c, r, s, device=DEV, generator=gen))
    oh = (h + 2 * padding[0] - r) // stride[0] + 1
    ow = (w_ + 2 * padding[1] - s) // stride[1] + 1
    gy = _cl(torch.randn(n, k, oh, ow, device=DEV, generator=gen))
    y64, gx64, gw64 = _reference(x, w, gy, stride, padding)

    def rel(a, b):
        return float((a.double() - b).abs().max() / b.abs().max().clamp_min(1e-30))

    for _ in range(2):  # the second round runs on recycled tickets / workspace
        y = _cl(torch.empty(n, k, oh, ow, device=DEV))
        _lib.conv2d_nhwc(0, y, x, w, n, h, w_, c, k, r, s, stride, padding)
        within(rel(y, y64), 2e-5)

        wT = w.permute(1, 2, 3, 0).contiguous()  # (I, H, W, O)
        gx = _cl(torch.empty(n, c, h, w_, device=DEV))
        _lib.conv2d_nhwc(1, gx, gy, wT, n, h, w_, c, k, r, s, stride, padding)
        within(rel(gx, gx64), 2e-5)

        gw = torch.zeros_like(w)
        _lib.conv2d_nhwc(2, gw, x, gy, n, h, w_, c, k, r, s, stride, padding)
        within(rel(gw, gw64), 2e-5)

        # data + weight gradient in ONE launch: the same numbers as the two separate launches
        if c % 4 == 0 and k % 4 == 0:  # (the merged launch has the 16-byte gather variant only)
            gx3, gw3 = torch.empty_like(gx), torch.zeros_like(w)
            _lib.conv2d_nhwc_backward(gx3, gw3, gy, x, wT, n, h, w_, c, k, r, s, stride, padding)
            assert rel(gx3, gx64) < 2e-5 and rel(gw3, gw64) < 2e-5

        # slab mode (what the curvature engine launches): the consumer sums the split-K slabs
        if c % 4 == 0 and k % 4 == 0:
            for d, act, mat, ref, shape in ((0, x, w, y64, (n, oh, ow, k)), (1, gy, wT, gx64, (n, h, w_, c)),
                                            (2, x, gy, gw64, (k, r, s, c))):
                sp = _lib.conv_plan(d, n, h, w_, c, k, r, s, stride, padding)
                numel = shape[0] * shape[1] * shape[2] * shape[3]
                slabs = torch.zeros((sp, numel), device=DEV)  # (zeros: direction 2 skips dead taps)
                _lib.conv2d_nhwc_slabs(d, slabs, act, mat, n, h, w_, c, k, r, s, stride, padding, sp)
                got = slabs.sum(0).view(shape).permute(0, 3, 1, 2)
                within(rel(got, ref), 2e-5, note=(d, sp))

        # bitwise repeatable
        y2, gx2, gw2 = torch.empty_like(y), torch.empty_like(gx), torch.zeros_like(w)
        _lib.conv2d_nhwc(0, y2, x, w, n, h, w_, c, k, r, s, stride, padding)
        _lib.conv2d_nhwc(1, gx2, gy, wT, n, h, w_, c, k, r, s, stride, padding)
        _lib.conv2d_nhwc(2, gw2, x, gy, n, h, w_, c, k, r, s, stride, padding)
        assert torch.equal(y, y2) and torch.equal(gx, gx2) and torch.equal(gw, gw2)


def test_channel_slice_of_a_wider_buffer():
    """``act_ld``: the gathered tensor is the first-C-channels slice of a wider NHWC buffer."""
    n, h, w_, c, k = 4, 6, 6, 16, 24
    gen = torch.Generator(device=DEV).manual_seed(3)
    wide = _cl(torch.randn(n, 2 * c, h, w_, device=DEV, generator=gen))
    x = wide[:, :c]
    w = _cl(torch.randn(k, c, 3, 3, device=DEV, generator=gen))
    y = _cl(torch.empty(n, k, h, w_, device=DEV))
    _lib.conv2d_nhwc(0, y, x, w, n, h, w_, c, k, 3, 3, (1, 1), (1, 1), act_ld=2 * c)
    want = torch.nn.functional.conv2d(x.double(), w.double(), None, 1, 1)
    within(float((y.double() - want).abs().max() / want.abs().max()), 2e-5)


def test_bad_geometry_is_refused():
    lib = _lib.load()
    x = torch.zeros(64, device=DEV)
    ws, tk = _lib.conv_scratch(x.device)
    p = _lib.c_void_p
    # kernel window larger than the padded input
    rc = lib.hf_conv2d_nhwc(0, p(x.data_ptr()), p(x.data_ptr()), p(x.data_ptr()), 1, 2, 2, 4, 4, 5, 5, 1, 1,
                            0, 0, 0, p(ws.data_ptr()), ws.numel() * 4, p(tk.data_ptr()), tk.numel(), 0, 0, None)
    assert rc == -1
    # float64 is not implemented by these kernels
    rc = lib.hf_conv2d_nhwc(0, p(x.data_ptr()), p(x.data_ptr()), p(x.data_ptr()), 1, 4, 4, 4, 4, 3, 3, 1, 1,
                            1, 1, 0, p(ws.data_ptr()), ws.numel() * 4, p(tk.data_ptr()), tk.numel(), 0, 1, None)
    assert rc == -1


def test_weight_slice_of_a_wider_buffer():
    """``mat_ld``: the weights are the first-C-channels slice of a wider [K][R][S][2C] buffer -- the W half
    of the tangent sweep's [W | v_W] operand, which the engine's own forward pass reads in place."""
    n, h, w_, c, k = 4, 6, 6, 16, 24
    gen = torch.Generator(device=DEV).manual_seed(5)
    x = _cl(torch.randn(n, c, h, w_, device=DEV, generator=gen))
    wcat = _cl(torch.randn(k, 2 * c, 3, 3, device=DEV, generator=gen))
    sp = _lib.conv_plan(0, n, h, w_, c, k, 3, 3, (1, 1), (1, 1))
    out = torch.empty((sp, n * h * w_ * k), device=DEV)
    _lib.conv2d_nhwc_slabs(0, out, x, wcat, n, h, w_, c, k, 3, 3, (1, 1), (1, 1), sp, mat_ld=2 * c)
    y = out.sum(0).view(n, h, w_, k).permute(0, 3, 1, 2)
    want = torch.nn.functional.conv2d(x.double(), wcat[:, :c].double(), None, 1, 1)
    within(float((y.double() - want).abs().max() / want.abs().max()), 2e-5)


def test_more_output_tiles_than_tickets_runs_unsplit():
    """A large batch of large maps: 9216 row tiles against 8192 ticket counters.  Such a launch needs no
    K split (and draws no tickets): it must run, not be refused -- and the patched layer's forward
    must work for it (own kernel or MIOpen, never an exception)."""
    n, h, w_, c, k = 64, 96, 96, 8, 8
    gen = torch.Generator(device=DEV).manual_seed(9)
    x = _cl(torch.randn(n, c, h, w_, device=DEV, generator=gen))
    w = _cl(torch.randn(k, c, 3, 3, device=DEV, generator=gen))
    y = _cl(torch.empty(n, k, h, w_, device=DEV))
    _lib.conv2d_nhwc(0, y, x, w, n, h, w_, c, k, 3, 3, (1, 1), (1, 1))
    want = torch.nn.functional.conv2d(x, w, None, 1, 1)
    within(float((y - want).abs().max() / want.abs().max()), 2e-5)
    from pytorchhessianfree_amd import modelprep

    conv = torch.nn.Conv2d(c, k, 3, 1, 1, bias=False).to(DEV)
    net = torch.nn.Sequential(conv)
    modelprep.fuse_conv_tangent(net, channels_last=True)
    got = net(x)
    ref = torch.nn.functional.conv2d(x.double(), conv.weight.detach().double(), None, 1, 1)
    within(float((got.double() - ref).abs().max() / ref.abs().max()), 2e-5)
    (g,) = torch.autograd.grad(got.sum(), conv.weight)
    assert torch.isfinite(g).all()


def test_convolution_launch_carrying_the_weight_scatter_equals_the_two_launches():
    """``hf_conv2d_nhwc_slabs_unpack``: a forward convolution over an im2col'd input (weight operand = a slice
    of the flat vector, as the stem's tangent convolution) whose launch carries ``hf_unpack_weights`` of other
    tensors as extra workgroups -- bitwise what ``hf_conv2d_nhwc_slabs`` + ``hf_unpack_weights`` write, for a
    vector (16-byte) and a scalar (C = 49) geometry, with a dead-tap mask on one scattered tensor; a geometry
    the merged launch does not take (128-wide tiles) is refused with HF_ERR_ARG, not run."""
    gen = torch.Generator(device=DEV).manual_seed(31)
    lib, P = _lib.load(), _lib.c_void_p
    for rows, c, k in ((6272, 49, 64), (1568, 52, 64)):
        cols = torch.randn(rows, c, device=DEV, generator=gen)  # [rows, 1, 1, c] NHWC
        k2, c2 = 24, 16
        n_vec = k * c + 5 + k2 * c2 * 9 + k2 * c2
        v = torch.randn(n_vec, device=DEV, generator=gen)
        vw = v[:k * c]  # the convolution's own weight operand: a slice of the vector
        off3, off1 = k * c + 5, k * c + 5 + k2 * c2 * 9
        sp = _lib.conv_plan(0, rows, 1, 1, c, k, 1, 1, (1, 1), (0, 0))

        def buffers():
            b3 = _cl(torch.full((k2, 2 * c2, 3, 3), 7.0, device=DEV))
            b1 = _cl(torch.full((k2, 2 * c2, 1, 1), 7.0, device=DEV))
            return [(off3, b3, c2, 0b000111010), (off1, b1, c2, 0)], torch.full((sp, rows * k), -1.0, device=DEV)

        slots_a, out_a = buffers()
        _lib.conv2d_nhwc_slabs(0, out_a, cols, vw, rows, 1, 1, c, k, 1, 1, (1, 1), (0, 0), sp)
        _lib.unpack_tangent(v, slots_a)
        slots_b, out_b = buffers()
        st = _lib.current_stream_ptr(v.device)
        rc = lib.hf_conv2d_nhwc_slabs_unpack(
            P(out_b.data_ptr()), P(cols.data_ptr()), P(vw.data_ptr()), rows, 1, 1, c, k, 1, 1, 1, 1, 0, 0, 0, 0, sp,
            out_b.shape[1], P(v.data_ptr()), *_lib.unpack_table(v, slots_b), _lib.HF_F32, st)
        assert rc == 0
        assert torch.equal(out_a, out_b)
        for (_, ba, _, _), (_, bb, _, _) in zip(slots_a, slots_b):
            assert torch.equal(ba, bb)
        got = out_b.sum(0).view(rows, k).double()
        want = cols.double() @ vw.view(k, c).double().t()
        within(float((got - want).abs().max() / want.abs().max()), 2e-5)
        assert not torch.equal(slots_b[0][1], torch.full_like(slots_b[0][1], 7.0))  # (the scatter really ran)
    # a large-map geometry runs the 128-wide configuration: the merged launch declines it
    n, h, c, k = 32, 32, 96, 96
    x = _cl(torch.randn(n, c, h, h, device=DEV, generator=gen))
    w = _cl(torch.randn(k, c, 3, 3, device=DEV, generator=gen))
    sp = _lib.conv_plan(0, n, h, h, c, k, 3, 3, (1, 1), (1, 1))
    out = torch.empty((sp, n * h * h * k), device=DEV)
    slots, _ = buffers()
    rc = lib.hf_conv2d_nhwc_slabs_unpack(
        P(out.data_ptr()), P(x.data_ptr()), P(w.data_ptr()), n, h, h, c, k, 3, 3, 1, 1, 1, 1, 0, 0, sp, out.shape[1],
        P(v.data_ptr()), *_lib.unpack_table(v, slots), _lib.HF_F32, _lib.current_stream_ptr(v.device))
    assert rc == _lib.HF_ERR_ARG


@pytest.mark.parametrize("geom", [
    (32, 7, 7, 64, 64, 3, 3, (1, 1), (1, 1)),     # 1568 rows: ragged last row tile
    (32, 7, 7, 64, 128, 3, 3, (2, 2), (1, 1)),    # strided, two column tiles
    (8, 5, 5, 32, 96, 1, 1, (1, 1), (0, 0)),      # 96 outputs: second column tile half full; 200 rows
    (4, 3, 3, 128, 256, 3, 3, (1, 1), (1, 1)),    # 36 rows: one ragged row tile, four column tiles
], ids=lambda g: str(g[:7]))
def test_tangent_convolution_with_batchnorm_partial_sums_in_its_epilogue(geom):
    """``hf_conv2d_nhwc_group_slabs_bnsum``: the slabs are bitwise those of the plain slab launch; the partial rows,
    added up, are the per-channel sums of the convolution output and of output * xhat (float64 reference; tolerance:
    the slab sums' fp32 rounding, 2e-6 of the column's sum of magnitudes); a second launch writes the same bits; one
    launch for two problems, one with sums and one without, equals the single launches."""
    n, h, w_, c, k, r, s, stride, padding = geom
    gen = torch.Generator(device=DEV).manual_seed(7 + c + k)
    x = _cl(torch.randn(n, c, h, w_, device=DEV, generator=gen))
    wt = _cl(torch.randn(k, c, r, s, device=DEV, generator=gen) / (c * r * s) ** 0.5)
    geo = (n, h, w_, c, k, r, s, stride, padding)
    splits = _lib.conv_plan(0, n, h, w_, c, k, r, s, stride, padding)
    oh, ow = (h + 2 * padding[0] - r) // stride[0] + 1, (w_ + 2 * padding[1] - s) // stride[1] + 1
    rows = n * oh * ow
    plain = torch.empty((splits, rows * k), device=DEV)
    _lib.conv2d_nhwc_slabs(0, plain, x, wt, n, h, w_, c, k, r, s, stride, padding, splits)
    a = torch.randn(rows, k, device=DEV, generator=gen)          # the layer's recorded convolution output
    mean = a.mean(0).contiguous()
    rstd = (1.0 / (a.var(0, unbiased=False) + 1e-5).sqrt()).contiguous()
    nparts = -(-rows // 64) * splits
    out = torch.empty_like(plain)
    p1, px = (torch.full((nparts, k), float("nan"), device=DEV) for _ in range(2))
    probs = [(0, out, x, wt, geo, splits, 0, 0)]
    assert _lib.conv_group_slabs_bnsum(probs, [(a, mean, rstd, px, p1)], DEV)
    assert torch.equal(out, plain)
    t64 = plain.double().sum(0).view(rows, k)
    xhat = ((a - mean) * rstd).double()
    for got, ref, mag in ((p1, t64.sum(0), t64.abs().sum(0)), (px, (t64 * xhat).sum(0), (t64 * xhat).abs().sum(0))):
        assert torch.isfinite(got).all()
        within(float(((got.double().sum(0) - ref).abs() / mag.clamp_min(1e-30)).max()), 2e-6)
    p1b, pxb = torch.empty_like(p1), torch.empty_like(px)
    assert _lib.conv_group_slabs_bnsum(probs, [(a, mean, rstd, pxb, p1b)], DEV)
    assert torch.equal(p1b, p1) and torch.equal(pxb, px)
    # two problems in one launch: the first without sums
    out2, out3 = torch.empty_like(plain), torch.empty_like(plain)
    p1c, pxc = torch.empty_like(p1), torch.empty_like(px)
    assert _lib.conv_group_slabs_bnsum([(0, out2, x, wt, geo, splits, 0, 0), (0, out3, x, wt, geo, splits, 0, 0)],
                                       [None, (a, mean, rstd, pxc, p1c)], DEV)
    assert torch.equal(out2, plain) and torch.equal(out3, plain)
    assert torch.equal(p1c, p1) and torch.equal(pxc, px)
    # a wrong number of partial rows is refused (the caller then takes the two-launch path)
    assert not _lib.conv_group_slabs_bnsum(probs, [(a, mean, rstd, px[:-1], p1[:-1])], DEV)


def test_weight_gradient_of_a_zero_padded_operand_on_the_flat_96_row_tiles():
    """a weight gradient whose X operand is zero-padded to ``cs`` channels (``out_c < cs``: the engine's
    im2col'd stem pads its rows to 16-byte multiples) must not take the Flat96 configuration's flat (tap, channel)
    column enumeration as it is -- the padded columns of one row landed in the next row's first entries and the last
    row wrote past the slab.  kout = 96, cs = 244, out_c = 243: with ONE tap (the im2col form) the flat tiles run and
    cut the padded column off (result against float64, the float behind the slab untouched); with 3 x 3 taps the
    launch is refused (HF_ERR_ARG), never silently wrong."""
    gen = torch.Generator(device=DEV).manual_seed(17)
    rows, cs, out_c, k = 16384, 244, 243, 96
    x = torch.zeros(rows, cs, device=DEV)
    x[:, :out_c] = torch.randn(rows, out_c, device=DEV, generator=gen)
    gy = torch.randn(rows, k, device=DEV, generator=gen)
    sp = _lib.conv_plan(2, rows, 1, 1, cs, k, 1, 1, (1, 1), (0, 0))
    guard = 7.0
    buf = torch.full((sp * k * out_c + 64,), guard, device=DEV)
    out = buf[: sp * k * out_c].view(sp, k * out_c)
    _lib.conv2d_nhwc_slabs(2, out, x, gy, rows, 1, 1, cs, k, 1, 1, (1, 1), (0, 0), sp, out_c=out_c)
    got = out.sum(0).view(k, out_c)
    want = gy.double().t() @ x[:, :out_c].double()
    within(float((got.double() - want).abs().max() / want.abs().max()), 2e-5)
    assert bool((buf[sp * k * out_c:] == guard).all())  # nothing written behind the last row
    out2 = torch.empty_like(out)
    _lib.conv2d_nhwc_slabs(2, out2, x, gy, rows, 1, 1, cs, k, 1, 1, (1, 1), (0, 0), sp, out_c=out_c)
    assert torch.equal(out, out2)
    # several taps: 32 x 16 x 16 maps of 244 channels, 3 x 3 -- Flat96 territory (96 rows, every tap live)
    n, h, w_ = 32, 16, 16
    x4 = _cl(torch.zeros(n, cs, h, w_, device=DEV))
    gy4 = _cl(torch.randn(n, k, h, w_, device=DEV, generator=gen))
    sp4 = _lib.conv_plan(2, n, h, w_, cs, k, 3, 3, (1, 1), (1, 1))
    out4 = torch.empty((sp4, k * 9 * out_c), device=DEV)
    p = _lib.c_void_p
    rc = _lib.load().hf_conv2d_nhwc_slabs(2, p(out4.data_ptr()), p(x4.data_ptr()), p(gy4.data_ptr()), n, h, w_, cs, k, 3, 3,
                                          1, 1, 1, 1, 0, 0, out_c, sp4, out4.shape[1], _lib.HF_F32,
                                          _lib.current_stream_ptr(out4.device))
    if rc != _lib.HF_ERR_ARG:  # (another tile configuration was planned for this shape: then it must be RIGHT)
        _lib.check(rc, "hf_conv2d_nhwc_slabs")
        x4[:, :out_c] = torch.randn(n, out_c, h, w_, device=DEV, generator=gen)
        _lib.conv2d_nhwc_slabs(2, out4, x4, gy4, n, h, w_, cs, k, 3, 3, (1, 1), (1, 1), sp4, out_c=out_c)
        w0 = torch.zeros(k, out_c, 3, 3, device=DEV, dtype=torch.float64, requires_grad=True)
        y = torch.nn.functional.conv2d(x4[:, :out_c].double(), w0, None, 1, 1)
        (gw,) = torch.autograd.grad(y, w0, gy4.double())
        got4 = out4.sum(0).view(k, 3, 3, out_c).permute(0, 3, 1, 2)
        within(float((got4.double() - gw).abs().max() / gw.abs().max()), 2e-5)
