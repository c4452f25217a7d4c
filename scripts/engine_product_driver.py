"""One-product driver for profiling the fused curvature engine's kernels (rocprofv3 --pmc / --kernel-trace;
run the program itself after ``--``:  rocprofv3 --pmc FETCH_SIZE -d <dir> -- python3 scripts/engine_product_driver.py).

Builds the ResNet-18 workload's engine (BASELINE.json configs[1]: batch 32, 1x28x28), issues
``--products`` GGN products as plain eager launches (no hipGraph: every dispatch carries its kernel name)
and writes, to ``--out``, the ALGORITHMIC bytes / flops of every launch of one product in launch order
(operands counted once per launch: each input tensor once, split-K slabs once each, dead kernel taps
not at all) -- scripts/pmc_engine_table.py joins that list with the counter CSVs by dispatch order."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.getcwd())
import torch

import pytorchhessianfree_amd as hf
from pytorchhessianfree_amd import _lib, curvature, modelprep
from pytorchhessianfree_amd import testproblems as tp


def live_taps(h, w, r, s, sh, sw, ph, pw):
    oh, ow = (h + 2 * ph - r) // sh + 1, (w + 2 * pw - s) // sw + 1
    rows = sum(any(0 <= o * sh - ph + i < h for o in range(oh)) for i in range(r))
    cols = sum(any(0 <= o * sw - pw + j < w for o in range(ow)) for j in range(s))
    return rows * cols, oh, ow


def conv_cost(direction, n, h, w, c, k, r, s, sh, sw, ph, pw, splits, out_c=0):
    taps, oh, ow = live_taps(h, w, r, s, sh, sw, ph, pw)
    rows = n * oh * ow
    flops = 2.0 * rows * k * c * taps
    if direction == 0:
        return 4 * (n * h * w * c + k * taps * c), 4 * splits * rows * k, flops
    if direction == 1:
        return 4 * (rows * k + c * taps * k), 4 * splits * n * h * w * c, flops
    return 4 * (n * h * w * c + rows * k), 4 * splits * k * taps * (out_c or c), flops


class Recorder:
    """Wraps the ctypes library: logs (entry point, kernel name, algorithmic read / written bytes, flops)."""

    def __init__(self, lib):
        self._lib, self.log, self.on = lib, [], False

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        cost = getattr(self, "_cost_" + name, None)
        if cost is None:
            return fn

        def call(*a):
            if self.on:
                for rec in cost(*a):
                    self.log.append(dict(zip(("entry", "kernel", "read", "written", "flops"), (name,) + rec)))
            return fn(*a)

        return call

    @staticmethod
    def _cost_hf_conv2d_nhwc_slabs(direction, out, act, mat, n, h, w, c, k, r, s, sh, sw, ph, pw, act_ld, mat_ld,
                                   out_c, splits, slab, dtype, stream):
        rd, wr, fl = conv_cost(direction, n, h, w, c, k, r, s, sh, sw, ph, pw, splits, out_c)
        scalar = (c % 4) or (k % 4)
        return [(("k_conv_tn" if direction == 2 else "k_conv_nt") + ("<true>" if scalar else "<false>"), rd, wr, fl)]

    @staticmethod
    def _cost_hf_conv2d_nhwc_slabs_unpack(out, act, mat, n, h, w, c, k, r, s, sh, sw, ph, pw, act_ld, mat_ld, splits,
                                          slab, usrc, dsts, offs, numels, slabs, inners, live, halves, nt, dtype, stream):
        rd, wr, fl = conv_cost(0, n, h, w, c, k, r, s, sh, sw, ph, pw, splits)
        tot = 0
        for t in range(nt):
            hwc = slabs[t] // inners[t] if inners[t] else 1
            frac = bin(live[t]).count("1") / hwc if (live and live[t]) else 1.0
            tot += int(4 * numels[t] * frac)
        return [("k_conv_nt_unpack", rd + tot, wr + tot, fl)]  # the stem's convolution + the v_W scatter it carries

    @staticmethod
    def _cost_hf_conv2d_nhwc_backward_slabs(dx, dw, dy, x, wt, n, h, w, c, k, r, s, sh, sw, ph, pw, sd, ld, sw_, lw,
                                            dtype, stream):
        r1, w1, f1 = conv_cost(1, n, h, w, c, k, r, s, sh, sw, ph, pw, sd)
        r2, w2, f2 = conv_cost(2, n, h, w, c, k, r, s, sh, sw, ph, pw, sw_)
        taps, oh, ow = live_taps(h, w, r, s, sh, sw, ph, pw)
        return [("k_conv_dw", r1 + r2 - 4 * n * oh * ow * k, w1 + w2, f1 + f2)]  # dY counted once

    @staticmethod
    def _cost_hf_conv2d_nhwc_dw_slabs(d, w, dtype, stream):
        rd = wr = fl = 0
        for ptr in (d, w):
            q = _lib.ctypes.cast(ptr, _lib.ctypes.POINTER(_lib.ConvProblem)).contents
            a, b, c_ = conv_cost(q.direction, q.n, q.h, q.w, q.c, q.k, q.r, q.s, q.stride_h, q.stride_w, q.pad_h,
                                 q.pad_w, q.splits, q.out_c)
            rd, wr, fl = rd + a, wr + b, fl + c_
        return [("k_conv_dw", rd, wr, fl)]

    @staticmethod
    def _cost_hf_conv2d_nhwc_group_slabs(problems, count, dtype, stream):
        arr = _lib.ctypes.cast(problems, _lib.ctypes.POINTER(_lib.ConvProblem * count)).contents
        rd = wr = fl = 0
        for q in arr:
            a, b, c_ = conv_cost(q.direction, q.n, q.h, q.w, q.c, q.k, q.r, q.s, q.stride_h, q.stride_w, q.pad_h,
                                 q.pad_w, q.splits, q.out_c)
            rd, wr, fl = rd + a, wr + b, fl + c_
        return [("k_conv_group", rd, wr, fl)]

    @staticmethod
    def _cost_hf_chan_affine_ex(out, a, x, mean, rstd, w, q, r, add, mask, relu, n, c, hw, cl, old, ald, splits, slab,
                                dtype, stream):
        tot = 4 * n * c * hw
        rd = tot * ((splits if a else 0) + (1 if q else 0) + (1 if add else 0) + (1 if mask else 0))
        return [("k_chan_affine<", rd, tot, 6.0 * n * c * hw)]

    @staticmethod
    def _cost_hf_chan_affine_pair(problems, dtype, stream):
        arr = _lib.ctypes.cast(problems, _lib.ctypes.POINTER(_lib.AffineProblem * 2)).contents
        rd = wr = 0
        for q in arr:
            tot = 4 * q.n * q.c * q.hw
            rd += tot * ((q.a_splits if q.a else 0) + (1 if q.q else 0) + (1 if q.add else 0) + (1 if q.mask_src else 0))
            wr += tot
        return [("k_chan_affine_pair", rd, wr, 0.0)]

    @staticmethod
    def _cost_hf_chan_affine_bwd_ex(gx, gw, gb, gres, gy, s1, l1, gy2, s2, l2, x, mean, rstd, w, mask, n, c, hw, cl, rb,
                                    dtype, stream):
        tot = 4 * n * c * hw
        rd = tot * (s1 + (s2 if gy2 else 0) + (1 if x else 0) + (1 if mask else 0))
        wr = tot * ((1 if gx else 0) + (1 if gres else 0))
        name = "k_bn_adjoint_rows(" if rb > 1 else ("k_chan_affine_bwd_nhwc" if hw > 1 else "k_chan_affine_bwd<")
        return [(name, rd, wr, 8.0 * n * c * hw)]

    @staticmethod
    def _cost_hf_chan_affine_train(out, a, x, mean, rstd, w, px, p1, nparts, vq, vr, count, add, mask, n, c, hw, old,
                                   ald, splits, slab, dtype, stream):
        tot = 4 * n * c * hw
        rd = tot * (splits + 1 + (1 if add else 0) + (1 if mask else 0)) + 4 * 2 * nparts * c
        return [("k_chan_affine_v4_train", rd, tot, 8.0 * n * c * hw)]

    @staticmethod
    def _cost_hf_chan_affine_train_pair(problems, dtype, stream):
        arr = _lib.ctypes.cast(problems, _lib.ctypes.POINTER(_lib.AffineTrainProblem * 2)).contents
        rd = wr = 0
        for q in arr:
            tot = 4 * q.n * q.c * q.hw
            rd += tot * (q.a_splits + 1 + (1 if q.mask_src else 0)) + 4 * 2 * q.nparts * q.c
            wr += tot
        return [("k_chan_affine_v4_train_pair", rd, wr, 0.0)]

    @staticmethod
    def _cost_hf_conv2d_nhwc_group_slabs_bnsum(problems, count, sums, dtype, stream):
        arr = _lib.ctypes.cast(problems, _lib.ctypes.POINTER(_lib.ConvProblem * count)).contents
        bn = _lib.ctypes.cast(sums, _lib.ctypes.POINTER(_lib.ConvBnSum * count)).contents
        rd = wr = fl = 0
        for q, b in zip(arr, bn):
            a, b_, c_ = conv_cost(q.direction, q.n, q.h, q.w, q.c, q.k, q.r, q.s, q.stride_h, q.stride_w, q.pad_h,
                                  q.pad_w, q.splits, q.out_c)
            rd, wr, fl = rd + a, wr + b_, fl + c_
            if b.part_1:  # the tile's xhat operand once per split; two partial rows per (row tile, split)
                oh = (q.h + 2 * q.pad_h - q.r) // q.stride_h + 1
                ow = (q.w + 2 * q.pad_w - q.s) // q.stride_w + 1
                rd += 4 * q.n * oh * ow * q.k * q.splits
                wr += 4 * 2 * b.part_rows * q.k
        return [("k_conv_group" if count > 1 else "k_conv_nt<false", rd, wr, fl)]

    @staticmethod
    def _cost_hf_chan_affine_bwd_pair(problems, dtype, stream):
        arr = _lib.ctypes.cast(problems, _lib.ctypes.POINTER(_lib.BnAdjointProblem * 2)).contents
        rd = wr = 0
        for q in arr:
            tot = 4 * q.n * q.c * q.hw
            rd += tot * (q.gy_splits + (q.gy2_splits if q.gy2 else 0) + (1 if q.x else 0) + (1 if q.mask_src else 0))
            wr += tot * ((1 if q.gx else 0) + (1 if q.gres else 0))
        return [("k_bn_adjoint_rows_pair", rd, wr, 0.0)]

    @staticmethod
    def _cost_hf_unpack_weights(src, dsts, offs, numels, slabs, inners, live, halves, nt, dtype, stream):
        tot = 0
        for t in range(nt):
            hwc = slabs[t] // inners[t] if inners[t] else 1
            frac = bin(live[t]).count("1") / hwc if (live and live[t]) else 1.0
            tot += int(4 * numels[t] * frac)
        return [("k_unpack_tangent", tot, tot, 0.0)]

    @staticmethod
    def _cost_hf_pack_ex(dst, srcs, numels, perm, splits, live, nt, scale, mode, dtype, stream):
        rd = wr = 0
        for t in range(nt):
            hwc = perm[2 * t + 1] if perm and perm[2 * t] else 1
            frac = bin(live[t]).count("1") / hwc if (live and live[t]) else 1.0
            rd += int(4 * numels[t] * frac * (splits[2 * t] if splits else 1))
            wr += 4 * numels[t]
        return [("k_pack", rd, wr, 0.0)]

    @staticmethod
    def _cost_hf_maxpool_tangent_nhwc(out, t, idx, n, h, w, oh, ow, c, ld, dtype, stream):
        tot = 4 * n * oh * ow * c
        return [("k_maxpool_tangent", 3 * tot, tot, 0.0)]  # idx + the gathered element (a 32-B sector at least)

    @staticmethod
    def _cost_hf_maxpool_adjoint_nhwc(g, a, sa, la, b, sb, lb, idx, n, h, w, oh, ow, c, kh, kw, sh, sw, ph, pw, dtype,
                                      stream):
        po = 4 * n * oh * ow * c
        return [("k_maxpool_adjoint", po * (1 + sa + (sb if b else 0)), 4 * n * h * w * c, 0.0)]

    @staticmethod
    def _cost_hf_pool_ce_head(g, jv, t, p, scale, n, hw, k, dtype, stream):
        tot = 4 * n * hw * k
        return [("k_pool_ce_head", tot + 4 * n * k, tot, 0.0)]

    @staticmethod
    def _cost_hf_softmax_ce_hvp(out, p, v, scale, rows, cols, dtype, stream):
        return [("k_softmax_ce_hvp", 2 * 4 * rows * cols, 4 * rows * cols, 0.0)]

    @staticmethod
    def _cost_hf_linear_ce_head(gf, gw, gb, tf, f, w, vw, vb, p, scale, rows, feat, classes, dtype, stream):
        groups = (rows + 3) // 4
        rd = 4 * (2 * rows * feat + 2 * classes * feat * groups + rows * classes)
        wr = 4 * (rows * feat + groups * classes * (feat + 1))
        return [("k_linear_ce_head", rd, wr, 2.0 * rows * feat * classes * 4)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--products", type=int, default=20)
    ap.add_argument("--out", default="")
    ap.add_argument("--workload", default="resnet18", choices=["resnet18", "allcnnc", "resnet50"])
    ap.add_argument("--curvature", default="ggn", choices=["ggn", "hessian"])
    ap.add_argument("--bn", default="eval", choices=["eval", "train"])
    ap.add_argument("--freeze", default="", choices=["", "stem+layer1"])
    args = ap.parse_args()
    hf.configure()
    dev = "cuda"
    if args.workload == "resnet18":
        model, (x, t), lossf = tp.resnet18_mnist(batch_size=32, device=dev, data_seed=tp.RESNET18_B32_SEPARATED_SEEDS[0])
    elif args.workload == "allcnnc":
        model, (x, t), lossf = tp.allcnnc_cifar100(batch_size=32, device=dev)
    else:
        model, (x, t), lossf = tp.resnet50_small_images(batch_size=32, device=dev)
    if args.bn == "train":
        model.train()
    if args.freeze:
        tp.freeze_stem_and_layer1(model)
    modelprep.prepare_model(model, channels_last=True)
    params = [p for p in model.parameters() if p.requires_grad]
    out = model(x)
    if args.curvature == "hessian":
        eng = curvature.hessian_operator(lossf(out, t), out, params)
    else:
        eng = curvature.ggn_operator(lossf(out, t), out, params)
    assert "engine" in eng.mode, "the engine did not take this model"
    v = torch.randn(eng.n, device=dev)
    res = torch.empty(eng.n, device=dev)
    for _ in range(3):
        eng.local(v, out=res)
    torch.cuda.synchronize()
    rec = Recorder(_lib.load())
    _lib._lib = rec  # every later _lib.load() returns the recorder
    rec.on = True
    eng.local(v, out=res)
    rec.on = False
    torch.cuda.synchronize()
    for _ in range(args.products - 1):
        eng.local(v, out=res)
    torch.cuda.synchronize()
    if args.out:
        os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
        with open(args.out, "w") as f:
            json.dump({"products": args.products, "launches_per_product": len(rec.log), "launches": rec.log}, f, indent=1)
    print(f"{len(rec.log)} launches per product, {args.products} products", file=sys.stderr)


if __name__ == "__main__":
    main()
