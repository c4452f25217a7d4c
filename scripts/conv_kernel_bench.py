"""Micro-benchmark: the package's implicit-GEMM convolution kernels (hf_conv2d_nhwc) against
PyTorch-ROCm -> MIOpen on the convolution shapes of one ResNet-18 (28x28, batch 32) GGN
product.  Each op is timed as a hipGraph of 20 back-to-back launches (what a product replay
does), per direction: T = tangent (2*Cin channels), D = data gradient, W = weight gradient.

    python scripts/conv_kernel_bench.py [--big 1] [--no-reduce 1]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pytorchhessianfree_amd as hf  # noqa: E402
from pytorchhessianfree_amd import _lib  # noqa: E402

hf.configure()
DEV = "cuda"
REP = 20
SHAPES = [("layer1", 32, 7, 7, 64, 64, 3, 1, 1), ("layer2.0", 32, 7, 7, 64, 128, 3, 2, 1),
          ("layer2.ds", 32, 7, 7, 64, 128, 1, 2, 0), ("layer2", 32, 4, 4, 128, 128, 3, 1, 1),
          ("layer3.0", 32, 4, 4, 128, 256, 3, 2, 1), ("layer3", 32, 2, 2, 256, 256, 3, 1, 1),
          ("layer4.0", 32, 2, 2, 256, 512, 3, 2, 1), ("layer4", 32, 1, 1, 512, 512, 3, 1, 1)]
# large-M problems (All-CNN-C at batch 32; ResNet-50 on 64x64 images): the 128x128-tile configuration
BIG_SHAPES = [("allcnnc.2", 32, 32, 32, 96, 96, 3, 1, 1), ("allcnnc.3", 32, 32, 32, 96, 96, 3, 2, 1),
              ("allcnnc.4", 32, 16, 16, 96, 192, 3, 1, 1), ("allcnnc.5", 32, 16, 16, 192, 192, 3, 1, 1),
              ("allcnnc.6", 32, 16, 16, 192, 192, 3, 2, 1), ("allcnnc.7", 32, 8, 8, 192, 192, 3, 1, 0),
              ("allcnnc.8", 32, 6, 6, 192, 192, 1, 1, 0), ("r50.l1.conv2", 32, 16, 16, 64, 64, 3, 1, 1),
              ("r50.l1.conv3", 32, 16, 16, 64, 256, 1, 1, 0), ("r50.l2.conv2", 32, 8, 8, 128, 128, 3, 1, 1),
              ("r50.l2.conv3", 32, 8, 8, 128, 512, 1, 1, 0), ("r50.l3.conv2", 32, 4, 4, 256, 256, 3, 1, 1)]


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(REP):
            fn()
    g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (10 * REP) * 1e6


def cl(t):
    return t.contiguous(memory_format=torch.channels_last)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--no-reduce", type=int, default=0,
                    help="1: also time the slab-mode launches (hf_conv2d_nhwc_slabs: split-K partial results left "
                         "for the consumer kernel's prologue, as the fused curvature engine runs them)")
    ap.add_argument("--big", type=int, default=0, help="1: the large-M shapes (All-CNN-C, ResNet-50) instead")
    args = ap.parse_args()
    shapes = BIG_SHAPES if args.big else SHAPES
    torch.backends.cudnn.benchmark = False  # immediate mode on the shipped find-db, as the product runs
    for name, n, h, w, c, k, r, st, pd in shapes:
        stride, pad = (st, st), (pd, pd)
        oh = (h + 2 * pd - r) // st + 1
        x2, w2 = cl(torch.randn(n, 2 * c, h, w, device=DEV)), cl(torch.randn(k, 2 * c, r, r, device=DEV))
        x, wt = cl(torch.randn(n, c, h, w, device=DEV)), cl(torch.randn(k, c, r, r, device=DEV))
        gy = cl(torch.randn(n, k, oh, oh, device=DEV))
        wT = wt.permute(1, 2, 3, 0).contiguous()
        y = cl(torch.empty(n, k, oh, oh, device=DEV))
        gx = cl(torch.empty(n, c, h, w, device=DEV))
        gw = torch.zeros_like(wt)
        line = {"layer": name, "M": n * oh * oh, "Cin": c, "Cout": k}
        line["T_own"] = timed(lambda: _lib.conv2d_nhwc(0, y, x2, w2, n, h, w, 2 * c, k, r, r, stride, pad))
        line["D_own"] = timed(lambda: _lib.conv2d_nhwc(1, gx, gy, wT, n, h, w, c, k, r, r, stride, pad))
        line["W_own"] = timed(lambda: _lib.conv2d_nhwc(2, gw, x, gy, n, h, w, c, k, r, r, stride, pad))
        line["DW_own"] = timed(lambda: _lib.conv2d_nhwc_backward(gx, gw, gy, x, wT, n, h, w, c, k, r, r, stride, pad))
        if args.no_reduce:
            sT = _lib.conv_plan(0, n, h, w, 2 * c, k, r, r, stride, pad)
            sD = _lib.conv_plan(1, n, h, w, c, k, r, r, stride, pad)
            sW = _lib.conv_plan(2, n, h, w, c, k, r, r, stride, pad)
            ys = torch.empty(sT, y.numel(), device=DEV)
            gxs = torch.empty(sD, gx.numel(), device=DEV)
            gws = torch.zeros(sW, gw.numel(), device=DEV)
            line["splits"] = [sT, sD, sW]
            line["T_slab"] = timed(lambda: _lib.conv2d_nhwc_slabs(0, ys, x2, w2, n, h, w, 2 * c, k, r, r, stride, pad, sT))
            line["D_slab"] = timed(lambda: _lib.conv2d_nhwc_slabs(1, gxs, gy, wT, n, h, w, c, k, r, r, stride, pad, sD))
            line["W_slab"] = timed(lambda: _lib.conv2d_nhwc_slabs(2, gws, x, gy, n, h, w, c, k, r, r, stride, pad, sW))
        line["T_miopen"] = timed(lambda: torch.nn.functional.conv2d(x2, w2, None, stride, pad))
        line["D_miopen"] = timed(lambda: torch.ops.aten.convolution_backward(
            gy, x, wt, None, stride, pad, [1, 1], False, [0, 0], 1, [True, False, False]))
        line["W_miopen"] = timed(lambda: torch.ops.aten.convolution_backward(
            gy, x, wt, None, stride, pad, [1, 1], False, [0, 0], 1, [False, True, False]))
        taps = r * r
        gf = 2.0 * n * oh * oh * k * c * taps * 1e-9  # GFLOP of D and of W; T is twice that (2*Cin channels)
        line["GFLOP_DW_each"] = gf
        for key, mul in (("T", 2.0), ("D", 1.0), ("W", 1.0)):
            for impl in ("own", "slab", "miopen"):
                if f"{key}_{impl}" in line:
                    line[f"{key}_{impl}_TF"] = mul * gf / line[f"{key}_{impl}"] * 1e3  # GFLOP / us = PFLOP/s
        print(json.dumps({k_: (round(v, 2) if isinstance(v, float) else v) for k_, v in line.items()}), flush=True)


if __name__ == "__main__":
    main()
