"""Feasibility probe (round 2): how fast are library fp32 GEMMs at the shapes an explicit
im2col formulation of the ResNet-18 (28x28, batch 32) curvature product needs?
T: [M, 2K] x [2K, N] (tangent), D: [M, N] x [N, K] (data gradient), W: [N, M] x [M, K]
(weight gradient).  Each op is timed as a hipGraph of 20 back-to-back launches."""
import sys, time, json
import torch

dev = "cuda"
shapes = [("conv1", 6272, 49, 64), ("l1", 1568, 576, 64), ("l2a", 512, 576, 128), ("l2", 512, 1152, 128),
          ("l3a", 128, 1152, 256), ("l3", 128, 2304, 256), ("l4a", 32, 1024, 512), ("l4", 32, 512, 512)]
if len(sys.argv) > 1:
    torch.backends.cuda.preferred_blas_library(sys.argv[1])
REP = 20


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(REP):
            fn()
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (10 * REP) * 1e6


for name, M, K, N in shapes:
    A2 = torch.randn(M, 2 * K, device=dev); B2 = torch.randn(N, 2 * K, device=dev); C = torch.empty(M, N, device=dev)
    gy = torch.randn(M, N, device=dev); Wm = torch.randn(N, K, device=dev); Gc = torch.empty(M, K, device=dev)
    cols = torch.randn(M, K, device=dev); gW = torch.empty(N, K, device=dev)
    t_T = timed(lambda: torch.mm(A2, B2.t(), out=C))
    t_D = timed(lambda: torch.mm(gy, Wm, out=Gc))
    t_W = timed(lambda: torch.mm(gy.t(), cols, out=gW))
    fl = 2.0 * M * K * N
    print(json.dumps({"layer": name, "M": M, "K": K, "N": N, "T_us": round(t_T, 2), "D_us": round(t_D, 2),
                      "W_us": round(t_W, 2), "T_TF": round(2 * fl / t_T / 1e6, 1), "D_TF": round(fl / t_D / 1e6, 1),
                      "W_TF": round(fl / t_W / 1e6, 1)}), flush=True)
