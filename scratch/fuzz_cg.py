import sys, os, warnings, random
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import pytorchhessianfree_amd as hf
from oracle import pcg as oracle
random.seed(0); bad = 0; N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for case in range(N):
    n = random.choice([1, 2, 3, 4, 5, 7, 63, 64, 65, 255, 256, 257, 511, 1000, 1023, 1025, 4097, 20011])
    dtype = random.choice([torch.float32, torch.float64])
    mode = random.choice(["none", "diag", "ext"])
    warm = random.random() < 0.5; martens = random.random() < 0.6
    max_iter = random.choice([1, 2, 5, 17, 60])
    store = random.choice([[], [0], None, list(range(0, 61, 3)), [0, 1, 2, 1000]])
    lam = random.choice([0.0, 0.3, 2.0]); tol = random.choice([0.0, 1e-5, 1e-2]); atol = random.choice([None, 1e-6])
    g = torch.Generator().manual_seed(case)
    d = (torch.rand(n, generator=g) * 10 + 0.05).to(dtype); b = torch.randn(n, generator=g).to(dtype)
    x0 = torch.randn(n, generator=g).to(dtype) if warm else None
    diag = torch.rand(n, generator=g).to(dtype)
    dd = d.cuda()
    def Bc(v): return (d.double() * v.double()).to(dtype)
    def Bg(v): return (dd.double() * v.double()).to(dtype)
    Mg = hf.DiagonalPreconditioner(diag.cuda(), lam if lam else 0.1) if mode != "none" else None
    minv = Mg.minv.cpu() if Mg is not None else None
    Mc = (lambda v: minv * v) if mode != "none" else None
    Mgg = Mg if mode == "diag" else ((lambda v: Mg.minv * v) if mode == "ext" else None)
    kw = dict(max_iter=max_iter, tol=tol, atol=atol, martens_conv_crit=martens, store_x_at_iters=store)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ox, om, orr = oracle.pcg(lambda v: Bc(v) + lam * v, b, x0=x0, M=Mc, accumulate="fp64", **kw)
        gx, gm, grr = hf.cg(hf.DampedCurvature(Bg, lam), b.cuda(), x0=None if x0 is None else x0.cuda(), M=Mgg, **kw)
    ok = (orr == grr) and len(ox) == len(gx)
    tolx = 3e-5 if dtype == torch.float32 else 1e-8
    if ok:
        for a, o in zip(gx, ox):
            if (a is None) != (o is None): ok = False; break
            if a is not None:
                e = float((a.cpu() - o).abs().max() / max(float(o.abs().max()), 1e-30))
                if not (e < tolx or not np.isfinite(float(o.abs().max()))): ok = False; break
        if ok and martens:
            gmv = np.array([float(m) for m in gm]); omv = np.array([float(m) for m in om])
            ok = len(gmv) == len(omv) and np.allclose(gmv, omv, rtol=1e-4 if dtype == torch.float32 else 1e-8, atol=1e-6, equal_nan=True)
    if not ok:
        bad += 1
        print("MISMATCH case", case, dict(n=n, dtype=str(dtype), mode=mode, warm=warm, martens=martens, max_iter=max_iter, store=store if store is None or len(store) < 6 else "many", lam=lam, tol=tol, atol=atol), orr, grr, len(ox), len(gx))
print("cases", N, "mismatches", bad)
