import sys, subprocess, os, gc
sys.path.insert(0, os.getcwd())
if len(sys.argv) > 1:
    import torch
    from pytorchhessianfree_amd import curvature, testproblems as tp
    variant = sys.argv[1]
    m, (x, t), lf = tp.resnet18_mnist(batch_size=8, device="cuda")
    ps = [p for p in m.parameters()]
    def builder():
        o = m(x); return curvature.GGNOperator(lf(o, t), o, ps)
    v = torch.randn(sum(p.numel() for p in ps), device="cuda")
    if variant == "A":
        e = builder(); r = e(v); torch.cuda.synchronize(); del e, r; gc.collect(); torch.cuda.empty_cache(); torch.cuda.synchronize()
    elif variant == "B":
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            e = builder(); r = e(v)
        torch.cuda.synchronize()
    elif variant == "C":
        import torch.cuda.graphs as G
        orig = torch.cuda.graph.__init__
        def patched(self, *a, **k):
            k["capture_error_mode"] = "thread_local"; orig(self, *a, **k)
        torch.cuda.graph.__init__ = patched
        e = builder(); r = e(v); torch.cuda.synchronize()
    elif variant == "D":
        e = builder(); torch.cuda.synchronize()
    elif variant == "F":
        g0 = curvature.GraphedOperator(builder); r0 = g0(v).clone(); torch.cuda.synchronize()
    elif variant == "G":  # eager forward+backward only (plain training-style)
        o = m(x); lf(o, t).backward(); torch.cuda.synchronize()
    elif variant == "H":  # eager first, relaxed mode
        import torch.cuda.graphs as G
        orig = torch.cuda.graph.__init__
        def patched(self, *a, **k):
            k["capture_error_mode"] = "relaxed"; orig(self, *a, **k)
        torch.cuda.graph.__init__ = patched
        e = builder(); r = e(v); torch.cuda.synchronize()
    g = curvature.GraphedOperator(builder)
    r2 = g(v).clone(); torch.cuda.synchronize()
    print("OK", variant)
else:
    for var in "ABCDFGH":
        p = subprocess.run([sys.executable, __file__, var], capture_output=True, text=True)
        print(var, "rc", p.returncode, (p.stdout.strip().splitlines() or ["-"])[-1], flush=True)
