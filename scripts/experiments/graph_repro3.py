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
    if variant == "D1":   # builder kept alive + empty_cache
        e = builder(); torch.cuda.synchronize(); gc.collect(); torch.cuda.empty_cache()
    elif variant == "G1":  # backward + empty_cache
        o = m(x); lf(o, t).backward(); torch.cuda.synchronize(); gc.collect(); torch.cuda.empty_cache()
    elif variant == "G2":  # forward only, no grad
        with torch.no_grad(): o = m(x)
        torch.cuda.synchronize()
    elif variant == "G3":  # forward with grad, result dropped
        o = m(x); del o; torch.cuda.synchronize()
    elif variant == "G4":  # autograd.grad like bench
        g_ = torch.autograd.grad(lf(m(x), t), ps); torch.cuda.synchronize()
    elif variant == "G5":  # autograd.grad then drop + no empty cache
        g_ = torch.autograd.grad(lf(m(x), t), ps); del g_; torch.cuda.synchronize()
    elif variant == "G6":  # backward, then set grads None, no empty_cache
        o = m(x); lf(o, t).backward(); torch.cuda.synchronize()
        for p in ps: p.grad = None
        del o; gc.collect()
    g = curvature.GraphedOperator(builder)
    r2 = g(v).clone(); torch.cuda.synchronize()
    print("OK", variant)
else:
    for var in ["D1","G1","G2","G3","G4","G5","G6"]:
        p = subprocess.run([sys.executable, __file__, var], capture_output=True, text=True)
        print(var, "rc", p.returncode, (p.stdout.strip().splitlines() or ["-"])[-1], flush=True)
