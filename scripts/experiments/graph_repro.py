import sys, subprocess, os
sys.path.insert(0, os.getcwd())
if len(sys.argv) > 1:
    import torch
    from pytorchhessianfree_amd import curvature, testproblems as tp
    wl, batch, eager_first = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    make = tp.resnet18_mnist if wl == "resnet18" else tp.allcnnc_cifar100
    m, (x, t), lf = make(batch_size=batch, device="cuda")
    ps = [p for p in m.parameters()]
    def builder():
        o = m(x); return curvature.GGNOperator(lf(o, t), o, ps)
    v = torch.randn(sum(p.numel() for p in ps), device="cuda")
    if eager_first:
        e = builder(); r = e(v); torch.cuda.synchronize()
    g = curvature.GraphedOperator(builder)
    r2 = g(v).clone(); torch.cuda.synchronize()
    e = builder(); r = e(v)
    print("OK", wl, batch, eager_first, float((r - r2).abs().max() / r.abs().max()))
else:
    for wl in ("resnet18", "allcnnc"):
        for batch in (8, 32):
            for ef in (0, 1):
                p = subprocess.run([sys.executable, __file__, wl, str(batch), str(ef)], capture_output=True, text=True)
                print(wl, batch, ef, "rc", p.returncode, (p.stdout.strip().splitlines() or ["-"])[-1], flush=True)
                if p.returncode != 0:
                    print("   ", [l for l in p.stderr.splitlines() if "rror" in l][:3])
