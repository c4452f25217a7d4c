import sys, os, subprocess, collections
sys.path.insert(0, os.getcwd())
if len(sys.argv) > 1:
    import torch
    import pytorchhessianfree_amd as hf
    from pytorchhessianfree_amd import curvature, testproblems as tp, modelprep
    from torch.profiler import profile, ProfilerActivity
    def product(prep, cl, prof=False):
        m, (x, t), lf = tp.resnet18_mnist(batch_size=32, device="cuda", data_seed=1000)
        if prep: modelprep.prepare_model(m, channels_last=cl)
        ps = list(m.parameters())
        o = m(x); op = curvature.GGNOperator(lf(o, t), o, ps)
        v = torch.randn(op.n, device="cuda", generator=torch.Generator(device="cuda").manual_seed(0))
        op(v)
        names = set()
        if prof:
            with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as p:
                r = op(v).clone(); torch.cuda.synchronize()
            for e in p.events():
                for k in e.kernels:
                    n = k.name
                    if ("igemm" in n or "ck" in n.lower() or "Conv" in n or "conv" in n or "Cijk" in n) : names.add(n[:70])
        else:
            r = op(v).clone()
        return r, names
    # reference: float64 product on GPU (exact to 1e-15)
    m, (x, t), lf = tp.resnet18_mnist(batch_size=32, device="cuda", data_seed=1000)
    m = m.double(); x = x.double()
    ps = list(m.parameters()); o = m(x); op = curvature.GGNOperator(lf(o, t), o, ps)
    v = torch.randn(op.n, device="cuda", generator=torch.Generator(device="cuda").manual_seed(0))
    ref = op(v.double()).clone()
    r, names = product(True, True, prof=True)
    err = float((r.double() - ref).abs().max() / ref.abs().max())
    print("RESULT err %.2e" % err)
    for n in sorted(names): print("   K", n)
else:
    for i in range(5):
        db = os.path.join(os.getcwd(), "gpurun_out", "flaky_db_%d" % i); os.makedirs(db, exist_ok=True)
        env = dict(os.environ, MIOPEN_USER_DB_PATH=db)
        p = subprocess.run([sys.executable, __file__, "x"], capture_output=True, text=True, env=env)
        out = [l for l in p.stdout.splitlines() if l.startswith("RESULT") or l.startswith("   K")]
        print("run", i, out[0] if out else p.stderr[-300:]); 
        print("\n".join(o for o in out[1:] if "ck" in o.lower() or "Cijk" in o or "naive" in o), flush=True)
