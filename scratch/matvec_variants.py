import sys, os, time, subprocess
sys.path.insert(0, os.getcwd())
if len(sys.argv) > 1:
    import torch
    from pytorchhessianfree_amd import curvature, testproblems as tp, modelprep
    variant = sys.argv[1]
    if "bench" in variant: torch.backends.cudnn.benchmark = True
    m, (x, t), lf = tp.resnet18_mnist(batch_size=32, device="cuda", data_seed=1000)
    if "bn" in variant: print("fused", modelprep.fuse_eval_batchnorm(m))
    if "conv" in variant: print("fusedconv", modelprep.fuse_conv_tangent(m))
    if "blk" in variant: print("fusedblk", modelprep.fuse_residual_blocks(m))
    if "cl" in variant:
        m = m.to(memory_format=torch.channels_last); x = x.contiguous(memory_format=torch.channels_last)
    ps = [p for p in m.parameters()]
    def builder():
        o = m(x); return curvature.GGNOperator(lf(o, t), o, ps)
    n = sum(p.numel() for p in ps)
    v = torch.randn(n, device="cuda")
    e = builder(); ref = e(v).clone(); del e
    import gc; gc.collect()
    g = curvature.GraphedOperator(builder, params=ps)
    g.input_buffer.copy_(v)
    for _ in range(5): g(g.input_buffer)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    K = 100
    for _ in range(K): g(g.input_buffer)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
    r = g(g.input_buffer)
    print("RESULT", variant, "ms/matvec %.3f" % (dt * 1e3), "relerr vs eager %.2e" % float((r - ref).abs().max() / ref.abs().max()), "refnorm %.4e" % float(ref.norm()))
else:
    db = os.path.join(os.getcwd(), "gpurun_out", "miopen_db2"); os.makedirs(db, exist_ok=True)
    import shutil
    for f in os.listdir("pytorchhessianfree_amd/miopen_db"): shutil.copy(os.path.join("pytorchhessianfree_amd/miopen_db", f), db)
    for var, envx in [("bn_conv_bench", {}), ("bn_conv_blk_bench", {})]:
        env = dict(os.environ, MIOPEN_USER_DB_PATH=db); env.update(envx)
        t0 = time.time()
        p = subprocess.run([sys.executable, __file__, var], capture_output=True, text=True, env=env)
        print(var, "rc", p.returncode, "%.0fs" % (time.time() - t0), [l for l in p.stdout.splitlines() if "RESULT" in l], flush=True)
        if p.returncode: print(p.stderr[-400:])
