import sys, os, subprocess
sys.path.insert(0, os.getcwd())
if len(sys.argv) > 1:
    import torch
    import pytorchhessianfree_amd  # env + benchmark
    cl = torch.channels_last
    shapes = [(1, 64, 7, 2, 3, 28), (64, 64, 3, 1, 1, 7), (64, 128, 3, 2, 1, 7), (128, 128, 3, 1, 1, 4), (64, 128, 1, 2, 0, 7),
              (128, 256, 3, 2, 1, 4), (256, 256, 3, 1, 1, 2), (128, 256, 1, 2, 0, 4), (256, 512, 3, 2, 1, 2), (512, 512, 3, 1, 1, 1), (256, 512, 1, 2, 0, 2)]
    B = 32; bad = 0
    g = torch.Generator(device="cuda").manual_seed(0)
    for (ci, co, k, s, p, H) in shapes:
        x = torch.randn(B, ci, H, H, device="cuda", generator=g); w = torch.randn(co, ci, k, k, device="cuda", generator=g) * 0.05
        y64 = torch.nn.functional.conv2d(x.double(), w.double(), None, s, p)
        gy = torch.randn(y64.shape, device="cuda", generator=g)
        gx64, gw64, _ = torch.ops.aten.convolution_backward(gy.double(), x.double(), w.double(), None, [s, s], [p, p], [1, 1], False, [0, 0], 1, [True, True, False])
        xc, wc, gyc = x.contiguous(memory_format=cl), w.contiguous(memory_format=cl), gy.contiguous(memory_format=cl)
        for rep in range(2):
            gx, gw, _ = torch.ops.aten.convolution_backward(gyc, xc, wc, None, [s, s], [p, p], [1, 1], False, [0, 0], 1, [True, True, False])
            xcat = torch.cat([xc, xc], 1).contiguous(memory_format=cl); wcat = torch.cat([wc, wc], 1).contiguous(memory_format=cl)
            y2 = torch.nn.functional.conv2d(xcat, wcat, None, s, p)
            torch.cuda.synchronize()
        names = []
        def rel(a, b): return float((a.double() - b).abs().max() / b.abs().max())
        e = (rel(gx, gx64), rel(gw, gw64), rel(y2, 2 * y64))
        flag = "BAD" if max(e) > 1e-4 or any(v != v for v in e) else "ok"
        if flag == "BAD": bad += 1
        print(flag, (ci, co, k, s, p, H), "gx %.1e gw %.1e y2 %.1e" % e, "| gw fmt cl=%s contig=%s" % (gw.is_contiguous(memory_format=cl), gw.is_contiguous()), names if flag == "BAD" else "")
    print("BADCOUNT", bad)
else:
    for i in range(6):
        db = os.path.join(os.getcwd(), "gpurun_out", "solver_db_%d" % i); os.makedirs(db, exist_ok=True)
        p = subprocess.run([sys.executable, __file__, "x"], capture_output=True, text=True, env=dict(os.environ, MIOPEN_USER_DB_PATH=db))
        lines = [l for l in p.stdout.splitlines() if l.startswith("BAD") or l.startswith("ok")]
        print("run", i, [l for l in lines if l.startswith("BADCOUNT")], flush=True)
        for l in lines:
            if l.startswith("BAD "): print("   ", l[:400])
        if p.returncode: print(p.stderr[-400:])
