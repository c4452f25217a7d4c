import os, subprocess, sys, json
sys.path.insert(0, os.getcwd())
from pytorchhessianfree_amd.csrc import build
variants = {"u421": [], "u441": ["-DHF_U2=4"], "u411": ["-DHF_U2=1"], "u222": ["-DHF_U1=2", "-DHF_U3=2"], "u848": ["-DHF_U1=8", "-DHF_U2=4", "-DHF_U3=8"]}
for name, fl in variants.items():
    out = f"/tmp/libhf_{name}.so"
    build.build(force=True, verbose=False, out=out, extra_flags=fl)
    for blocks in (512, 1024, 2048, 4096):
        env = dict(os.environ, HF_PCG_LIB=out, HF_PCG_BLOCKS=str(blocks))
        p = subprocess.run([sys.executable, "scripts/pcg_kernel_bench.py", "--sizes", "11175370,100000000", "--iters", "100"], env=env, capture_output=True, text=True)
        for l in p.stdout.splitlines():
            if l.startswith("{"):
                d = json.loads(l)
                print(name, blocks, d["n"], "k1 %.1f k2 %.1f k3 %.1f us | all %.0f GB/s" % (d["k1_us"], d["k2_us"], d["k3_us"], d["all_GBs"]), flush=True)
        if p.returncode: print(p.stderr[-300:])
