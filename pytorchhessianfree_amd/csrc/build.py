"""Build ``libhfpcg.so`` (the C-ABI shared library) in-tree with hipcc for gfx950.

Usage: ``python -m pytorchhessianfree_amd.csrc.build [--force]`` or
``__graft_entry__.build()``.  hipcc cross-compiles without a GPU.
"""

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
# one translation unit per kernel family: PCG solver | gather / scatter | BatchNorm / bias maps | convolutions |
# pooling + classifier heads | the RCCL shim
SOURCES = [os.path.join(HERE, f) for f in ("hf_pcg.hip", "hf_pack.hip", "hf_bn.hip", "hf_conv.hip", "hf_head.hip",
                                           "hf_rccl.hip")]
HDR = os.path.join(ROOT, "include", "hf_pcg.h")
HDRS_SHARED = [os.path.join(HERE, f) for f in ("hf_common.h", "hf_unpack.h")]
OUT = os.path.join(HERE, "libhfpcg.so")

FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-fPIC",
    # elementwise arithmetic mirrors the reference's separate roundings
    "-ffp-contract=off",
    "-fno-fast-math",
    "-Wall",
    "-Wno-unused-function",
]


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(f) > t for f in (*SOURCES, HDR, *HDRS_SHARED, __file__))


def build(force=False, verbose=True, out=None, extra_flags=()):
    """``out`` / ``extra_flags`` build tuning variants (e.g. ``-DHF_U2=4``) next to
    the default library; ``HF_PCG_LIB`` makes ``_lib`` load such a variant."""
    out = out or OUT
    if out == OUT and not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    import tempfile
    from concurrent.futures import ThreadPoolExecutor

    with tempfile.TemporaryDirectory(prefix="hfpcg_build_") as tmp:
        objs = [os.path.join(tmp, os.path.basename(src)[:-4] + ".o") for src in SOURCES]

        def compile_one(pair):
            src, obj = pair
            cmd = [hipcc, *FLAGS, *extra_flags, "-I", os.path.join(ROOT, "include"), "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)

        with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as pool:
            list(pool.map(compile_one, zip(SOURCES, objs)))
        link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", out, "-ldl"]
        if verbose:
            print(" ".join(link), flush=True)
        subprocess.run(link, check=True)
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv)
