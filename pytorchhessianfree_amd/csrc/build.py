"""Build ``libhfpcg.so`` (the C-ABI shared library) in-tree with hipcc for gfx950.

Usage: ``python -m pytorchhessianfree_amd.csrc.build [--force]`` or
``__graft_entry__.build()``.  hipcc cross-compiles without a GPU.
"""

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SRC = os.path.join(HERE, "hf_pcg.hip")
SRC_CONV = os.path.join(HERE, "hf_conv.hip")
SRC_HEAD = os.path.join(HERE, "hf_head.hip")
SOURCES = [SRC, SRC_CONV, SRC_HEAD]
HDR = os.path.join(ROOT, "include", "hf_pcg.h")
HDR_SHARED = os.path.join(HERE, "hf_unpack.h")  # shared by hf_pcg.hip and hf_conv.hip
OUT = os.path.join(HERE, "libhfpcg.so")

FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-shared",
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
    return any(os.path.getmtime(f) > t for f in (*SOURCES, HDR, HDR_SHARED, __file__))


def build(force=False, verbose=True, out=None, extra_flags=()):
    """``out`` / ``extra_flags`` build tuning variants (e.g. ``-DHF_U2=4``) next to
    the default library; ``HF_PCG_LIB`` makes ``_lib`` load such a variant."""
    out = out or OUT
    if out == OUT and not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, *FLAGS, *extra_flags, "-I", os.path.join(ROOT, "include"), *SOURCES, "-o", out, "-ldl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv)
