"""Explicit, opt-in configuration of the convolution library underneath the
curvature products (MIOpen on PyTorch-ROCm).  NOTHING here runs at import:
``import pytorchhessianfree_amd`` leaves the process untouched.  ``configure()``
is called by ``modelprep.prepare_model`` (first call only), by ``bench.py`` and by
the test-suite's ``conftest``; a user who wants reference-grade fp32 products
from STOCK layers calls it once before the first convolution.

What it does and why (measured on MI355X, DESIGN.md sections 4 and 5), each with its
own switch:

``winograd=False``  (env ``HF_ALLOW_WINOGRAD=1`` keeps MIOpen's default)
    ``MIOPEN_DEBUG_CONV_WINOGRAD=0``.  MIOpen's fp32 Winograd solvers are not
    fp32-accurate: the stock gradient of the ResNet-18 workload is off by 2.3e-4
    (7e-4 in the first block) against float64, every other solver family stays at
    3e-7 at the same speed once MIOpen has measured its solvers.
``wrw_xdlops=False``  (env ``HF_ALLOW_WRW_XDLOPS=1``)
    ``MIOPEN_DEBUG_GROUP_CONV_IMPLICIT_GEMM_HIP_WRW_XDLOPS=0``.  The composable-
    kernel split-K weight gradient (memset + atomically accumulating kernel) came
    out wrong whenever the find step ranked it first: 1e-3 eagerly, garbage when
    replayed from a hipGraph.
``suggest_nhwc=True``
    ``PYTORCH_MIOPEN_SUGGEST_NHWC=1``: PyTorch hands channels_last tensors to
    MIOpen as NHWC instead of converting them back.
``find=True``
    ``torch.backends.cudnn.benchmark = True``: without Winograd, MIOpen's immediate
    mode can fall back to naive solvers (49 ms instead of 1.7 ms per product), so it
    is asked to measure its solvers once per shape.
``db="user"``  (env ``HF_MIOPEN_DB=<dir>|off``)
    ``MIOPEN_USER_DB_PATH`` -> a per-user WRITABLE copy of the find-db / perf-db
    records that ship in ``miopen_db/`` (BASELINE.json workloads).  MIOpen appends
    to its user db at run time; it never writes into the installed package.

The ``MIOPEN_*`` variables are read by MIOpen when it first needs them, i.e. at the
first convolution of the process: ``configure()`` after that point only changes
``cudnn.benchmark``.  Variables the user already set are left alone.
"""

import os
import shutil
import tempfile

_SHIPPED_DB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "miopen_db")
_state = {"done": False, "settings": None}


def user_db_dir():
    base = os.environ.get("XDG_CACHE_HOME") or os.path.join(os.path.expanduser("~"), ".cache")
    return os.path.join(base, "pytorchhessianfree_amd", "miopen_db")


def _seed_user_db(dst):
    """Copy the shipped records into ``dst`` unless a file of that name is already
    there (MIOpen may have appended to it).  Atomic per file, so ranks starting at
    the same time on a fresh machine do not see half-written records."""
    os.makedirs(dst, exist_ok=True)
    for name in sorted(os.listdir(_SHIPPED_DB)):
        src, out = os.path.join(_SHIPPED_DB, name), os.path.join(dst, name)
        if not os.path.isfile(src) or os.path.exists(out):
            continue
        fd, tmp = tempfile.mkstemp(dir=dst, prefix=name + ".")
        os.close(fd)
        try:
            shutil.copyfile(src, tmp)
            os.replace(tmp, out)
        finally:
            if os.path.exists(tmp):
                os.unlink(tmp)
    return dst


def configure(winograd=None, wrw_xdlops=None, suggest_nhwc=True, find=True, db=None, force=False):
    """Apply the settings described in the module docstring; idempotent (a second
    call is a no-op unless ``force``).  Returns the dict of what was set."""
    if _state["done"] and not force:
        return _state["settings"]
    env = os.environ
    if winograd is None:
        winograd = env.get("HF_ALLOW_WINOGRAD", "0") not in ("", "0")
    if wrw_xdlops is None:
        wrw_xdlops = env.get("HF_ALLOW_WRW_XDLOPS", "0") not in ("", "0")
    if db is None:
        db = env.get("HF_MIOPEN_DB", "user")
    applied = {}
    if not winograd:
        applied["MIOPEN_DEBUG_CONV_WINOGRAD"] = env.setdefault("MIOPEN_DEBUG_CONV_WINOGRAD", "0")
    if not wrw_xdlops:
        key = "MIOPEN_DEBUG_GROUP_CONV_IMPLICIT_GEMM_HIP_WRW_XDLOPS"
        applied[key] = env.setdefault(key, "0")
    if suggest_nhwc:
        applied["PYTORCH_MIOPEN_SUGGEST_NHWC"] = env.setdefault("PYTORCH_MIOPEN_SUGGEST_NHWC", "1")
    if db != "off" and "MIOPEN_USER_DB_PATH" not in env:
        path = user_db_dir() if db == "user" else db
        try:
            env["MIOPEN_USER_DB_PATH"] = _seed_user_db(path)
        except OSError:  # read-only home: MIOpen keeps its own default location
            pass
    applied["MIOPEN_USER_DB_PATH"] = env.get("MIOPEN_USER_DB_PATH")
    if find:
        import torch

        torch.backends.cudnn.benchmark = True
        applied["cudnn.benchmark"] = True
    _state["done"], _state["settings"] = True, applied
    return applied


def configured():
    return _state["done"]
