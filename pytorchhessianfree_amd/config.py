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


# ----------------------------------------------------------------------------------------------------------------
# Every environment switch the package reads, once.  ``tests/test_boundary.py`` checks that no other ``HF_*`` name is
# read anywhere in the package and that each of these is exercised with a non-default value by the test named here.
# Round 5 removed the switches whose non-default value was a measured-and-rejected variant (their numbers stay in
# DESIGN.md / profiles/): HF_BN_TRAIN_FORM, HF_BN_FOLD, HF_BN_FWD_PROLOGUE, HF_BN_ROW_BLOCKS / _ROW_PASSES /
# _EPILOGUE_ROWS, HF_HESSIAN_PARALLEL, HF_ACC_PARALLEL, HF_CONV_BLOCKS / _BIG_BLOCKS / _FEW_TILES(_CAP) / _DCLASS /
# _AUTO / _WS_MB, HF_OWN_CONV, HF_AFFINE_SCALAR, HF_CARRY_SCATTER, HF_ENGINE_LIVE / _HEAD / _GROUP / _DIAG_EF,
# HF_DIAG_EF_GRAPH, HF_INPLACE_PREFIX, HF_COMPACT_ALLREDUCE, HF_CHUNK_TAIL, HF_RCCL_GROUPED, HF_CG_LAG,
# HF_TRAIN_SESSION.
# ----------------------------------------------------------------------------------------------------------------
SWITCHES = {
    # name: (default, what the non-default value does, test that exercises it)
    "HF_ENGINE": ("1", "0: curvature products by the autograd operators (no fused engine)",
                  "tests/test_engine_gpu.py::test_engine_declines_what_it_does_not_know"),
    "HF_ENGINE_VERIFY": ("first", "always / never: the engine's first-use check against the autograd product",
                         "tests/test_engine_gpu.py::test_train_mode_prologue_form_variants_agree_and_state_is_independent_of_the_first_use_check"),
    "HF_ENGINE_DEBUG": ("", "1: say why a model was not taken by the engine",
                        "tests/test_engine_gpu.py::test_engine_declines_what_it_does_not_know"),
    "HF_CONV": ("auto", "own / miopen: which convolution implementation the autograd path of a prepared model uses",
                "tests/test_optimizer_gpu.py::test_deterministic_mode_products_are_bitwise_repeatable"),
    "HF_NHWC_FIND": ("", "1: keep MIOpen's find step for NHWC problems (used once to produce the shipped records)",
                     "tests/test_host_logic_cpu.py::test_configure_is_explicit_and_idempotent"),
    "HF_SESSION": ("1", "0: no persistent engine session (engine + graphs rebuilt per step)",
                   "tests/test_session_gpu.py::test_session_equals_generic_path_and_is_faster_to_restart"),
    "HF_SESSION_VERIFY": ("", "1: re-verify the session against the model's own forward pass every step",
                          "tests/test_session_gpu.py::test_session_is_reverified_against_the_models_own_forward"),
    "HF_SESSION_VERIFY_EVERY": ("16", "every how many steps the session is re-verified",
                                "tests/test_session_gpu.py::test_session_is_reverified_against_the_models_own_forward"),
    "HF_ACC_SESSION": ("1", "0: acc_step on the generic accumulation",
                       "tests/test_acc_session_gpu.py::test_acc_step_train_mode_batchnorm_session_equals_generic_accumulation"),
    "HF_GRAPH_VERIFY": ("first", "always / never: replay check of a freshly captured product graph",
                        "tests/test_optimizer_gpu.py::test_graphed_operator_refuses_a_replay_that_differs_from_the_eager_product"),
    "HF_FUSE_ITERATION": ("1", "0: product, K1, K2, K3 as separate launches (no iteration graph)",
                          "tests/test_engine_gpu.py::test_solve_as_one_graph_per_iteration_equals_separate_launches"),
    "HF_BN_EPILOGUE": ("1", "0: train-mode tangent partial sums by the reduction launch, not the convolution's epilogue",
                       "tests/test_engine_gpu.py::test_train_mode_prologue_form_variants_agree_and_state_is_independent_of_the_first_use_check"),
    "HF_BN_TRAIN_PAIR": ("1", "0: a downsample block's two train-mode units in launches of their own",
                         "tests/test_engine_gpu.py::test_train_mode_prologue_form_variants_agree_and_state_is_independent_of_the_first_use_check"),
    "HF_CHUNKED_ALLREDUCE": ("auto", "0 / 1: force the single-graph resp. the two-phase data-parallel product",
                             "tests/test_distributed_gpu.py::test_step_two_ranks_engine_session_equals_cpu_whole_batch"),
    "HF_DIRECT_RCCL": ("1", "0: every collective through torch.distributed (no communicator of the package's own)",
                       "tests/test_distributed_gpu.py::test_bench_ladder_reaches_the_plainest_rung"),
    "HF_TEST_DP_FAULT": ("", "test hook: hang / raise / mismatch[:not_plain] -- the last rank misbehaves in a data-parallel "
                             "session product (session._inject_fault)",
                         "tests/test_distributed_gpu.py::test_bench_ladder_falls_past_a_rung_that_hangs"),
    "HF_PCG_BLOCKS": ("0", "workgroups of the PCG vector kernels (0: 2 per CU); must agree on all ranks",
                      "tests/test_cg_gpu.py::test_kernel_grid_override_gives_the_same_solve"),
    "HF_PCG_LIB": ("", "path of another build of libhfpcg.so (tuning variants)",
                   "tests/test_boundary.py::test_alternate_library_path"),
    "HF_ALLOW_WINOGRAD": ("0", "1: keep MIOpen's fp32 Winograd solvers", "tests/test_host_logic_cpu.py::test_configure_is_explicit_and_idempotent"),
    "HF_ALLOW_WRW_XDLOPS": ("0", "1: keep MIOpen's CK split-K weight gradient",
                            "tests/test_host_logic_cpu.py::test_configure_is_explicit_and_idempotent"),
    "HF_MIOPEN_DB": ("user", "<dir> / off: where the writable copy of the shipped MIOpen records lives",
                     "tests/test_host_logic_cpu.py::test_configure_is_explicit_and_idempotent"),
}
