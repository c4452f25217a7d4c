"""ctypes binding of ``libhfpcg.so`` (C ABI declared in ``include/hf_pcg.h``).

The library is loaded AFTER ``import torch`` so that its ``DT_NEEDED``
``libamdhip64.so.7`` resolves to the HIP runtime PyTorch-ROCm already mapped
(same SONAME) -- kernels, streams and device pointers then live in ONE runtime.
There is no fallback: if the shared object is missing or a call fails, a
``RuntimeError`` is raised.
"""

import ctypes
import os

import torch  # noqa: F401  (must be imported first, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HF_PCG_LIB") or os.path.join(_HERE, "csrc", "libhfpcg.so")

HF_F32, HF_F64 = 0, 1
ABI_VERSION = 13
HF_ERR_ARG = -1  # hf_status of include/hf_pcg.h: null / negative / inconsistent argument
HF_M_NONE, HF_M_DIAG, HF_M_EXTERNAL = 0, 1, 2
REASONS = {
    1: "Convergence (Martens)",
    2: "Number of iterations",
    3: "Divergence",
    4: "Convergence (tolerances)",
}

c_void_p, c_int, c_int64, c_double = (
    ctypes.c_void_p,
    ctypes.c_int,
    ctypes.c_int64,
    ctypes.c_double,
)


class Status(ctypes.Structure):
    _fields_ = [
        ("done", ctypes.c_int32),
        ("reason", ctypes.c_int32),
        ("n_iters", c_int64),
        ("iter_next", c_int64),
        ("nonpos_count", c_int64),
        ("last_alpha", c_double),
        ("last_beta", c_double),
        ("last_pAp", c_double),
        ("last_res_norm", c_double),
        ("res_bound", c_double),
        ("n_stored", c_int64),
    ]


# name -> (restype, argtypes); must list EVERY symbol include/hf_pcg.h declares
SIGNATURES = {
    "hf_abi_version": (c_int, []),
    "hf_error_string": (ctypes.c_char_p, [c_int]),
    "hf_pcg_create": (c_int, [ctypes.POINTER(c_void_p), c_int64, c_int, c_int]),
    "hf_pcg_destroy": (c_int, [c_void_p]),
    "hf_pcg_begin": (
        c_int,
        [c_void_p] * 6
        + [c_int, c_int64, c_double, c_double, c_int, c_void_p, c_int64, c_int, c_void_p, c_int64, c_void_p],
    ),
    "hf_pcg_init": (c_int, [c_void_p, c_void_p, c_void_p]),
    "hf_pcg_init_external": (c_int, [c_void_p, c_void_p, c_void_p]),
    "hf_pcg_iterate": (c_int, [c_void_p, c_void_p, c_double, c_void_p]),
    "hf_pcg_curvature": (c_int, [c_void_p, c_void_p, c_double, c_void_p]),
    "hf_pcg_update_xr": (c_int, [c_void_p, c_void_p, c_double, c_void_p]),
    "hf_pcg_update_p": (c_int, [c_void_p, c_void_p, c_void_p]),
    "hf_pcg_graph_create": (c_int, [ctypes.POINTER(c_void_p), c_void_p, c_void_p, c_void_p, c_double, c_int]),
    "hf_pcg_graph_update": (c_int, [c_void_p, c_void_p, c_double]),
    "hf_pcg_graph_launch": (c_int, [c_void_p, c_int, c_void_p]),
    "hf_pcg_graph_collect_timing": (c_int, [c_void_p]),
    "hf_pcg_graph_destroy": (c_int, [c_void_p]),
    "hf_pcg_poll": (c_int, [c_void_p, ctypes.POINTER(Status)]),
    "hf_pcg_finish": (c_int, [c_void_p, ctypes.POINTER(Status), c_void_p]),
    "hf_pcg_read_nonpos": (
        c_int,
        [c_void_p, ctypes.POINTER(c_int64), ctypes.POINTER(c_double), c_int],
    ),
    "hf_pcg_timing_enable": (c_int, [c_void_p, c_int]),
    "hf_pcg_timing_read": (
        c_int,
        [c_void_p] + [ctypes.POINTER(c_double)] * 3 + [ctypes.POINTER(c_int64)],
    ),
    "hf_pack": (
        c_int,
        [c_void_p, ctypes.POINTER(c_void_p), ctypes.POINTER(c_int64), ctypes.POINTER(c_int64), c_int,
         c_double, c_int, c_int, c_void_p],
    ),
    "hf_unpack_tangent": (
        c_int,
        [c_void_p, ctypes.POINTER(c_void_p)] + [ctypes.POINTER(c_int64)] * 4 + [c_int, c_int, c_void_p],
    ),
    "hf_unpack_tangent_ex": (
        c_int,
        [c_void_p, ctypes.POINTER(c_void_p)] + [ctypes.POINTER(c_int64)] * 5 + [c_int, c_int, c_void_p],
    ),
    "hf_unpack_weights": (
        c_int,
        [c_void_p, ctypes.POINTER(c_void_p)] + [ctypes.POINTER(c_int64)] * 6 + [c_int, c_int, c_void_p],
    ),
    "hf_bn_forward": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int, c_int64] + [c_void_p] * 5
                      + [c_int64, c_int, c_int64, c_int64, c_int, c_void_p]),
    "hf_maxpool_forward_nhwc": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p] + [c_int64] * 12
                                + [c_int, c_void_p]),
    "hf_live_copy": (c_int, [c_void_p, c_void_p, c_int] + [ctypes.POINTER(c_int64)] * 4 + [c_int, c_int, c_void_p]),
    "hf_precond_build": (c_int, [c_void_p, c_void_p, c_double, c_double, c_int64, c_int, c_void_p]),
    "hf_axpy_out": (c_int, [c_void_p, c_void_p, c_void_p, c_double, c_int64, c_int, c_void_p]),
    "hf_chan_affine": (c_int, [c_void_p] * 10 + [c_int, c_int64, c_int64, c_int64, c_int, c_int64, c_int64,
                                                  c_int, c_void_p]),
    "hf_chan_affine_bwd": (c_int, [c_void_p] * 11 + [c_int64, c_int64, c_int64, c_int, c_int, c_void_p]),
    "hf_conv2d_nhwc": (c_int, [c_int, c_void_p, c_void_p, c_void_p] + [c_int64] * 12
                       + [c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "hf_conv2d_nhwc_backward": (c_int, [c_void_p] * 5 + [c_int64] * 11
                                + [c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "hf_conv2d_nhwc_plan": (c_int, [c_int] + [c_int64] * 11 + [c_int]),
    "hf_conv2d_nhwc_slabs": (c_int, [c_int, c_void_p, c_void_p, c_void_p] + [c_int64] * 14
                             + [c_int, c_int64, c_int, c_void_p]),
    "hf_conv2d_nhwc_slabs_unpack": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int64] * 13 + [c_int, c_int64, c_void_p,
                                            ctypes.POINTER(c_void_p)] + [ctypes.POINTER(c_int64)] * 6
                                    + [c_int, c_int, c_void_p]),
    "hf_conv2d_nhwc_group_slabs": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "hf_conv2d_nhwc_dw_slabs": (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    "hf_chan_affine_pair": (c_int, [c_void_p, c_int, c_void_p]),
    "hf_chan_affine_bwd_pair": (c_int, [c_void_p, c_int, c_void_p]),
    "hf_conv2d_nhwc_backward_slabs": (c_int, [c_void_p] * 5 + [c_int64] * 11
                                      + [c_int, c_int64, c_int, c_int64, c_int, c_void_p]),
    "hf_chan_affine_ex": (c_int, [c_void_p] * 10 + [c_int, c_int64, c_int64, c_int64, c_int, c_int64, c_int64,
                                                     c_int, c_int64, c_int, c_void_p]),
    "hf_chan_affine_bwd_ex": (c_int, [c_void_p] * 5 + [c_int, c_int64, c_void_p, c_int, c_int64] + [c_void_p] * 5
                              + [c_int64, c_int64, c_int64, c_int, c_int, c_int, c_void_p]),
    "hf_chan_affine_train": (c_int, [c_void_p] * 8 + [c_int, c_void_p, c_void_p, c_double, c_void_p, c_void_p]
                             + [c_int64] * 5 + [c_int, c_int64, c_int, c_void_p]),
    "hf_conv2d_nhwc_group_slabs_bnsum": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p]),
    "hf_bn_forward_train": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int] + [c_void_p] * 4
                            + [c_double] * 3 + [c_void_p] * 3 + [c_int64, c_int, c_int64, c_int64, c_int, c_void_p]),
    "hf_chan_affine_train_pair": (c_int, [c_void_p, c_int, c_void_p]),
    "hf_bn_train_hessian_coeffs": (c_int, [c_void_p] * 5 + [c_int, c_void_p, c_void_p, c_int] + [c_void_p] * 5
                                   + [c_double, c_int64, c_int, c_void_p]),
    "hf_bn_train_hessian_apply": (c_int, [c_void_p] * 5 + [c_int, c_int64] + [c_void_p] * 4 + [c_int64, c_int64, c_int, c_void_p]),
    "hf_bn_stats_rows": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_void_p, c_int64, c_int64, c_int, c_int, c_void_p]),
    "hf_bn_adjoint_pre": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_void_p, c_int, c_int64,
                                  c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_void_p]),
    "hf_pack_ex": (
        c_int,
        [c_void_p, ctypes.POINTER(c_void_p), ctypes.POINTER(c_int64), ctypes.POINTER(c_int64),
         ctypes.POINTER(c_int64), ctypes.POINTER(c_int64), c_int, c_double, c_int, c_int, c_void_p],
    ),
    "hf_softmax_ce_hvp": (c_int, [c_void_p, c_void_p, c_void_p, c_double, c_int64, c_int64, c_int, c_void_p]),
    "hf_maxpool_tangent_nhwc": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int64] * 7 + [c_int, c_void_p]),
    "hf_maxpool_adjoint_nhwc": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_void_p, c_int, c_int64, c_void_p]
                                + [c_int64] * 12 + [c_int, c_void_p]),
    "hf_linear_ce_head": (c_int, [c_void_p] * 9 + [c_double, c_int64, c_int64, c_int64, c_int, c_void_p]),
    "hf_linear_ce_head_slabs": (c_int, [c_int64]),
    "hf_pool_ce_head": (c_int, [c_void_p] * 4 + [c_double, c_int64, c_int64, c_int64, c_int, c_void_p]),
    "hf_comm_unique_id": (c_int, [ctypes.c_char_p]),
    "hf_comm_create": (c_int, [ctypes.POINTER(c_void_p), ctypes.c_char_p, c_int, c_int]),
    "hf_comm_destroy": (c_int, [c_void_p]),
    "hf_allreduce_sum": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "hf_allreduce_sum_multi": (c_int, [c_void_p, ctypes.POINTER(c_void_p), ctypes.POINTER(c_int64), c_int, c_int,
                                       c_void_p]),
}

_lib = None


def load():
    """Load the shared library (once) and type every entry point."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH) and not os.environ.get("HF_PCG_LIB"):
        # in-tree build on first use (hipcc cross-compiles in ~20 s); still an error
        # if the toolchain is missing -- there is no other implementation to fall back to
        try:
            from .csrc import build as _build

            if os.path.abspath(LIB_PATH) == os.path.abspath(_build.OUT):
                _build.build(force=True, verbose=False)
        except Exception:
            pass
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP extension is REQUIRED (there is no "
            "CPU fallback). Build it with `python -m pytorchhessianfree_amd.csrc.build`."
        )
    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.hf_abi_version() != ABI_VERSION:
        raise RuntimeError("libhfpcg.so ABI version mismatch")
    _lib = lib
    return lib


class Refused(RuntimeError):
    """The library REFUSED its arguments (``HF_ERR_ARG`` / ``HF_ERR_ALIGN`` / ``HF_ERR_CAPACITY``: a geometry,
    alignment or size it does not cover) -- nothing was launched; callers with another implementation of the same
    operation may fall back to it.  Any other non-zero status (a HIP launch error, a call-order violation) is a
    plain ``RuntimeError`` and must not be swallowed."""


_REFUSALS = (-1, -2, -5)  # HF_ERR_ARG, HF_ERR_ALIGN, HF_ERR_CAPACITY of include/hf_pcg.h


def check(code, what=""):
    if code != 0:
        msg = load().hf_error_string(code)
        msg = msg.decode() if msg else "?"
        kind = Refused if code in _REFUSALS else RuntimeError
        raise kind(f"libhfpcg {what} failed with code {code}: {msg}")


def dtype_code(dtype):
    if dtype == torch.float32:
        return HF_F32
    if dtype == torch.float64:
        return HF_F64
    raise TypeError(f"libhfpcg supports float32/float64 vectors, not {dtype}")


def current_stream_ptr(device):
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)


def require_device_tensor(t, name):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"`{name}` should be a torch.Tensor, not {type(t)}.")
    if not t.is_cuda:
        raise RuntimeError(
            f"`{name}` lives on {t.device}: the PCG solver runs only as HIP kernels "
            "on an AMD GPU (no CPU fallback exists in this package)."
        )


# ---- thin helpers used by several modules -----------------------------------
def pack(dst, tensors, scale=1.0, mode=0):
    """dst[off_t : off_t+numel_t] = scale * tensors[t]  (mode 0)
    dst[...] += (scale * tensors[t])**2                 (mode 1)
    One multi-tensor gather launch instead of ``torch.cat``."""
    lib = load()
    require_device_tensor(dst, "dst")
    n = len(tensors)
    ptrs = (c_void_p * n)()
    numels = (c_int64 * n)()
    perm = (c_int64 * (2 * n))()
    keep = []
    total = 0
    for i, t in enumerate(tensors):
        if t.dtype != dst.dtype or t.device != dst.device:
            raise RuntimeError("pack: dtype/device mismatch")
        if not t.is_contiguous():
            if t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last):
                # stored (O, H, W, I): un-permuted inside the gather, no copy
                perm[2 * i], perm[2 * i + 1] = t.shape[1], t.shape[2] * t.shape[3]
            else:
                t = t.contiguous()
        keep.append(t)
        ptrs[i] = t.data_ptr()
        numels[i] = t.numel()
        total += t.numel()
    if total != dst.numel():
        raise RuntimeError(f"pack: {total} source elements for a vector of {dst.numel()}")
    check(
        lib.hf_pack(
            c_void_p(dst.data_ptr()), ptrs, numels, perm, n, float(scale), int(mode),
            dtype_code(dst.dtype), current_stream_ptr(dst.device),
        ),
        "hf_pack",
    )
    return dst


def unpack_tangent(v, slots, half=1):
    """Scatter weight-shaped slices of the flat vector ``v`` into the ``v_W`` halves
    (``half=1``) or the ``W`` halves (``half=0``: ``v`` is the parameter vector) of
    the tangent convolutions' weight buffers, all layers in one launch.  ``slots`` is
    a list of ``(offset into v, buffer [O, 2I, H, W], I)``; the buffer is NCHW-contiguous
    or channels_last."""
    lib = load()
    require_device_tensor(v, "v")
    table = unpack_table(v, slots, half)
    check(
        lib.hf_unpack_weights(c_void_p(v.data_ptr()), *table, dtype_code(v.dtype), current_stream_ptr(v.device)),
        "hf_unpack_weights",
    )


def unpack_table(v, slots, half=1):
    """The argument arrays of ``hf_unpack_weights`` for ``slots`` (see :func:`unpack_tangent`):
    ``(dsts, offs, numels, slabs, inners, live, halves, n)``."""
    n = len(slots)
    dsts = (c_void_p * n)()
    offs, numels, slabs, inners, live, halves = ((c_int64 * n)() for _ in range(6))
    for k, slot in enumerate(slots):
        halves[k] = int(half)  # 0: W half, 1: v_W half, 2: dense (I, H, W, O) transposed copy
        off, buf, cin = slot[:3]
        live[k] = slot[3] if len(slot) > 3 else 0  # bit mask of the kernel taps that meet data (0 = all)
        if half == 2:  # buf: (I, H, W, O) dense
            if buf.dtype != v.dtype or buf.device != v.device or buf.dim() != 4 or buf.shape[0] != cin or \
                    not buf.is_contiguous():
                raise RuntimeError("unpack_tangent: transposed buffer does not match")
            hw = buf.shape[1] * buf.shape[2]
            inners[k] = cin
            dsts[k] = buf.data_ptr()
            offs[k] = off
            slabs[k] = cin * hw
            numels[k] = buf.shape[3] * cin * hw
            if off < 0 or off + numels[k] > v.numel():
                raise RuntimeError("unpack_tangent: slice outside the vector")
            live[k] = 0
            continue
        if buf.dtype != v.dtype or buf.device != v.device or buf.dim() != 4 or buf.shape[1] != 2 * cin:
            raise RuntimeError("unpack_tangent: buffer does not match")
        hw = buf.shape[2] * buf.shape[3]
        if buf.is_contiguous():
            inners[k] = 0
        elif buf.is_contiguous(memory_format=torch.channels_last):
            inners[k] = cin
        else:
            raise RuntimeError("unpack_tangent: buffer is neither NCHW- nor NHWC-contiguous")
        dsts[k] = buf.data_ptr()
        offs[k] = off
        slabs[k] = cin * hw
        numels[k] = buf.shape[0] * cin * hw
        if off < 0 or off + numels[k] > v.numel():
            raise RuntimeError("unpack_tangent: slice outside the vector")
    return dsts, offs, numels, slabs, inners, live, halves, n


def softmax_ce_hvp(p, v, scale):
    """scale * p * (v - <p, v>) row-wise (Hessian of softmax cross-entropy applied to v)."""
    lib = load()
    require_device_tensor(v, "v")
    p, v = p.contiguous(), v.contiguous()
    if p.shape != v.shape or p.dim() != 2 or p.dtype != v.dtype:
        raise RuntimeError("softmax_ce_hvp: p and v must be [rows, cols] of one dtype")
    out = torch.empty_like(v)
    check(
        lib.hf_softmax_ce_hvp(
            c_void_p(out.data_ptr()), c_void_p(p.data_ptr()), c_void_p(v.data_ptr()), float(scale),
            v.shape[0], v.shape[1], dtype_code(v.dtype), current_stream_ptr(v.device),
        ),
        "hf_softmax_ce_hvp",
    )
    return out


def axpy_out(out, a, s, alpha):
    """out = a + alpha * s (separately rounded mul and add, like the reference)."""
    lib = load()
    require_device_tensor(out, "out")
    check(
        lib.hf_axpy_out(
            c_void_p(out.data_ptr()), c_void_p(a.data_ptr()), c_void_p(s.data_ptr()),
            float(alpha), out.numel(), dtype_code(out.dtype), current_stream_ptr(out.device),
        ),
        "hf_axpy_out",
    )
    return out


def precond_build(minv, diag, damping, exponent):
    lib = load()
    require_device_tensor(minv, "minv")
    check(
        lib.hf_precond_build(
            c_void_p(minv.data_ptr()), c_void_p(diag.data_ptr()), float(damping),
            float(exponent), minv.numel(), dtype_code(minv.dtype),
            current_stream_ptr(minv.device),
        ),
        "hf_precond_build",
    )
    return minv


_conv_scratch = {}


def conv_scratch(device):
    """Per-device scratch of the split-K convolution kernels: partial-tile workspace and
    the (self-resetting) ticket counters.  Shared by every call on the device: the calls
    of one product are enqueued on one stream / replayed from one graph, in order."""
    key = device.index if device.index is not None else torch.cuda.current_device()
    sc = _conv_scratch.get(key)
    if sc is None:
        ws = torch.empty(64 << 18, dtype=torch.float32, device=device)
        tickets = torch.zeros(8192, dtype=torch.int32, device=device)
        sc = _conv_scratch[key] = (ws, tickets)
    return sc


def conv2d_nhwc(direction, out, act, mat, n, h, w, c, k, r, s, stride, padding, act_ld=0):
    """``hf_conv2d_nhwc`` on raw NHWC buffers (see include/hf_pcg.h)."""
    lib = load()
    require_device_tensor(out, "out")
    ws, tickets = conv_scratch(out.device)
    check(
        lib.hf_conv2d_nhwc(
            int(direction), c_void_p(out.data_ptr()), c_void_p(act.data_ptr()), c_void_p(mat.data_ptr()),
            n, h, w, c, k, r, s, stride[0], stride[1], padding[0], padding[1], act_ld,
            c_void_p(ws.data_ptr()), ws.numel() * 4, c_void_p(tickets.data_ptr()), tickets.numel(),
            0, HF_F32, current_stream_ptr(out.device)),
        "hf_conv2d_nhwc")
    return out


def conv2d_nhwc_backward(dx, dw, dy, x, w_t, n, h, w, c, k, r, s, stride, padding):
    """Data and weight gradient in one launch (``hf_conv2d_nhwc_backward``)."""
    lib = load()
    require_device_tensor(dx, "dx")
    ws, tickets = conv_scratch(dx.device)
    check(
        lib.hf_conv2d_nhwc_backward(
            c_void_p(dx.data_ptr()), c_void_p(dw.data_ptr()), c_void_p(dy.data_ptr()), c_void_p(x.data_ptr()),
            c_void_p(w_t.data_ptr()), n, h, w, c, k, r, s, stride[0], stride[1], padding[0], padding[1],
            c_void_p(ws.data_ptr()), ws.numel() * 4, c_void_p(tickets.data_ptr()), tickets.numel(),
            0, HF_F32, current_stream_ptr(dx.device)),
        "hf_conv2d_nhwc_backward")
    return dx, dw


def pack_ex(dst, tensors, perms, splits, scale=1.0, live=None, mode=0):
    """``pack`` for sources the caller describes itself: ``perms[i] = (I, H*W)`` marks tensor i as
    stored (O, H, W, I); ``splits[i] = (count, stride)`` makes it the sum of ``count`` split-K
    slabs ``stride`` elements apart (``hf_pack_ex``); ``live[i]`` = bit mask of the kernel taps
    whose gradients are not structurally zero.  ``tensors[i]`` is the first slab."""
    lib = load()
    require_device_tensor(dst, "dst")
    n = len(tensors)
    ptrs = (c_void_p * n)()
    numels = (c_int64 * n)()
    perm = (c_int64 * (2 * n))()
    spl = (c_int64 * (2 * n))()
    lv = (c_int64 * n)()
    total = 0
    for i, t in enumerate(tensors):
        if t.dtype != dst.dtype or t.device != dst.device:
            raise RuntimeError("pack_ex: dtype/device mismatch")
        ptrs[i] = t.data_ptr()
        numels[i] = t.numel()
        total += t.numel()
        if i in perms:
            perm[2 * i], perm[2 * i + 1] = perms[i]
        spl[2 * i], spl[2 * i + 1] = splits.get(i, (1, 0))
        lv[i] = (live or {}).get(i, 0)
    if total != dst.numel():
        raise RuntimeError(f"pack_ex: {total} source elements for a vector of {dst.numel()}")
    check(
        lib.hf_pack_ex(c_void_p(dst.data_ptr()), ptrs, numels, perm, spl, lv, n, float(scale), int(mode),
                       dtype_code(dst.dtype), current_stream_ptr(dst.device)),
        "hf_pack_ex")
    return dst


class AffineProblem(ctypes.Structure):
    """``hf_affine_problem`` of include/hf_pcg.h."""

    _fields_ = ([("out", c_void_p)] + [(nm, c_void_p) for nm in ("a", "x", "mean", "rstd", "w", "q", "r", "add",
                                                                   "mask_src")]
                + [("relu_self", c_int)] + [(nm, c_int64) for nm in ("n", "c", "hw", "out_ld", "add_ld")]
                + [("a_splits", c_int), ("a_slab", c_int64)])


class BnAdjointProblem(ctypes.Structure):
    """``hf_bn_adjoint_problem`` of include/hf_pcg.h."""

    _fields_ = ([(nm, c_void_p) for nm in ("gx", "gw", "gb", "gres", "gy")] + [("gy_splits", c_int), ("gy_slab", c_int64),
                                                                                 ("gy2", c_void_p), ("gy2_splits", c_int),
                                                                                 ("gy2_slab", c_int64)]
                + [(nm, c_void_p) for nm in ("x", "mean", "rstd", "w", "mask_src")]
                + [(nm, c_int64) for nm in ("n", "c", "hw")] + [("row_blocks", c_int)])


class AffineTrainProblem(ctypes.Structure):
    """``hf_affine_train_problem`` of include/hf_pcg.h."""

    _fields_ = ([("out", c_void_p)] + [(nm, c_void_p) for nm in ("a", "x", "mean", "rstd", "w", "part_x", "part_1")]
                + [("nparts", c_int), ("vq", c_void_p), ("vr", c_void_p), ("count", ctypes.c_double),
                   ("add", c_void_p), ("mask_src", c_void_p)]
                + [(nm, c_int64) for nm in ("n", "c", "hw", "out_ld")] + [("a_splits", c_int), ("a_slab", c_int64)])


class ConvProblem(ctypes.Structure):
    """``hf_conv_problem`` of include/hf_pcg.h."""

    _fields_ = [("direction", c_int), ("out", c_void_p), ("act", c_void_p), ("mat", c_void_p)] + [
        (name, c_int64) for name in ("n", "h", "w", "c", "k", "r", "s", "stride_h", "stride_w", "pad_h", "pad_w",
                                     "act_ld", "out_c")] + [("splits", c_int), ("slab_stride", c_int64),
                                                            ("mat_ld", c_int64)]


def conv_group_slabs(problems, device):
    """One launch for up to 4 slab-mode convolutions; ``problems``: tuples ``(direction, out, act, mat,
    (n, h, w, c, k, r, s, stride, padding), splits, act_ld, out_c)`` with ``out`` a [splits, ...] buffer."""
    arr = (ConvProblem * len(problems))()
    for q, prob in zip(arr, problems):
        direction, out, act, mat, geo, splits, act_ld, out_c = prob[:8]
        q.mat_ld = prob[8] if len(prob) > 8 else 0
        n, h, w, c, k, r, s, st, pd = geo
        q.direction, q.out, q.act, q.mat = int(direction), out.data_ptr(), act.data_ptr(), mat.data_ptr()
        q.n, q.h, q.w, q.c, q.k, q.r, q.s = n, h, w, c, k, r, s
        q.stride_h, q.stride_w, q.pad_h, q.pad_w = st[0], st[1], pd[0], pd[1]
        q.act_ld, q.out_c, q.splits = act_ld, out_c, splits
        q.slab_stride = out.shape[1] if out.dim() == 2 else 0
    check(load().hf_conv2d_nhwc_group_slabs(ctypes.cast(arr, c_void_p), len(problems), HF_F32,
                                            current_stream_ptr(device)), "hf_conv2d_nhwc_group_slabs")


class ConvBnSum(ctypes.Structure):
    """``hf_conv_bnsum`` (include/hf_pcg.h)."""

    _fields_ = [("x", c_void_p), ("mean", c_void_p), ("rstd", c_void_p), ("part_x", c_void_p), ("part_1", c_void_p),
                ("part_rows", c_int64)]


def conv_group_slabs_bnsum(problems, sums, device):
    """``conv_group_slabs`` for tangent convolutions whose epilogue also writes the per-channel partial sums of a
    train-mode BatchNorm behind them; ``sums``: per problem ``None`` or ``(x, mean, rstd, part_x, part_1)`` with
    ``part_*`` of shape [ceil(rows / 64) * splits, k].  Returns False when the library refuses the geometry (the caller
    issues the plain launch + the reduction launch)."""
    arr = (ConvProblem * len(problems))()
    bn = (ConvBnSum * len(problems))()
    for q, b, prob, sm in zip(arr, bn, problems, sums):
        _fill_problem(q, prob)
        if sm is not None:
            x, mean, rstd, part_x, part_1 = sm
            b.x, b.mean, b.rstd = x.data_ptr(), mean.data_ptr(), rstd.data_ptr()
            b.part_x, b.part_1, b.part_rows = part_x.data_ptr(), part_1.data_ptr(), part_1.shape[0]
    rc = load().hf_conv2d_nhwc_group_slabs_bnsum(ctypes.cast(arr, c_void_p), len(problems), ctypes.cast(bn, c_void_p),
                                                 HF_F32, current_stream_ptr(device))
    if rc == HF_ERR_ARG:
        return False
    check(rc, "hf_conv2d_nhwc_group_slabs_bnsum")
    return True


def _fill_problem(q, prob):
    direction, out, act, mat, geo, splits, act_ld, out_c = prob[:8]
    q.mat_ld = prob[8] if len(prob) > 8 else 0
    n, h, w, c, k, r, s, st, pd = geo
    q.direction, q.out, q.act, q.mat = int(direction), out.data_ptr(), act.data_ptr(), mat.data_ptr()
    q.n, q.h, q.w, q.c, q.k, q.r, q.s = n, h, w, c, k, r, s
    q.stride_h, q.stride_w, q.pad_h, q.pad_w = st[0], st[1], pd[0], pd[1]
    q.act_ld, q.out_c, q.splits = act_ld, out_c, splits
    q.slab_stride = out.shape[1] if out.dim() == 2 else 0


def conv_dw_slabs(d_problem, w_problem, device):
    """One data-gradient (or forward) problem and one weight-gradient problem in ONE launch; tuples as for
    ``conv_group_slabs``."""
    arr = (ConvProblem * 2)()
    _fill_problem(arr[0], d_problem)
    _fill_problem(arr[1], w_problem)
    check(load().hf_conv2d_nhwc_dw_slabs(ctypes.byref(arr[0]), ctypes.byref(arr[1]), HF_F32,
                                         current_stream_ptr(device)), "hf_conv2d_nhwc_dw_slabs")


def conv_plan(direction, n, h, w, c, k, r, s, stride, padding):
    """Number of K-splits a slab-mode launch of this geometry uses (``hf_conv2d_nhwc_plan``)."""
    sp = load().hf_conv2d_nhwc_plan(int(direction), n, h, w, c, k, r, s, stride[0], stride[1], padding[0],
                                    padding[1], 0)
    if sp < 1:
        check(sp, "hf_conv2d_nhwc_plan")
    return sp


def conv2d_nhwc_slabs(direction, out, act, mat, n, h, w, c, k, r, s, stride, padding, splits, act_ld=0, out_c=0,
                      mat_ld=0):
    """Slab-mode launch: ``out`` is [splits, numel] -- slab s receives split s's partial result."""
    check(
        load().hf_conv2d_nhwc_slabs(
            int(direction), c_void_p(out.data_ptr()), c_void_p(act.data_ptr()), c_void_p(mat.data_ptr()),
            n, h, w, c, k, r, s, stride[0], stride[1], padding[0], padding[1], act_ld, mat_ld, out_c, int(splits),
            out.shape[1], HF_F32, current_stream_ptr(out.device)),
        "hf_conv2d_nhwc_slabs")
    return out
