"""Fused curvature engine: the GGN product ``v -> J^T H_L J v`` of a prepared ResNet-family
model (conv - eval-BatchNorm - ReLU units with residual connections, NHWC fp32) by EXPLICIT
tangent and adjoint sweeps over its layers, issued as direct kernel launches.

Why.  The reference obtains the product from BackPACK's R-op / L-op (optimizer.py:457-462), i.e.
from autograd; so does ``curvature.GGNOperator``.  On an MI355X that product is bound by the
NUMBER of dependent launches (~170 x ~6 us for ResNet-18 on 28x28 inputs), and the convolutions
inside it are MIOpen split-K kernels: a zero-fill launch + a kernel that accumulates with
atomics (not repeatable).  Here every convolution is ONE launch of the package's implicit-GEMM
kernels (hf_conv.hip) whose split-K partial results ("slabs") are summed by the kernel that
CONSUMES them, in its prologue -- the launch boundary publishes them, there is no zero-fill, no
atomic, no in-launch reduction:

    tangent sweep, per unit :  T-conv([t_x | x], [W | v_W]) -> slabs
                               BatchNorm tangent (+ residual tangent, ReLU mask) sums the slabs and
                               writes straight into the next unit's [t_x | x] operand
    adjoint sweep, per unit :  BatchNorm adjoint: g = mask * (sum of the consumers' cotangent slabs),
                               g_a = g * w * rstd, per-channel sums      (hf_chan_affine_bwd_ex)
                               data + weight gradient of the convolution in ONE launch -> slabs
    once per product        :  hf_unpack_tangent_ex (v_W of all layers), hf_maxpool_tangent_nhwc,
                               hf_linear_ce_head (logits' tangent, loss Hessian, the head's three
                               gradients), hf_maxpool_adjoint_nhwc, hf_pack_ex (all parameter
                               gradients, summing the weight-gradient slabs while it gathers);
                               slices of kernel taps that never meet data are skipped throughout

4 launches per conv-BN unit instead of 8 -- 2 where a block's first convolution and its downsample
branch share their launches (grouped kernels) --, 74 per product of ResNet-18; bitwise repeatable.  Under
data parallelism only the entries of the product that can be non-zero are all-reduced (``reduce``).
The layer topology is taken from the prepared model's module tree and from the activations its
patched layers recorded during the step's forward pass (``modelprep`` stores them detached);
anything the engine does not recognise makes ``try_build`` return ``None`` and the caller uses
the autograd operator.  The first product of every model signature is compared with that
operator's product on a random vector.
"""

from .common import _Unsupported, _live_taps, ce_loss_spec, loss_spec_of, mse_loss_spec  # noqa: F401
from .core import FusedGGNEngine  # noqa: F401
from .plain import PlainStackEngine  # noqa: F401
