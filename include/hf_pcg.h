/*
 * hf_pcg.h -- C ABI of the MI355X-native Hessian-free Newton-step solver.
 *
 * This is the drop-in boundary for ONE hot path of ltatzel/PyTorchHessianFree:
 * the preconditioned-CG loop (reference hessianfree/cg.py:9-231) and the vector
 * algebra around the curvature matvec (hessianfree/optimizer.py:266, :288-294,
 * :349-350, :462; hessianfree/preconditioners.py:98, :124-125).  The reference
 * is pure Python and has no FFI; every entry point below names the reference
 * lines whose work it replaces.  INTEGRATION.md shows the ctypes stub a
 * maintainer of the reference would add.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no torch types.
 *   - every `void*` vector is a DEVICE pointer to `n` contiguous elements of
 *     the handle's dtype (HF_F32 / HF_F64), 16-byte aligned unless stated.
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).
 *   - the library BORROWS vectors for the duration of the enqueue; it owns only
 *     its scalar block, partial-sum scratch and a pinned host mirror, all
 *     created in hf_pcg_create.  No allocation, free or synchronisation happens
 *     in any per-iteration entry point (they are hipGraph-capturable).
 *   - return value: 0 (HF_OK) or a negative HF_ERR_* / a positive hipError_t.
 */
#ifndef HF_PCG_H
#define HF_PCG_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HF_ABI_VERSION 13

enum hf_dtype { HF_F32 = 0, HF_F64 = 1 };

/* Termination reasons, in the order the reference tests them (cg.py:95-115). */
enum hf_reason {
  HF_RUNNING = 0,
  HF_REASON_MARTENS = 1,  /* "Convergence (Martens)"     cg.py:103 */
  HF_REASON_MAXITER = 2,  /* "Number of iterations"      cg.py:107 */
  HF_REASON_DIVERGED = 3, /* "Divergence"                cg.py:111 */
  HF_REASON_TOL = 4       /* "Convergence (tolerances)"  cg.py:115 */
};

/* How y = M(r) is obtained (cg.py:190, :220). */
enum hf_precond {
  HF_M_NONE = 0,    /* y = r                                   (M is None)            */
  HF_M_DIAG = 1,    /* y = minv * r, minv = (diag+lambda)^-a   preconditioners.py:124 */
  HF_M_EXTERNAL = 2 /* y supplied by the caller after every residual update          */
};

enum hf_error {
  HF_OK = 0,
  HF_ERR_ARG = -1,      /* null / negative / inconsistent argument */
  HF_ERR_ALIGN = -2,    /* a vector pointer is not 16-byte aligned */
  HF_ERR_STATE = -3,    /* call order violated (e.g. iterate before begin) */
  HF_ERR_NOSYMBOL = -4, /* RCCL symbol not resolvable in this process */
  HF_ERR_CAPACITY = -5  /* too many tensors / snapshots for one call */
};

typedef struct hf_pcg hf_pcg_t; /* opaque solver handle */

typedef struct hf_pcg_status {
  int32_t done;          /* 0 while running, else an hf_reason              */
  int32_t reason;        /* same as done                                    */
  int64_t n_iters;       /* iterations performed (len(x_iters)-1, optimizer.py:276) */
  int64_t iter_next;     /* next iteration the device would run             */
  int64_t nonpos_count;  /* how often pAp <= 0 was seen (cg.py:133-139)     */
  double last_alpha, last_beta, last_pAp, last_res_norm, res_bound;
  int64_t n_stored;      /* snapshot slots written so far                   */
} hf_pcg_status;

int hf_abi_version(void);
const char* hf_error_string(int code);

/* ---- solver lifetime ------------------------------------------------------ */
/* max_blocks: grid size of the vector kernels (0 = default: 2 per CU). */
int hf_pcg_create(hf_pcg_t** out, int64_t n, int dtype, int max_blocks);
int hf_pcg_destroy(hf_pcg_t* h);

/*
 * Begin a solve (cg.py:75-76, :177-192).  On entry `x` holds x0 (zeros when the
 * reference would be called with x0=None).  `store_iters` is a DEVICE int64
 * array, sorted ascending, of the iterations whose iterate must be copied into
 * `slab + slot*slab_stride` (slot = position in the array); may be NULL/0.
 * `store_x0` must be 1 iff store_iters[0] == 0 (x0 itself is a snapshot).
 * `m_hist` is a DEVICE array of max_iter+1 elements (NULL iff !martens).
 * atol < 0 means "no atol" (cg.py:76).
 */
int hf_pcg_begin(hf_pcg_t* h, void* x, void* r, void* p, const void* b,
                 const void* minv, int precond, int64_t max_iter, double tol,
                 double atol, int martens, const int64_t* store_iters,
                 int64_t n_store, int store_x0, void* slab, int64_t slab_stride,
                 void* m_hist);

/* r = A x0 - b, snapshot of x0, partial sums for m_0 and ||b|| (cg.py:187-189).
 * For HF_M_NONE / HF_M_DIAG this also sets p = -M r and ry (cg.py:190-192). */
int hf_pcg_init(hf_pcg_t* h, const void* Ax0, void* stream);
/* HF_M_EXTERNAL only: p = -y, ry = r.y with the caller's y = M(r). */
int hf_pcg_init_external(hf_pcg_t* h, const void* y, void* stream);

/*
 * One PCG iteration for HF_M_NONE / HF_M_DIAG (cg.py:205-224), three kernels:
 *   K1 curvature  : pAp = p.(Bp + damping*p)                (optimizer.py:266, cg.py:206)
 *   K2 update_xr  : alpha, x += alpha p, r += alpha Ap, snapshot,
 *                   partials of r.y, r.r, (r-b).x           (cg.py:207-211, :93, :97, :221)
 *   K3 update_p   : ||r||, m_i, Martens / max-iter / NaN / tol tests on device,
 *                   beta, p = -y + beta p                   (cg.py:95-115, :222-224)
 * `Bp` is the UNDAMPED curvature product B p (damping is added in-kernel); pass
 * damping = 0 when `Bp` already is A p.  After termination the kernels are
 * no-ops, so a host that enqueued speculative iterations still gets the
 * reference's final iterate.
 */
int hf_pcg_iterate(hf_pcg_t* h, const void* Bp, double damping, void* stream);

/* The same three phases individually (HF_M_EXTERNAL needs y = M(r) between the
 * residual update and the direction update). */
int hf_pcg_curvature(hf_pcg_t* h, const void* Bp, double damping, void* stream);
int hf_pcg_update_xr(hf_pcg_t* h, const void* Bp, double damping, void* stream);
int hf_pcg_update_p(hf_pcg_t* h, const void* y_external, void* stream);

/*
 * ONE hipGraph launch per PCG iteration (cg.py:200-227 as a single device-side unit):
 *   [curvature product] -> K1 -> K2 -> K3.
 * `product_graph` is a hipGraph_t (as void*) holding the captured curvature product
 * B.p -- the autograd sweeps of optimizer.py:457-462 + hf_pack, reading the solver's `p`
 * and writing `Bp`; it is CLONED, the caller keeps ownership.  NULL builds K1-K3 only
 * (data-parallel runs: product graph -> all-reduce -> this graph).  Call after
 * hf_pcg_begin; `hf_pcg_graph_update` refreshes the kernel arguments after a later
 * hf_pcg_begin (new x / b / slab / m_hist / damping) without re-instantiating.
 * `with_timing` adds a second executable with event-record nodes around K1/K2/K3;
 * `hf_pcg_graph_launch(g, timed=1, ...)` runs that one, `hf_pcg_graph_collect_timing`
 * waits for it and adds its three durations to what hf_pcg_timing_read reports.
 */
typedef struct hf_pcg_graph hf_pcg_graph_t;
int hf_pcg_graph_create(hf_pcg_graph_t** out, hf_pcg_t* h, void* product_graph,
                        const void* Bp, double damping, int with_timing);
int hf_pcg_graph_update(hf_pcg_graph_t* g, const void* Bp, double damping);
int hf_pcg_graph_launch(hf_pcg_graph_t* g, int timed, void* stream);
int hf_pcg_graph_collect_timing(hf_pcg_graph_t* g);
int hf_pcg_graph_destroy(hf_pcg_graph_t* g);

/* Non-blocking: reads the pinned host mirror the device writes at termination
 * (fills done/reason and, once done, n_iters = the terminating iteration). */
int hf_pcg_poll(hf_pcg_t* h, hf_pcg_status* out);
/* Synchronises `stream`, copies the scalar block back, fills `out`. */
int hf_pcg_finish(hf_pcg_t* h, hf_pcg_status* out, void* stream);
/* After finish: the first `cap` non-positive-curvature events (cg.py:136-139). */
int hf_pcg_read_nonpos(hf_pcg_t* h, int64_t* iters, double* values, int cap);

/* Per-kernel HIP-event timing of hf_pcg_iterate (for bench.py's roofline). */
int hf_pcg_timing_enable(hf_pcg_t* h, int enable);
/* After finish: mean duration in ms of K1,K2,K3 over the recorded iterations. */
int hf_pcg_timing_read(hf_pcg_t* h, double* ms_k1, double* ms_k2, double* ms_k3,
                       int64_t* n_recorded);

/* ---- vector helpers around the matvec -------------------------------------- */
/*
 * Multi-tensor gather: dst[off_t + i] = scale * src_t[i] for t < n_tensors
 * (replaces parameters_to_vector's torch.cat, optimizer.py:234, :455, :462).
 * `srcs` / `numels` are HOST arrays; pointers are passed to the kernel by value.
 * src tensors must be contiguous; only 4-byte (8 for f64) alignment is needed.
 * mode 0: dst = scale*src ; mode 1: dst += (scale*src)^2 (preconditioners.py:98).
 * `perm` (HOST, 2 per tensor, or NULL): {I, H*W} for a 4-D tensor [O, I, H, W] that is
 * stored channels-last, i.e. as (O, H, W, I) -- what MIOpen's NHWC weight-gradient
 * kernels produce; {0, 0} for plain contiguous.  The gather writes the reference's
 * (O, I, H, W) order either way.
 */
int hf_pack(void* dst, const void* const* srcs, const int64_t* numels,
            const int64_t* perm, int n_tensors, double scale, int mode, int dtype,
            void* stream);
/* The same with `splits` (HOST, 2 per tensor, or NULL): {count, stride in elements} when source t
 * is the sum of `count` split-K slabs (hf_conv2d_nhwc_*_slabs weight gradients), and `live` (HOST,
 * 1 per tensor, or NULL): for a permuted source with H*W <= 16, bit hw set = kernel tap hw can
 * meet data; the gradients of the other taps (3x3 kernels on 1x1 / 2x2 maps) are structurally
 * zero and are written as zeros without being read.  0 = read everything. */
int hf_pack_ex(void* dst, const void* const* srcs, const int64_t* numels, const int64_t* perm,
               const int64_t* splits, const int64_t* live, int n_tensors, double scale, int mode,
               int dtype, void* stream);
/*
 * Multi-tensor scatter for the tangent sweep, the counterpart of hf_pack: tensor t is the
 * contiguous [O, slab] block at src + src_offs[t] (a weight-shaped slice of the CG
 * vector, slab = I*H*W).  It is written into the SECOND half of the input-channel axis of
 * dsts[t], a [O, 2I, H, W] buffer -- the weight operand [W | v_W] of the single
 * convolution conv([v_x | x], [W | v_W]) that evaluates a conv layer's tangent map
 * inside BackPACK's R-op (optimizer.py:461):
 *   inners[t] == 0 : dst stored (O, 2I, H, W):  dst[o*2*slab + slab + r]      = src[o*slab + r]
 *   inners[t] == I : dst stored (O, H, W, 2I):  dst[(o*HW + hw)*2I + I + i]   = src[(o*I + i)*HW + hw]
 * One launch instead of one strided copy per conv layer and product.  All arrays HOST.
 */
int hf_unpack_tangent(const void* src, void* const* dsts, const int64_t* src_offs,
                      const int64_t* numels, const int64_t* slabs, const int64_t* inners,
                      int n_tensors, int dtype, void* stream);
/* The same with `live` (HOST, 1 per tensor, or NULL; NHWC destinations with H*W <= 16): only the
 * slices of the taps whose bit is set are copied -- the convolution kernels never read the
 * others (hf_conv2d_nhwc drops taps that only meet padding).  0 = copy everything. */
int hf_unpack_tangent_ex(const void* src, void* const* dsts, const int64_t* src_offs,
                         const int64_t* numels, const int64_t* slabs, const int64_t* inners,
                         const int64_t* live, int n_tensors, int dtype, void* stream);
/* The same scatter with a per-tensor choice of the half it fills (`halves`, HOST, or NULL = all 1):
 * 1 = the v_W half (above), 0 = the W half -- dst[(o*HW + hw)*2I + i] resp. dst[o*2*slab + r]; 2 = a
 * DENSE transposed copy dst[(i*HW + hw)*O + o] (inners[t] = I required): the (I, H, W, O) operand of the
 * data-gradient convolutions -- the weights once per step, and the vector's weight slices V once per
 * Hessian product (the term V^T g of the R-op of the backward pass, optimizer.py:450-455).  The
 * persistent curvature engine refreshes the W halves of all [W | v_W] operands from the flat
 * parameter vector with ONE launch per trial point theta0 + alpha*step (optimizer.py:288-294), where
 * the reference re-binds every parameter (utils.py:8-38) and PyTorch re-converts every NCHW weight. */
int hf_unpack_weights(const void* src, void* const* dsts, const int64_t* src_offs,
                      const int64_t* numels, const int64_t* slabs, const int64_t* inners,
                      const int64_t* live, const int64_t* halves, int n_tensors, int dtype, void* stream);

/*
 * Data-parallel products (the `result += N * mb_result` of optimizer.py:677-684, across GPUs): the
 * entries of a curvature product that are structurally zero on EVERY rank -- conv-weight slices of
 * kernel taps that never meet data -- need not travel.  hf_live_copy gathers the other entries
 * of the flat vector `full` into `compact` (scatter = 0) or writes them back (scatter = 1); the
 * all-reduce runs on `compact`.  The vector is described as n_segments <= 24 consecutive segments
 * (HOST arrays): full_offs[i] = start in the full vector, counts[i] = elements IN THE FULL VECTOR,
 * periods[i] = 0 for a dense segment, else H*W <= 16 of a weight [O, I, H, W] whose live taps are
 * the bits of masks[i].  Compact order = segment order, inside a masked segment (o, i, live tap).
 */
int hf_live_copy(void* full, void* compact, int scatter, const int64_t* full_offs, const int64_t* counts,
                 const int64_t* periods, const int64_t* masks, int n_segments, int dtype, void* stream);

/* minv = (diag + damping)^(-exponent)   (preconditioners.py:124, hoisted out of
 * the CG loop). */
int hf_precond_build(void* minv, const void* diag, double damping,
                     double exponent, int64_t n, int dtype, void* stream);

/* out = a + alpha*s  (params_vec + step / params_vec + lr*step_vec,
 * optimizer.py:293, :349); out may alias a.  No alignment requirement. */
int hf_axpy_out(void* out, const void* a, const void* s, double alpha, int64_t n,
                int dtype, void* stream);

/* ---- eval-mode BatchNorm (+ residual add, + ReLU) inside the curvature product -- */
/*
 * An eval-mode BatchNorm is y = xhat*w[c] + b[c], xhat = (x - mean[c])*rstd[c] on an
 * [n, c, hw] tensor stored NCHW (channels_last = 0) or NHWC (channels_last = 1, the layout
 * MIOpen's implicit-GEMM kernels run in without transposes); ResNet blocks follow it by "+ identity" and/or
 * ReLU.  PyTorch's generic double-backward of batch_norm (which BackPACK's R-op, and
 * ours, differentiates through on every GGN product, optimizer.py:461) issues ~16
 * small kernels per layer, plus 2 per add and 2 per ReLU; these two entry points are
 * the whole group in one launch each.
 *   hf_chan_affine    : t   = a*(w*rstd) + xhat*q + r + add      (a,q,r,add,w nullable; rstd == NULL
 *                       means 1: a convolution + bias layer without BatchNorm, then q must be NULL)
 *                       out = relu_self ? max(t,0) : mask_src ? (mask_src>0 ? t : 0) : t
 *   hf_chan_affine_bwd: g = mask_src ? (gy+gy2)*(mask_src>0) : gy+gy2 ;
 *                       gx = g*w*rstd, gw = sum_{n,hw} g*xhat, gb = sum g, gres = g
 *                       (gx, gw, gb, gres nullable; with gw == NULL, x/mean may be NULL,
 *                       and with gx == NULL too rstd may be NULL: gb alone is the bias
 *                       gradient of a convolution layer, sum_{n,hw} gy -- deterministic and
 *                       safe inside a hipGraph, which PyTorch's multi-block reduction of an
 *                       NHWC tensor was measured not to be on this stack)
 */
/* out_ld / add_ld (elements, 0 = dense): `out` resp. `add` is the slice of the first c channels
 * of a buffer with more channels -- NHWC: element (row, ch) at row*ld + ch; NCHW: (n, ch, hw)
 * at n*ld + ch*hw_count + hw.  The tangent sweep uses it to write a layer's output tangent
 * straight into the next convolution's [v_x | x] operand (and to read it from there for
 * the residual branch) instead of copying it. */
int hf_chan_affine(void* out, const void* a, const void* x, const void* mean,
                   const void* rstd, const void* w, const void* q, const void* r,
                   const void* add, const void* mask_src, int relu_self, int64_t n,
                   int64_t c, int64_t hw, int channels_last, int64_t out_ld, int64_t add_ld,
                   int dtype, void* stream);
/* hf_chan_affine with operand `a` given as `a_splits` split-K slabs `a_slab` elements apart
 * (the tangent convolution's partial results, summed here in split order). */
int hf_chan_affine_ex(void* out, const void* a, const void* x, const void* mean, const void* rstd,
                      const void* w, const void* q, const void* r, const void* add,
                      const void* mask_src, int relu_self, int64_t n, int64_t c, int64_t hw,
                      int channels_last, int64_t out_ld, int64_t add_ld, int a_splits, int64_t a_slab,
                      int dtype, void* stream);
/* hf_chan_affine_bwd with both cotangents given as split-K slabs (data gradients of
 * hf_conv2d_nhwc_backward_slabs), summed in split order while they are loaded; NHWC (or hw == 1).
 * row_blocks > 1 (NHWC fp32, c % 4 == 0, c <= 1024): a row-major kernel -- `row_blocks`
 * workgroups each take a share of the rows and ALL channels, reading whole contiguous rows
 * (the default kernel gives each workgroup one 16-byte channel column of all rows: one cache
 * line per lane) -- and gw / gb receive `row_blocks` partial sums each, c elements apart, for
 * hf_pack_ex to add up.  ceil(rows / row_blocks) rows per workgroup; no empty workgroup allowed. */
int hf_chan_affine_bwd_ex(void* gx, void* gw, void* gb, void* gres, const void* gy, int gy_splits,
                          int64_t gy_slab, const void* gy2, int gy2_splits, int64_t gy2_slab,
                          const void* x, const void* mean, const void* rstd, const void* w,
                          const void* mask_src, int64_t n, int64_t c, int64_t hw, int channels_last,
                          int row_blocks, int dtype, void* stream);
/* TWO independent layers in one launch (a residual block's first BatchNorm and its downsample
 * branch's, whose inputs come out of one grouped convolution launch): the arguments of
 * hf_chan_affine_ex resp. hf_chan_affine_bwd_ex (row_blocks >= 2) per problem; NHWC fp32 only. */
typedef struct hf_affine_problem {
  void* out;
  const void *a, *x, *mean, *rstd, *w, *q, *r, *add, *mask_src;
  int relu_self;
  int64_t n, c, hw, out_ld, add_ld;
  int a_splits;
  int64_t a_slab;
} hf_affine_problem;
int hf_chan_affine_pair(const hf_affine_problem* problems /* [2] */, int dtype, void* stream);
typedef struct hf_bn_adjoint_problem {
  void *gx, *gw, *gb, *gres;
  const void* gy;
  int gy_splits;
  int64_t gy_slab;
  const void* gy2;
  int gy2_splits;
  int64_t gy2_slab;
  const void *x, *mean, *rstd, *w, *mask_src;
  int64_t n, c, hw;
  int row_blocks;
} hf_bn_adjoint_problem;
int hf_chan_affine_bwd_pair(const hf_bn_adjoint_problem* problems /* [2] */, int dtype, void* stream);

/*
 * Forward pass of conv -> (eval-BatchNorm | bias) (+ residual) (+ ReLU) from the convolution's split-K
 * slabs, NHWC fp32 [rows = n*hw, c] -- what the reference evaluates by `forward()` for the loss
 * and for every trial point of LM damping / CG-backtracking / line search (optimizer.py:216-229,
 * :288-294); the persistent curvature engine replays it on its own static buffers:
 *   s     = sum over `splits` slabs of a (split order);         a_out = s (nullable)
 *   t     = rstd ? ((s - mean[c])*rstd[c])*w[c] : s;  t += b[c] (nullable);  t += res (nullable)
 *   y     = relu ? max(t, 0) : t   -> y (dense, nullable) and y2 (nullable; pixel stride y2_ld >= c:
 *           the x half of the consumer's [t_x | x] operand)
 * res_ld: pixel stride of res (0 = dense).  Same rounding sequence as hf_chan_affine's forward.
 */
int hf_bn_forward(void* y, void* y2, int64_t y2_ld, void* a_out, const void* a, int splits, int64_t slab_stride,
                  const void* mean, const void* rstd, const void* w, const void* b, const void* res,
                  int64_t res_ld, int relu, int64_t rows, int64_t c, int dtype, void* stream);

/*
 * TRAIN-mode BatchNorm inside the curvature product (what examples/run_resnet18_mnist.py runs: no
 * model.eval(), so BackPACK's R-op / L-op differentiate through the batch statistics, optimizer.py:461).
 * Tangent and adjoint of xhat = (a - mean_B(a)) * rstd_B are the same operator
 *     xhat' = rstd * [a' - mean(a') - xhat * mean(xhat * a')],
 * whose two per-channel corrections fold into the per-channel vectors of hf_chan_affine_ex
 * (t = a*(w*rstd) + xhat*q + r):   q = vq - w*rstd*S_x/count,   r = vr - w*rstd*S_1/count,
 * S_x = sum(xhat*a'), S_1 = sum(a') over the batch.  The finalisation runs in the PROLOGUE of the elementwise launch
 * (fp32 NHWC, c % 4 == 0, c <= 1024, 16-byte aligned operands): every workgroup adds the `nparts` partial rows of
 * S_x (`part_x`) and S_1 (`part_1`) -- what hf_chan_affine_bwd_ex wrote with gx = NULL, or the tangent convolution's
 * own epilogue (hf_conv2d_nhwc_group_slabs_bnsum) -- up in the same fixed order, forms q and r (vq / vr nullable: the
 * adjoint has none) and applies  out = mask_src > 0 ? t : 0,  t = sum(a slabs)*(w*rstd) + xhat*q + r + add.
 * (Round 4 also shipped the finalisation as a launch of its own, as the reduction launch's last workgroup and as one
 * launch around a grid barrier; measured slower -- profiles/r04_train_bn_forms.jsonl -- and removed in ABI v11.)
 */
int hf_chan_affine_train(void* out, const void* a, const void* x, const void* mean, const void* rstd, const void* w,
                         const void* part_x, const void* part_1, int nparts, const void* vq, const void* vr,
                         double count, const void* add, const void* mask_src, int64_t n, int64_t c, int64_t hw,
                         int64_t out_ld, int64_t add_ld, int a_splits, int64_t a_slab, int dtype, void* stream);
/* Two independent layers (a residual block's first BatchNorm and its downsample branch's; no residual operand) in ONE
 * launch: each problem is what hf_chan_affine_train takes. */
typedef struct hf_affine_train_problem {
  void* out;
  const void *a, *x, *mean, *rstd, *w, *part_x, *part_1;
  int nparts;
  const void *vq, *vr;
  double count;
  const void *add, *mask_src; /* add must be NULL */
  int64_t n, c, hw, out_ld;
  int a_splits;
  int64_t a_slab;
} hf_affine_train_problem;
int hf_chan_affine_train_pair(const hf_affine_train_problem* problems /* [2] */, int dtype, void* stream);

/*
 * One-pass batch statistics of a train-mode BatchNorm in the engine's own forward pass (optimizer.py:216-229,
 * :288-294 on a model in train mode): a_out (nullable) = sum of `splits` slabs of a (split order), per-channel
 * sum a and sum a^2 in fp64 per row block -> `part`: [row_blocks, 2, c] doubles.  hf_bn_forward_train finalises them.
 */
int hf_bn_stats_rows(void* a_out, const void* a, int splits, int64_t slab_stride, void* part, int64_t rows, int64_t c,
                     int row_blocks, int dtype, void* stream);

/*
 * The forward of a train-mode BatchNorm (+ residual, + ReLU) with the statistics' finalisation in its PROLOGUE:
 * hf_bn_stats_rows leaves its partial rows (`part`: [nparts][2][c] doubles); every workgroup of this launch adds them
 * up (fixed order), forms mean, biased variance = E[a^2] - mean^2 (fp64) and rstd = 1/sqrt(var + eps), applies  y = act(((a - mean)*rstd)*w + b + res)  to its share (`a`: the summed convolution output), and
 * workgroup 0 writes mean / rstd and moves the running statistics as torch.nn.BatchNorm2d's
 * forward does (momentum < 0: not): r <- (1 - momentum) r + momentum * {mean, var * count/(count - 1)}.  fp32 NHWC, c % 4 == 0.
 */
int hf_bn_forward_train(void* y, void* y2, int64_t y2_ld, const void* a, const void* part, int nparts, void* mean,
                        void* rstd, void* running_mean, void* running_var, double count, double eps, double momentum,
                        const void* w, const void* b, const void* res, int64_t res_ld, int relu, int64_t rows,
                        int64_t c, int dtype, void* stream);

/*
 * Hessian product (optimizer.py:450-455, forward-over-reverse) through a TRAIN-mode BatchNorm
 * z = gamma*xhat + beta, xhat = (a - mean(a))*rstd(a) -- the model of examples/run_resnet18_mnist.py:19-35 with
 * curvature_opt="hessian".  With the step's first-order cotangents g_z (masked), g_a, the tangent sweep's a' and the
 * second-order masked cotangent g_z' (m = rows; S1 = mean(a'), Sx = mean(a' xhat)):
 *   g_gamma' = sum(g_z' xhat) + rstd sum(g_z a') - rstd (S1 g_beta + Sx g_gamma)
 *   g_a'     = c0 g_a + c1 g_z + c2 g_z' + c3 a' + c4 xhat + c5
 * hf_bn_train_hessian_coeffs adds up the partial rows ([rows][c] floats each, fp64 sums in row order) of
 * sum(g_z' xhat), sum(g_z'), rstd sum(g_z a') (`nparts` rows: what hf_chan_affine_bwd_ex leaves) and of sum(a' xhat),
 * sum(a') (`nparts_t` rows: hf_chan_affine_bwd_ex's, or the tangent convolution's epilogue sums) and
 * writes coef ([6][c]) and the closed-form share of g_gamma' (gw_corr, [c]: one more partial row for hf_pack_ex);
 * g_gamma1 / g_beta1: the first-order parameter gradients of the layer.  hf_bn_train_hessian_apply is the elementwise
 * pass; a' arrives as the tangent convolution's `t_splits` split-K slabs.  fp32 NHWC, c % 4 == 0.
 */
int hf_bn_train_hessian_coeffs(void* coef, void* gw_corr, const void* sum_gx2, const void* sum_g2, const void* sum_ga,
                               int nparts, const void* sum_tx, const void* sum_t1, int nparts_t, const void* g_gamma1,
                               const void* g_beta1, const void* gamma, const void* v_gamma, const void* rstd,
                               double count, int64_t c, int dtype, void* stream);
int hf_bn_train_hessian_apply(void* out, const void* ga1, const void* gz1, const void* gz2, const void* t, int t_splits,
                              int64_t t_slab, const void* a, const void* mean, const void* rstd, const void* coef,
                              int64_t rows, int64_t c, int dtype, void* stream);

/* Elementwise adjoint pre-pass of a fused BatchNorm(+add+ReLU) layer, NHWC [rows, c]:
 *   g = (sum of gy_a's slabs + sum of gy_b's slabs) * [mask_src > 0];  g_out = g (nullable);
 *   ga_out = g * w[c]*rstd[c] (nullable): the cotangent of the convolution output that
 *   hf_conv2d_nhwc_backward* then consumes; the per-channel sums (hf_chan_affine_bwd with
 *   gx = NULL) are taken of g_out.  gy_b, mask_src, w nullable. */
int hf_bn_adjoint_pre(void* g_out, void* ga_out, const void* gy_a, int a_splits, int64_t a_slab,
                      const void* gy_b, int b_splits, int64_t b_slab, const void* mask_src,
                      const void* w, const void* rstd, int64_t rows, int64_t c, int dtype,
                      void* stream);
/* gy2 (nullable): a second cotangent, added to gy first -- the output of a residual block has
 * two consumers (the next block's first convolution and its identity / downsample branch);
 * handing their cotangents over separately saves the addition autograd would issue. */
int hf_chan_affine_bwd(void* gx, void* gw, void* gb, void* gres, const void* gy,
                       const void* gy2, const void* x, const void* mean, const void* rstd,
                       const void* w, const void* mask_src, int64_t n, int64_t c, int64_t hw,
                       int channels_last, int dtype, void* stream);

/* ---- convolution layers on small feature maps inside the curvature product -------- */
/*
 * Implicit-GEMM convolution, fp32 MFMA, NHWC, DETERMINISTIC split-K (one launch: every
 * K-split writes its partial tile to `workspace`, the last workgroup to arrive at a tile
 * sums them in split order -- no zero-fill launch, no float atomics).  These are the three
 * convolutions BackPACK's R-op / L-op (optimizer.py:461) run per layer and product through
 * PyTorch -> MIOpen, whose split-K kernels cost one extra launch each and are not bitwise
 * repeatable.  Kernel taps that only ever meet padding (3x3 windows on 1x1 / 2x2 maps) are
 * skipped.
 *   direction 0 (F): out[n,oh,ow,k] = sum act[n, oh*s-p+r, ow*s-p+q, c] * mat[k][r][q][c]
 *                    act = X [n,h,w,c] (pixel stride act_ld >= c, 0 = dense), mat = weights (O,H,W,I)
 *   direction 1 (D): out[n,h,w,c]   = sum act[n, (ih+p-r)/s, (iw+p-q)/s, k] * mat[c][r][q][k]
 *                    act = dY [n,oh,ow,k], mat = weights transposed to (I,H,W,O)
 *   direction 2 (W): out[k][r][q][c] = sum_m mat[m][k] * act[pix(m,r,q)][c]
 *                    act = X, mat = dY [n*oh*ow][k]; entries of taps that never meet data are
 *                    NOT written (keep `out` zero-initialised)
 * c and k must be multiples of 4; all pointers 16-byte aligned.  `tickets`: n_tickets zeroed
 * ints (self-resetting); `workspace`: scratch for the partial tiles -- both may be shared by
 * all calls enqueued on ONE stream.  target_blocks <= 0: the number of K-splits comes from a
 * measured cost model; > 0: split until about that many workgroups exist.  HF_F32 only.
 */
int hf_conv2d_nhwc(int direction, void* out, const void* act, const void* mat, int64_t n,
                   int64_t h, int64_t w, int64_t c, int64_t k, int64_t r, int64_t s,
                   int64_t stride_h, int64_t stride_w, int64_t pad_h, int64_t pad_w,
                   int64_t act_ld, void* workspace, int64_t workspace_bytes, void* tickets,
                   int64_t n_tickets, int target_blocks, int dtype, void* stream);

/* hf_conv2d_nhwc_slabs (direction 0) and hf_unpack_weights in ONE launch: the scatter's workgroups follow the
 * convolution's.  For a convolution that reads none of the scattered operands -- the stem's tangent convolution of
 * the GGN product (BackPACK's R-op through the first layer, /root/reference/hessianfree/optimizer.py:457-462) -- while
 * every later layer's [W | v_W] operand must be complete before ITS launch: 14 us of scatter hide behind a 12 us
 * latency-bound launch.  u* arguments as hf_unpack_weights (at most 64 non-empty tensors); fp32, small-map tile
 * configuration only (else HF_ERR_ARG: call the two entry points separately). */
int hf_conv2d_nhwc_slabs_unpack(void* out, const void* act, const void* mat, int64_t n, int64_t h, int64_t w,
                                int64_t c, int64_t k, int64_t r, int64_t s, int64_t stride_h, int64_t stride_w,
                                int64_t pad_h, int64_t pad_w, int64_t act_ld, int64_t mat_ld, int splits,
                                int64_t slab_stride, const void* usrc, void* const* udsts, const int64_t* usrc_offs,
                                const int64_t* unumels, const int64_t* uslabs, const int64_t* uinners,
                                const int64_t* ulive, const int64_t* uhalves, int n_tensors, int dtype,
                                void* stream);

/*
 * Consumer-side reduction ("slab") variants: split s of the reduction writes its partial result,
 * in the output's own layout, to out + s*slab_stride and the launch ends there -- no workspace,
 * tickets or fences; the launch boundary publishes the slabs and the kernel that CONSUMES the
 * tensor sums them, in split order, in its prologue (hf_chan_affine_ex: operand `a`;
 * hf_bn_adjoint_pre: both cotangents; hf_pack_ex: sources).  Measured 6-11 us per launch on the
 * ResNet-18 shapes against 12-20 us with the in-launch reduction.  `hf_conv2d_nhwc_plan`
 * returns the number of splits the launch will use (>= 1; pure host arithmetic), which the
 * caller needs to size the slab buffer; pass it back as `splits`.  For direction 2 every slab
 * must be zero-initialised once if the geometry has taps that never meet data; `out_c` (0 = c)
 * restricts the output to X's first out_c channels, laid out [k][r][q][out_c] -- X may carry
 * zero-padding channels that make its rows 16-byte multiples (the 49-tap im2col of a stem).
 * HF_ERR_ARG (not a silently wrong gradient) when out_c < c meets a geometry that was planned
 * for the 96-row tile configuration whose output columns are enumerated flat over (tap, channel)
 * AND has more than one live tap: that enumeration needs out_c == c unless there is a single tap
 * (where the padded columns are simply cut off).
 * `mat_ld` (directions 0 and 1; 0 = c resp. k): floats between consecutive taps of `mat`, when `mat`
 * is the first-channels slice of a wider [rows][r][q][mat_ld] buffer -- the forward pass of the
 * curvature engine reads the W half of the tangent sweep's [W | v_W] operand in place.
 */
int hf_conv2d_nhwc_plan(int direction, int64_t n, int64_t h, int64_t w, int64_t c, int64_t k,
                        int64_t r, int64_t s, int64_t stride_h, int64_t stride_w, int64_t pad_h,
                        int64_t pad_w, int target_blocks);
int hf_conv2d_nhwc_slabs(int direction, void* out, const void* act, const void* mat, int64_t n,
                         int64_t h, int64_t w, int64_t c, int64_t k, int64_t r, int64_t s,
                         int64_t stride_h, int64_t stride_w, int64_t pad_h, int64_t pad_w,
                         int64_t act_ld, int64_t mat_ld, int64_t out_c, int splits, int64_t slab_stride,
                         int dtype, void* stream);
int hf_conv2d_nhwc_backward_slabs(void* dx, void* dw, const void* dy, const void* x, const void* w_t,
                                  int64_t n, int64_t h, int64_t w, int64_t c, int64_t k, int64_t r,
                                  int64_t s, int64_t stride_h, int64_t stride_w, int64_t pad_h,
                                  int64_t pad_w, int splits_d, int64_t slab_stride_d, int splits_w,
                                  int64_t slab_stride_w, int dtype, void* stream);

/* Up to 4 independent slab-mode convolutions in ONE launch (any mix of directions; 16-byte gather
 * variant only: channel counts multiples of 4).  Each problem is what hf_conv2d_nhwc_slabs takes.
 * The curvature engine groups the two tangent convolutions that read a residual block's input (its
 * first convolution and its downsample branch), and the data + weight gradients of both layers. */
typedef struct hf_conv_problem {
  int direction;            /* 0 forward / tangent, 1 data gradient, 2 weight gradient */
  void* out;
  const void* act;
  const void* mat;
  int64_t n, h, w, c, k, r, s, stride_h, stride_w, pad_h, pad_w;
  int64_t act_ld, out_c;
  int splits;
  int64_t slab_stride;
  int64_t mat_ld;           /* see hf_conv2d_nhwc_slabs; 0 = dense */
} hf_conv_problem;
int hf_conv2d_nhwc_group_slabs(const hf_conv_problem* problems, int n_problems, int dtype, void* stream);
/* Tangent convolutions in front of TRAIN-mode BatchNorm layers (optimizer.py:457-462 with a model in train mode,
 * examples/run_resnet18_mnist.py:19-35): hf_conv2d_nhwc_group_slabs for direction-0 problems whose epilogue ALSO
 * writes, per problem with part_1 != NULL, the per-channel partial sums of every (row tile, split)'s output tile,
 *   part_1[(tile_m*splits + split)][ch] = sum_rows t,   part_x[...] = sum_rows t * (x - mean[ch]) * rstd[ch]
 * (x: the layer's recorded convolution output [rows][k]; part_rows = ceil(rows / 64) * splits rows of k floats) --
 * what the reduction launch between convolution and elementwise pass computed; hf_chan_affine_train adds the rows
 * up.  64x64-tile problems only (HF_ERR_ARG otherwise: the caller falls back to the two launches). */
typedef struct hf_conv_bnsum {
  const void *x, *mean, *rstd;
  void *part_x, *part_1;
  int64_t part_rows;
} hf_conv_bnsum;
int hf_conv2d_nhwc_group_slabs_bnsum(const hf_conv_problem* problems, int n_problems, const hf_conv_bnsum* sums,
                                     int dtype, void* stream);
/* ONE forward / data-gradient problem (`d`, direction 0 or 1) and ONE weight-gradient problem (`w`, direction 2)
 * in one launch, each described like a grouped problem (so either may read an operand that is a channel slice
 * of a wider buffer: act_ld / mat_ld).  hf_conv2d_nhwc_backward_slabs is the special case "same layer, dense
 * operands".  The Hessian sweep of the curvature engine issues, per layer, this launch twice: the GGN's pair
 * conv_D(g', W), conv_W(x, g') and the pair that carries the network's own curvature, conv_D(g, V),
 * conv_W(t_x, g) with t_x read in place from the [t_x | x] operand (optimizer.py:450-455). */
int hf_conv2d_nhwc_dw_slabs(const hf_conv_problem* d, const hf_conv_problem* w, int dtype, void* stream);

/* Data gradient AND weight gradient of one layer (directions 1 and 2 above) in ONE launch:
 * both read dY [n,oh,ow,k], neither depends on the other.  dx [n,h,w,c]; dw [k][r][q][c]
 * (dead taps not written); x [n,h,w,c]; w_t = weights stored (I,H,W,O). */
int hf_conv2d_nhwc_backward(void* dx, void* dw, const void* dy, const void* x, const void* w_t,
                            int64_t n, int64_t h, int64_t w, int64_t c, int64_t k, int64_t r,
                            int64_t s, int64_t stride_h, int64_t stride_w, int64_t pad_h,
                            int64_t pad_w, void* workspace, int64_t workspace_bytes, void* tickets,
                            int64_t n_tickets, int target_blocks, int dtype, void* stream);

/* ---- loss Hessian inside the GGN product ------------------------------------ */
/*
 * out[r, :] = scale * p[r, :] * (v[r, :] - <p[r, :], v[r, :]>),  p = softmax(logits) row-wise:
 * the Hessian of a softmax cross-entropy w.r.t. the logits applied to v = J v (scale = 1/B
 * for reduction "mean").  It is the middle factor of J^T H_L J v, which BackPACK's
 * ggn_vector_product_from_plist (optimizer.py:461) obtains by differentiating the loss
 * twice (~14 small kernels per product); the host verifies this closed form against that
 * autograd sweep once per operator before using it.  [rows, cols] row-major, contiguous.
 */
int hf_softmax_ce_hvp(void* out, const void* p, const void* v, double scale, int64_t rows,
                      int64_t cols, int dtype, void* stream);

/* ---- max-pool and classifier head inside the GGN product (hf_head.hip) ----------- */
/*
 * The remaining non-convolution stages of BackPACK's R-op / L-op sweeps through a ResNet
 * (optimizer.py:461), each as ONE launch with a fixed summation order; fp32, NHWC.
 *
 * hf_maxpool_tangent_nhwc: out[n,oy,ox,c] = t[n, idx[n,oy,ox,c], c] -- the tangent of a max-pool
 * is the tangent at the window's maximum.  idx: int32 [n,oh,ow,c], flat position y*w + x inside
 * the (n, c) plane (max_pool2d_with_indices, taken once per Newton step).  out may be the
 * first-channels slice of a wider NHWC buffer (out_ld floats per pixel, 0 = dense).
 * Replaced: a gather plus a strided copy.
 */
int hf_maxpool_tangent_nhwc(void* out, const void* t, const void* idx, int64_t n, int64_t h, int64_t w,
                            int64_t oh, int64_t ow, int64_t c, int64_t out_ld, int dtype, void* stream);
/*
 * hf_maxpool_forward_nhwc: out[n,oy,ox,c] = max over the window, idx[n,oy,ox,c] = its flat position
 * y*w + x (first maximum in window scan order, NaN wins -- ATen's max_pool2d_with_indices rule).
 * out (dense, nullable) and out2 (nullable, pixel stride out2_ld) receive the same values.  The
 * persistent curvature engine's own forward pass (positions are taken on every forward instead of
 * once per step by ATen).
 */
int hf_maxpool_forward_nhwc(void* out, void* out2, int64_t out2_ld, void* idx, const void* x, int64_t n,
                            int64_t h, int64_t w, int64_t oh, int64_t ow, int64_t c, int64_t kh, int64_t kw,
                            int64_t stride_h, int64_t stride_w, int64_t pad_h, int64_t pad_w, int dtype,
                            void* stream);
/*
 * hf_maxpool_adjoint_nhwc: g[n,y,x,c] = sum over the windows whose maximum sits at (y,x) of
 * (gy_a + gy_b)[n,oy,ox,c], gy_a / gy_b the cotangents from the pooled map's two consumers, each
 * `splits` slabs `slab` elements apart that are summed in split order (gy_b nullable).  Gather
 * form: no zero-fill, no atomics.  Replaced: the slab sum, a zero-fill and ATen's
 * max_pool2d_with_indices_backward (atomic accumulation in NHWC).
 */
int hf_maxpool_adjoint_nhwc(void* g, const void* gy_a, int a_splits, int64_t a_slab, const void* gy_b,
                            int b_splits, int64_t b_slab, const void* idx, int64_t n, int64_t h, int64_t w,
                            int64_t oh, int64_t ow, int64_t c, int64_t kh, int64_t kw, int64_t stride_h,
                            int64_t stride_w, int64_t pad_h, int64_t pad_w, int dtype, void* stream);
/*
 * hf_linear_ce_head: the classifier head of J^T H_L J v in one launch (one workgroup per 4 rows):
 *   Jv = t_feat W^T + feat V_W^T + v_b;  HJv = scale * p * (Jv - <p, Jv>) (as hf_softmax_ce_hvp);
 *   g_feat = HJv W [rows, features];  g_w = HJv^T feat [classes, features];  g_b = sum_rows HJv.
 * t_feat / feat [rows, features]: tangent and value of the features; w / v_w [classes, features];
 * v_b, g_b nullable; p = softmax(logits) [rows, classes].  g_w and g_b are written as
 * hf_linear_ce_head_slabs(rows) PARTIAL sums (one per workgroup, classes*features resp. classes
 * elements apart) that hf_pack_ex adds up (`splits`), like split-K weight gradients.  Small heads
 * only: classes <= 64, features <= 512 and a multiple of 4, (2*classes + 4)*features floats within
 * 64 KB of LDS -- returns -1 otherwise (the caller keeps the GEMM path).  Replaced: 4 rocBLAS
 * GEMMs, a reduction and hf_softmax_ce_hvp.
 */
int hf_linear_ce_head(void* g_feat, void* g_w, void* g_b, const void* t_feat, const void* feat, const void* w,
                      const void* v_w, const void* v_b, const void* p, double scale, int64_t rows,
                      int64_t features, int64_t classes, int dtype, void* stream);
int hf_linear_ce_head_slabs(int64_t rows);
/*
 * hf_pool_ce_head: the head of a network that ends in global average pooling (All-CNN-C,
 * examples/example_utils.py:59-83: logits = mean over the last map), inside J^T H_L J v, one launch:
 *   Jv[n,k] = mean_hw t[n,hw,k];  HJv = scale * p * (Jv - <p, Jv>);  g[n,hw,k] = HJv[n,k] / hw
 * t, g NHWC [n, hw, k]; p = softmax(logits) [n, k].  jv_out (nullable) receives Jv.  One workgroup per
 * sample.  Replaced: a mean reduction, hf_softmax_ce_hvp, a division and a broadcast copy.
 */
int hf_pool_ce_head(void* g, void* jv_out, const void* t, const void* p, double scale, int64_t n, int64_t hw,
                    int64_t k, int dtype, void* stream);

/* ---- RCCL (resolved at run time from the already-loaded librccl) ----------- */
typedef struct hf_comm hf_comm_t;
int hf_comm_unique_id(char* out128);                        /* ncclGetUniqueId */
int hf_comm_create(hf_comm_t** out, const char* id128, int nranks, int rank);
int hf_comm_destroy(hf_comm_t* c);
/* In-place sum all-reduce of the GGN.v partial (the `+=` of optimizer.py:677-684
 * across ranks), enqueued on `stream`. */
int hf_allreduce_sum(hf_comm_t* c, void* buf, int64_t n, int dtype, void* stream);
/* The same for `count` (<= 16) disjoint pieces of one product (the in-place dense runs of the vector and
 * the compact staging vector of the entries that can be non-zero), issued between ncclGroupStart /
 * ncclGroupEnd: ONE collective launch instead of `count`. */
int hf_allreduce_sum_multi(hf_comm_t* c, void* const* bufs, const int64_t* ns, int count, int dtype,
                           void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HF_PCG_H */
