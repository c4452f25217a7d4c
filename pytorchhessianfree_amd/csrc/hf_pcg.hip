// hf_pcg.hip -- hand-written gfx950 (CDNA4, wave64) kernels + C ABI of the
// Hessian-free Newton-step solver.  See include/hf_pcg.h for the boundary and
// DESIGN.md for the data layout / roofline of every kernel.
//
// Design in one paragraph.  The solver state is a set of persistent, contiguous
// HBM vectors (x, r, p, b, minv and the caller's B.p) plus one small device
// scalar block.  One PCG iteration is THREE streaming kernels, separated by the
// two global scalars alpha and beta that the algorithm forces:
//     K1 curvature  reads Bp,p                       -> G partial sums of p.(Bp+lambda p)
//     K2 update_xr  reads x,r,p,Bp,b(,minv) writes x,r -> partials of r.y, r.r, (r-b).x
//     K3 update_p   reads r,p(,minv)        writes p   (+ all termination tests)
// A kernel never reduces its own partial sums: each block of the NEXT kernel
// re-reduces the <=1024 fp64 partials (same order in every block -> bitwise
// identical scalars everywhere, no atomics, no extra launch, no host sync).
// Scalar fields are split so that no kernel reads a field that the same launch
// writes (blocks of one launch are not ordered).  After termination every
// kernel is a no-op, so speculative launches by the host are harmless.
//
// Arithmetic mirrors the reference's elementwise rounding (separate mul and add,
// no FMA contraction: this file is compiled with -ffp-contract=off); only the
// reduction ORDER of the dot products differs (fp64 accumulation here).
//
// Target: gfx950 only.  No CUDA paths, no hipify, no portability macros.

#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <new>
#include <vector>

#include "hf_common.h"

namespace {

constexpr int NP_CAP = 32;          // recorded non-positive-curvature events
constexpr int TIMING_CAP = 1024;    // iterations with per-kernel events

// ---------------------------------------------------------------------------
// device scalar block
// ---------------------------------------------------------------------------
struct DevState {
  // written by init_finalize / K3, read by K1/K2
  double ry_next;
  long long iter_next;
  long long slot_next;
  // written by K2 (block 0), read by K3
  double ry_cur;
  long long iter_cur;
  long long slot_cur;
  long long stored_cur;
  // written by init_finalize only
  double res_bound;
  // termination (written by K3 block 0)
  long long n_iters;
  int done;
  int pad0;
  // diagnostics
  double last_alpha, last_beta, last_pAp, last_res_norm;
  long long nonpos_count;
  long long nonpos_iter[NP_CAP];
  double nonpos_val[NP_CAP];
};


template <typename T> __device__ __forceinline__ T apply_damping(T bp, T p, T lam, bool damped) {
  // reference: mvp(x) + damping * x  (optimizer.py:266) -- two roundings
  return damped ? (T)(bp + (T)(lam * p)) : bp;
}

// ---------------------------------------------------------------------------
// init:  r = A x0 - b ; (optional) p = -M r ; partial sums
//   part[0] r.y   part[1] b.b   part[2] (r-b).x0
// ---------------------------------------------------------------------------
template <typename T, int MODE>
__global__ __launch_bounds__(BLOCK) void k_init(double* __restrict__ part, int stride,
                                                const T* __restrict__ x, T* __restrict__ r,
                                                T* __restrict__ p, const T* __restrict__ Ax0,
                                                const T* __restrict__ b,
                                                const T* __restrict__ minv, T* __restrict__ slot0,
                                                long long n) {
  constexpr int W = VecOf<T>::W;
  typedef typename VecOf<T>::type V;
  __shared__ double lds[3 * WAVES];
  double acc[3] = {0.0, 0.0, 0.0};
  const long long nvec = n / W;
  for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < nvec;
       i += (long long)gridDim.x * BLOCK) {
    VU<T> vx, va, vb, vm, vr, vp;
    vx.v = reinterpret_cast<const V*>(x)[i];
    va.v = reinterpret_cast<const V*>(Ax0)[i];
    vb.v = reinterpret_cast<const V*>(b)[i];
    if (MODE == HF_M_DIAG) vm.v = reinterpret_cast<const V*>(minv)[i];
#pragma unroll
    for (int c = 0; c < W; ++c) {
      const T rr = va.e[c] - vb.e[c];
      vr.e[c] = rr;
      acc[1] += (double)vb.e[c] * (double)vb.e[c];
      acc[2] += (double)(T)(rr - vb.e[c]) * (double)vx.e[c];
      if (MODE != HF_M_EXTERNAL) {
        const T y = (MODE == HF_M_DIAG) ? (T)(vm.e[c] * rr) : rr;
        acc[0] += (double)rr * (double)y;
        vp.e[c] = -y;
      }
    }
    reinterpret_cast<V*>(r)[i] = vr.v;
    if (MODE != HF_M_EXTERNAL) reinterpret_cast<V*>(p)[i] = vp.v;
    if (slot0) reinterpret_cast<V*>(slot0)[i] = vx.v;
  }
  if (blockIdx.x == 0) {
    const long long j = nvec * W + threadIdx.x;
    if (j < n) {
      const T rr = Ax0[j] - b[j];
      r[j] = rr;
      acc[1] += (double)b[j] * (double)b[j];
      acc[2] += (double)(T)(rr - b[j]) * (double)x[j];
      if (MODE != HF_M_EXTERNAL) {
        const T y = (MODE == HF_M_DIAG) ? (T)(minv[j] * rr) : rr;
        acc[0] += (double)rr * (double)y;
        p[j] = -y;
      }
      if (slot0) slot0[j] = x[j];
    }
  }
  write_partials<3>(part, stride, acc, lds);
}

// HF_M_EXTERNAL: p = -y, part[0] = r.y.  Same loop nest as k_init so that an
// identity M reproduces the HF_M_NONE sums bit for bit (tests/test_cg.py:217).
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_init_external(double* __restrict__ part, int stride,
                                                         const T* __restrict__ r,
                                                         T* __restrict__ p,
                                                         const T* __restrict__ y, long long n) {
  constexpr int W = VecOf<T>::W;
  typedef typename VecOf<T>::type V;
  __shared__ double lds[WAVES];
  double acc[1] = {0.0};
  const long long nvec = n / W;
  for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < nvec;
       i += (long long)gridDim.x * BLOCK) {
    VU<T> vr, vy, vp;
    vr.v = reinterpret_cast<const V*>(r)[i];
    vy.v = reinterpret_cast<const V*>(y)[i];
#pragma unroll
    for (int c = 0; c < W; ++c) {
      acc[0] += (double)vr.e[c] * (double)vy.e[c];
      vp.e[c] = -vy.e[c];
    }
    reinterpret_cast<V*>(p)[i] = vp.v;
  }
  if (blockIdx.x == 0) {
    const long long j = nvec * W + threadIdx.x;
    if (j < n) {
      acc[0] += (double)r[j] * (double)y[j];
      p[j] = -y[j];
    }
  }
  write_partials<1>(part, stride, acc, lds);
}

template <typename T>
__global__ __launch_bounds__(BLOCK) void k_init_finalize(DevState* __restrict__ st,
                                                         const double* __restrict__ part,
                                                         int nparts, int stride, double tol,
                                                         double atol, T* __restrict__ m_hist,
                                                         const long long* __restrict__ store_iters,
                                                         long long n_store, int* host_flag) {
  __shared__ double lds[3 * WAVES];
  double s[3];
  reduce_partials<3>(part, nparts, stride, s, lds);
  if (threadIdx.x == 0) {
    const T ry = (T)s[0];
    const T bnorm = (T)sqrt(s[1]);            // torch.linalg.norm(b)          cg.py:75
    double bound = tol * (double)bnorm;       // python float arithmetic       cg.py:75
    if (atol >= 0.0) bound = bound > atol ? bound : atol;  // cg.py:76
    if (m_hist) m_hist[0] = (T)0.5 * (T)s[2]; // 0.5*dot(r-b, x0)              cg.py:189
    st->ry_next = (double)ry;
    st->iter_next = 1;
    st->slot_next = (n_store > 0 && store_iters[0] == 0) ? 1 : 0;
    st->ry_cur = 0.0;
    st->iter_cur = 0;
    st->slot_cur = 0;
    st->stored_cur = 0;
    st->res_bound = bound;
    st->n_iters = 0;
    st->done = 0;
    st->last_alpha = st->last_beta = st->last_pAp = st->last_res_norm = 0.0;
    st->nonpos_count = 0;
    __hip_atomic_store(host_flag + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(host_flag, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// ---------------------------------------------------------------------------
// K1: partial sums of p.(Bp + lambda p)                         8N bytes
// ---------------------------------------------------------------------------
// All three streaming kernels issue the loads of their FIRST tile before anything
// else (the done flag, the re-reduction of the previous kernel's partials): at small N
// (All-CNN-C: 5.5 MB vectors, one tile per block) that prologue used to sit in front of
// the first load and cost more than the streaming itself.
// Non-temporal access to the solver's streams (template flag NT of K1 / K2, chosen per handle at run time: vectors
// of >= 16 M elements = six fp32 vectors of 1.5 x the 256 MiB Infinity Cache).  Beyond the
// cache nothing these kernels read survives until its next use, and non-temporal loads / stores stream faster
// (profiles/r04_pcg_nt_variants.jsonl, N = 100 M: K1 151.8 -> 137.2, K2 466.9 -> 440.0 us; all three kernels 0.70 ->
// 0.77 of 8 TB/s; N = 25.6 M: 0.71 -> 0.79).  At the ResNet-18 size (six vectors = 256 MiB, the cache's edge) the
// same-box A/B of the whole bench is inside its own noise (+1.8 % / +1.8 % in one batch, -3.0 % / +1.5 % in the
// next) while K2 / K3 themselves get ~1 us slower: default policy there, and for small vectors (All-CNN-C: 5.5 MB,
// L2-resident between kernels).
// NT covers: K1 both read streams; K2 the x / b / Bp loads and the x store (r and p are re-read by K3 right
// after).  K3 keeps the default policy up to HF_K3_NT_MIN elements: its r was written by K2 a moment ago and is still
// on-die (a non-temporal r load cost K3 24.8 -> 26.9 us at N = 11.2 M).  From 64 M elements (256 MiB per vector:
// ONE vector fills the Infinity Cache) nothing K2 wrote is on-die any more and nothing K3 writes survives until the
// product reads it: r / p / minv loads and the p store go non-temporal there (round 6; N = 100 M: DESIGN.md
// section 6.1, profiles/r06_pcg_kernel_bench.jsonl).
#ifndef HF_K3_NT_MIN
#define HF_K3_NT_MIN 64000000LL
#endif
#define HF_LD(NTFLAG, dst, ptr)                                                             \
  {                                                                                         \
    if constexpr (NTFLAG) {                                                                 \
      NV t_ = __builtin_nontemporal_load(reinterpret_cast<const NV*>(ptr));                 \
      __builtin_memcpy(&(dst), &t_, sizeof(NV));                                            \
    } else {                                                                                \
      (dst) = *(ptr);                                                                       \
    }                                                                                       \
  }
#define HF_ST(NTFLAG, ptr, src)                                                             \
  {                                                                                         \
    if constexpr (NTFLAG) {                                                                 \
      NV t_;                                                                                \
      __builtin_memcpy(&t_, &(src), sizeof(NV));                                            \
      __builtin_nontemporal_store(t_, reinterpret_cast<NV*>(ptr));                          \
    } else {                                                                                \
      *(ptr) = (src);                                                                       \
    }                                                                                       \
  }
template <typename T, int UNROLL, bool NT>
__global__ __launch_bounds__(BLOCK) void k_curvature(const DevState* __restrict__ st,
                                                     double* __restrict__ part1, int stride,
                                                     const T* __restrict__ p,
                                                     const T* __restrict__ Bp, T lam, int damped,
                                                     long long n) {
  constexpr int W = VecOf<T>::W;
  typedef typename VecOf<T>::type V;
  typedef T NV __attribute__((ext_vector_type(W)));
  __shared__ double lds[WAVES];
  double acc[1] = {0.0};
  const long long nvec = n / W;
  const long long tile = (long long)BLOCK * UNROLL;
  long long base = (long long)blockIdx.x * tile;
  VU<T> vp[UNROLL], vg[UNROLL];
#define HF_K1_LOAD()                                                        \
  _Pragma("unroll") for (int u = 0; u < UNROLL; ++u) {                       \
    const long long i = base + u * BLOCK + threadIdx.x;                     \
    if (i < nvec) {                                                         \
      HF_LD(NT, vp[u].v, reinterpret_cast<const V*>(p) + i)                 \
      HF_LD(NT, vg[u].v, reinterpret_cast<const V*>(Bp) + i)                \
    }                                                                       \
  }
  HF_K1_LOAD();
  if (st->done) return;
  while (base < nvec) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const long long i = base + u * BLOCK + threadIdx.x;
      if (i < nvec) {
#pragma unroll
        for (int c = 0; c < W; ++c) {
          const T ap = apply_damping<T>(vg[u].e[c], vp[u].e[c], lam, damped);
          acc[0] += (double)vp[u].e[c] * (double)ap;
        }
      }
    }
    base += (long long)gridDim.x * tile;
    HF_K1_LOAD();
  }
#undef HF_K1_LOAD
  if (blockIdx.x == 0) {
    const long long j = nvec * W + threadIdx.x;
    if (j < n) acc[0] += (double)p[j] * (double)apply_damping<T>(Bp[j], p[j], lam, damped);
  }
  write_partials<1>(part1, stride, acc, lds);
}

// ---------------------------------------------------------------------------
// K2: alpha; x += alpha p; r += alpha (Bp + lambda p); snapshot; partials
//   part2[0] r.y   part2[1] r.r   part2[2] (r-b).x         28N (32N with minv)
// ---------------------------------------------------------------------------
template <typename T, int MODE, int UNROLL, bool NT>
__global__ __launch_bounds__(BLOCK) void k_update_xr(
    DevState* __restrict__ st, const double* __restrict__ part1, double* __restrict__ part2,
    int nparts, int stride, T* __restrict__ x, T* __restrict__ r, const T* __restrict__ p,
    const T* __restrict__ Bp, const T* __restrict__ b, const T* __restrict__ minv, T lam,
    int damped, const long long* __restrict__ store_iters, long long n_store,
    T* __restrict__ slab, long long slab_stride, long long n) {
  constexpr int W = VecOf<T>::W;
  typedef typename VecOf<T>::type V;
  typedef T NV __attribute__((ext_vector_type(W)));
  __shared__ double lds[3 * WAVES];
  const long long nvec = n / W;
  const long long tile = (long long)BLOCK * UNROLL;
  long long base = (long long)blockIdx.x * tile;
  VU<T> vx[UNROLL], vr[UNROLL], vp[UNROLL], vg[UNROLL], vb[UNROLL], vm[UNROLL];
#define HF_K2_LOAD()                                                        \
  _Pragma("unroll") for (int u = 0; u < UNROLL; ++u) {                       \
    const long long i = base + u * BLOCK + threadIdx.x;                     \
    if (i < nvec) {                                                         \
      HF_LD(NT, vx[u].v, reinterpret_cast<const V*>(x) + i)                 \
      vr[u].v = reinterpret_cast<const V*>(r)[i];                           \
      vp[u].v = reinterpret_cast<const V*>(p)[i];                           \
      HF_LD(NT, vg[u].v, reinterpret_cast<const V*>(Bp) + i)                \
      HF_LD(NT, vb[u].v, reinterpret_cast<const V*>(b) + i)                 \
      if (MODE == HF_M_DIAG) vm[u].v = reinterpret_cast<const V*>(minv)[i]; \
    }                                                                       \
  }
  HF_K2_LOAD();  // in flight while alpha is being put together
  if (st->done) return;

  double s[1];
  reduce_partials<1>(part1, nparts, stride, s, lds);
  const T pAp = (T)s[0];                       // torch.dot(p, Ap)             cg.py:206
  const T ry = (T)st->ry_next;
  const T alpha = ry / pAp;                    //                              cg.py:207
  const long long iter = st->iter_next;
  const long long slot = st->slot_next;
  const bool store = (slot < n_store) && (store_iters[slot] == iter);  // cg.py:209
  T* __restrict__ snap = store ? slab + slot * slab_stride : nullptr;

  if (blockIdx.x == 0 && threadIdx.x == 0) {
    st->ry_cur = (double)ry;
    st->iter_cur = iter;
    st->slot_cur = slot;
    st->stored_cur = store ? 1 : 0;
    st->last_pAp = (double)pAp;
    st->last_alpha = (double)alpha;
    if (!(pAp > (T)0)) {                       // _postprocess_pAp             cg.py:133-139
      const long long c = st->nonpos_count;
      if (c < NP_CAP) { st->nonpos_iter[c] = iter; st->nonpos_val[c] = (double)pAp; }
      st->nonpos_count = c + 1;
    }
  }

  double acc[3] = {0.0, 0.0, 0.0};
  while (base < nvec) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const long long i = base + u * BLOCK + threadIdx.x;
      if (i < nvec) {
#pragma unroll
        for (int c = 0; c < W; ++c) {
          const T pp = vp[u].e[c];
          const T ap = apply_damping<T>(vg[u].e[c], pp, lam, damped);
          const T xn = vx[u].e[c] + (T)(alpha * pp);   // x = x + alpha*p   cg.py:208
          const T rn = vr[u].e[c] + (T)(alpha * ap);   // r = r + alpha*Ap  cg.py:211
          vx[u].e[c] = xn;
          vr[u].e[c] = rn;
          if (MODE == HF_M_DIAG) acc[0] += (double)rn * (double)(T)(vm[u].e[c] * rn);
          acc[1] += (double)rn * (double)rn;
          acc[2] += (double)(T)(rn - vb[u].e[c]) * (double)xn;  // dot(r-b, x)  cg.py:97
        }
        HF_ST(NT, reinterpret_cast<V*>(x) + i, vx[u].v)
        reinterpret_cast<V*>(r)[i] = vr[u].v;
        if (snap) reinterpret_cast<V*>(snap)[i] = vx[u].v;
      }
    }
    base += (long long)gridDim.x * tile;
    HF_K2_LOAD();
  }
#undef HF_K2_LOAD
  if (blockIdx.x == 0) {
    const long long j = nvec * W + threadIdx.x;
    if (j < n) {
      const T pp = p[j];
      const T ap = apply_damping<T>(Bp[j], pp, lam, damped);
      const T xn = x[j] + (T)(alpha * pp);
      const T rn = r[j] + (T)(alpha * ap);
      x[j] = xn;
      r[j] = rn;
      if (snap) snap[j] = xn;
      if (MODE == HF_M_DIAG) acc[0] += (double)rn * (double)(T)(minv[j] * rn);
      acc[1] += (double)rn * (double)rn;
      acc[2] += (double)(T)(rn - b[j]) * (double)xn;
    }
  }
  if (MODE == HF_M_NONE) acc[0] = acc[1];
  write_partials<3>(part2, stride, acc, lds);
}

// K3a (HF_M_EXTERNAL only): part3 = r.y.  Same tile walk as K2 (see k_init_external).
template <typename T, int UNROLL>
__global__ __launch_bounds__(BLOCK) void k_dot_ry(const DevState* __restrict__ st,
                                                  double* __restrict__ part3, int stride,
                                                  const T* __restrict__ r,
                                                  const T* __restrict__ y, long long n) {
  if (st->done) return;
  constexpr int W = VecOf<T>::W;
  typedef typename VecOf<T>::type V;
  __shared__ double lds[WAVES];
  double acc[1] = {0.0};
  const long long nvec = n / W;
  const long long tile = (long long)BLOCK * UNROLL;
  for (long long base = (long long)blockIdx.x * tile; base < nvec;
       base += (long long)gridDim.x * tile) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const long long i = base + u * BLOCK + threadIdx.x;
      if (i < nvec) {
        VU<T> vr, vy;
        vr.v = reinterpret_cast<const V*>(r)[i];
        vy.v = reinterpret_cast<const V*>(y)[i];
#pragma unroll
        for (int c = 0; c < W; ++c) acc[0] += (double)vr.e[c] * (double)vy.e[c];
      }
    }
  }
  if (blockIdx.x == 0) {
    const long long j = nvec * W + threadIdx.x;
    if (j < n) acc[0] += (double)r[j] * (double)y[j];
  }
  write_partials<1>(part3, stride, acc, lds);
}

// ---------------------------------------------------------------------------
// K3: finalise ||r||, m_i; termination tests (cg.py:95-115); beta; p = -y + beta p
//                                                           12N (16N with minv)
// ---------------------------------------------------------------------------
template <typename T, int MODE, int UNROLL, bool NT>
__global__ __launch_bounds__(BLOCK) void k_update_p(
    DevState* __restrict__ st, const double* __restrict__ part2,
    const double* __restrict__ part3, int nparts, int stride, const T* __restrict__ r,
    T* __restrict__ p, const T* __restrict__ minv, const T* __restrict__ yext,
    T* __restrict__ m_hist, long long max_iter, int* host_flag, long long n) {
  constexpr int W = VecOf<T>::W;
  typedef typename VecOf<T>::type V;
  typedef T NV __attribute__((ext_vector_type(W)));
  __shared__ double lds[3 * WAVES];
  const long long nvec = n / W;
  const long long tile = (long long)BLOCK * UNROLL;
  long long base = (long long)blockIdx.x * tile;
  VU<T> vr[UNROLL], vp[UNROLL], vm[UNROLL];
#define HF_K3_LOAD()                                                        \
  _Pragma("unroll") for (int u = 0; u < UNROLL; ++u) {                       \
    const long long i = base + u * BLOCK + threadIdx.x;                     \
    if (i < nvec) {                                                         \
      if (MODE == HF_M_EXTERNAL) vr[u].v = reinterpret_cast<const V*>(yext)[i]; \
      else HF_LD(NT, vr[u].v, reinterpret_cast<const V*>(r) + i)            \
      HF_LD(NT, vp[u].v, reinterpret_cast<const V*>(p) + i)                 \
      if (MODE == HF_M_DIAG) HF_LD(NT, vm[u].v, reinterpret_cast<const V*>(minv) + i) \
    }                                                                       \
  }
  HF_K3_LOAD();  // in flight while beta and the termination tests are evaluated
  if (st->done) return;

  double s[3];
  reduce_partials<3>(part2, nparts, stride, s, lds);
  if (MODE == HF_M_EXTERNAL) {
    double e[1];
    reduce_partials<1>(part3, nparts, stride, e, lds);
    s[0] = e[0];
  }
  const long long iter = st->iter_cur;
  const T ry_old = (T)st->ry_cur;
  const T ry_new = (T)s[0];                    // torch.dot(r, y)              cg.py:221
  const T res_norm = (T)sqrt(s[1]);            // torch.linalg.norm(r)         cg.py:93
  const T m_i = (T)0.5 * (T)s[2];              // 0.5*torch.dot(r-b, x)        cg.py:97

  int reason = HF_RUNNING;
  if (m_hist) {                                // Martens' test                cg.py:96-103
    const long long a = iter / 10;
    const long long k = a > 10 ? a : 10;
    if (k < iter) {
      const T num = m_i - m_hist[iter - k];
      const T den = m_i - m_hist[0];
      if ((T)(num / den) < (T)5e-4) reason = HF_REASON_MARTENS;
    }
  }
  if (reason == HF_RUNNING) {
    if (iter >= max_iter) reason = HF_REASON_MAXITER;                      // cg.py:106
    else if (res_norm != res_norm) reason = HF_REASON_DIVERGED;            // cg.py:110
    else if (res_norm < (T)st->res_bound) reason = HF_REASON_TOL;          // cg.py:114
  }
  const T beta = ry_new / ry_old;              //                              cg.py:222

  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (m_hist) m_hist[iter] = m_i;
    st->last_res_norm = (double)res_norm;
    st->slot_next = st->slot_cur + st->stored_cur;
    if (reason != HF_RUNNING) {
      st->n_iters = iter;
      st->done = reason;
      // host mirror: [1] = terminating iteration, then [0] = reason
      __hip_atomic_store(host_flag + 1, (int)(iter > 0x7fffffff ? 0x7fffffff : iter),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(host_flag, reason, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    } else {
      st->ry_next = (double)ry_new;
      st->iter_next = iter + 1;
      st->last_beta = (double)beta;
    }
  }
  if (reason != HF_RUNNING) return;

  while (base < nvec) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const long long i = base + u * BLOCK + threadIdx.x;
      if (i < nvec) {
#pragma unroll
        for (int c = 0; c < W; ++c) {
          const T y = (MODE == HF_M_DIAG) ? (T)(vm[u].e[c] * vr[u].e[c]) : vr[u].e[c];
          vp[u].e[c] = (-y) + (T)(beta * vp[u].e[c]);   // p = -y + beta*p   cg.py:224
        }
        HF_ST(NT, reinterpret_cast<V*>(p) + i, vp[u].v)
      }
    }
    base += (long long)gridDim.x * tile;
    HF_K3_LOAD();
  }
#undef HF_K3_LOAD
  if (blockIdx.x == 0) {
    const long long j = nvec * W + threadIdx.x;
    if (j < n) {
      const T rv = (MODE == HF_M_EXTERNAL) ? yext[j] : r[j];
      const T y = (MODE == HF_M_DIAG) ? (T)(minv[j] * rv) : rv;
      p[j] = (-y) + (T)(beta * p[j]);
    }
  }
}

// minv = (diag + damping)^(-exponent)
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_precond_build(T* __restrict__ minv,
                                                         const T* __restrict__ diag, T lam,
                                                         T neg_exp, long long n) {
  for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < n;
       i += (long long)gridDim.x * BLOCK)
    minv[i] = pow((T)(diag[i] + lam), neg_exp);
}

// out = a + alpha*s
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_axpy_out(T* out, const T* a, const T* s, T alpha,
                                                    long long n, int vec_ok) {
  constexpr int W = VecOf<T>::W;
  typedef typename VecOf<T>::type V;
  const long long nvec = vec_ok ? n / W : 0;
  for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < nvec;
       i += (long long)gridDim.x * BLOCK) {
    VU<T> va, vs;
    va.v = reinterpret_cast<const V*>(a)[i];
    vs.v = reinterpret_cast<const V*>(s)[i];
#pragma unroll
    for (int c = 0; c < W; ++c) va.e[c] = va.e[c] + (T)(alpha * vs.e[c]);
    reinterpret_cast<V*>(out)[i] = va.v;
  }
  for (long long j = nvec * W + (long long)blockIdx.x * BLOCK + threadIdx.x; j < n;
       j += (long long)gridDim.x * BLOCK)
    out[j] = a[j] + (T)(alpha * s[j]);
}

}  // namespace

// ===========================================================================
// host side
// ===========================================================================
struct hf_pcg {
  int64_t n;
  int dtype;
  int grid;            // blocks of the vector kernels at large N (2 per CU)
  int grid_cap;        // partial-sum slots per sum (>= any grid a kernel is launched with)
  DevState* d_state;
  double* d_part;      // [3 sums][3 slots][grid]
  DevState* h_state;   // pinned
  int* h_flag;         // pinned, device-visible
  int* d_flag;         // device alias of h_flag
  // borrowed for the current solve
  void *x, *r, *p;
  const void *b, *minv;
  int precond;
  int64_t max_iter;
  double tol, atol;
  int martens;
  const int64_t* store_iters;
  int64_t n_store;
  void* slab;
  int64_t slab_stride;
  int store_x0;
  void* m_hist;
  int begun, inited, finished;
  // timing
  int timing;
  int64_t t_count;
  hipEvent_t* ev;      // 4 per iteration
  double g_ms[3];      // sums of the sampled per-kernel times of hf_pcg_graph launches
  int64_t g_count;
};


namespace {
inline double* part_ptr(hf_pcg* h, int which) { return h->d_part + (size_t)which * 3 * h->grid_cap; }

// Blocks for a kernel whose blocks walk tiles of BLOCK*unroll 16-byte vectors.  Large
// vectors: 2 blocks per CU, each walking many tiles (tuned).  Small
// vectors (<= SMALL_TILES tiles, e.g. All-CNN-C's 5.5 MB): ONE tile per block -- with the
// capped grid some blocks walked two tiles and the rest one, a 2x tail on a kernel that
// lasts a few microseconds.
constexpr int SMALL_TILES = 2048;
int grid_for(const hf_pcg* h, int unroll) {
  const int W = h->dtype == HF_F32 ? 4 : 2;
  const int64_t nvec = h->n / W;
  int64_t tiles = (nvec + (int64_t)BLOCK * unroll - 1) / ((int64_t)BLOCK * unroll);
  if (tiles < 1) tiles = 1;
  if (tiles <= SMALL_TILES && tiles <= h->grid_cap) return (int)tiles;
  return (int)(tiles < h->grid ? tiles : h->grid);
}

// non-temporal streams for vectors that cannot stay cached between their uses (see HF_LD)
bool nt_streams(const hf_pcg* h) {
  return h->n >= 16000000LL;  // (six fp32 vectors of 1.5 x the 256 MiB Infinity Cache)
}
}  // namespace

// C linkage comes from the declarations in hf_pcg.h

int hf_abi_version(void) { return HF_ABI_VERSION; }

const char* hf_error_string(int code) {
  switch (code) {
    case HF_OK: return "ok";
    case HF_ERR_ARG: return "hf: invalid argument";
    case HF_ERR_ALIGN: return "hf: vector pointer not 16-byte aligned";
    case HF_ERR_STATE: return "hf: call order violated";
    case HF_ERR_NOSYMBOL: return "hf: RCCL symbol not found in this process";
    case HF_ERR_CAPACITY: return "hf: capacity exceeded";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "hf: unknown error";
  }
}

int hf_pcg_create(hf_pcg_t** out, int64_t n, int dtype, int max_blocks) {
  if (!out || n <= 0 || (dtype != HF_F32 && dtype != HF_F64) || max_blocks < 0) return HF_ERR_ARG;
  hf_pcg* h = new (std::nothrow) hf_pcg();
  if (!h) return HF_ERR_ARG;
  memset(h, 0, sizeof(*h));
  h->n = n;
  h->dtype = dtype;
  const bool explicit_blocks = max_blocks != 0;
  if (max_blocks == 0) {
    int dev = 0, cus = 256;
    HF_HIP(hipGetDevice(&dev));
    HF_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    max_blocks = 2 * cus;  // tuned on MI355X (scripts/experiments/tune.py): 2 per CU, unroll 2
  }
  h->grid = max_blocks;
  // an explicit max_blocks (tests, tuning) is a hard cap; the default allows the
  // one-tile-per-block grids of small vectors
  h->grid_cap = explicit_blocks ? max_blocks : (max_blocks > SMALL_TILES ? max_blocks : SMALL_TILES);
  HF_HIP(hipMalloc((void**)&h->d_state, sizeof(DevState)));
  HF_HIP(hipMemset(h->d_state, 0, sizeof(DevState)));
  HF_HIP(hipMalloc((void**)&h->d_part, sizeof(double) * 3 * 3 * (size_t)h->grid_cap));
  HF_HIP(hipMemset(h->d_part, 0, sizeof(double) * 3 * 3 * (size_t)h->grid_cap));
  HF_HIP(hipHostMalloc((void**)&h->h_state, sizeof(DevState), hipHostMallocDefault));
  HF_HIP(hipHostMalloc((void**)&h->h_flag, 64, hipHostMallocMapped | hipHostMallocCoherent));
  h->h_flag[0] = 0;
  h->h_flag[1] = 0;
  HF_HIP(hipHostGetDevicePointer((void**)&h->d_flag, h->h_flag, 0));
  *out = h;
  return HF_OK;
}

int hf_pcg_destroy(hf_pcg_t* h) {
  if (!h) return HF_OK;
  if (h->ev) {
    for (int i = 0; i < 4 * TIMING_CAP; ++i) (void)hipEventDestroy(h->ev[i]);
    delete[] h->ev;
  }
  (void)hipFree(h->d_state);
  (void)hipFree(h->d_part);
  (void)hipHostFree(h->h_state);
  (void)hipHostFree(h->h_flag);
  delete h;
  return HF_OK;
}

int hf_pcg_begin(hf_pcg_t* h, void* x, void* r, void* p, const void* b, const void* minv,
                 int precond, int64_t max_iter, double tol, double atol, int martens,
                 const int64_t* store_iters, int64_t n_store, int store_x0, void* slab,
                 int64_t slab_stride, void* m_hist) {
  if (!h || !x || !r || !p || !b || max_iter < 1) return HF_ERR_ARG;
  if (precond < HF_M_NONE || precond > HF_M_EXTERNAL) return HF_ERR_ARG;
  if (precond == HF_M_DIAG && !minv) return HF_ERR_ARG;
  if (martens && !m_hist) return HF_ERR_ARG;
  if (n_store < 0 || (n_store > 0 && (!store_iters || !slab || slab_stride < h->n)))
    return HF_ERR_ARG;
  if (store_x0 && n_store == 0) return HF_ERR_ARG;
  if (!aligned16(x) || !aligned16(r) || !aligned16(p) || !aligned16(b) ||
      (minv && !aligned16(minv)) || (slab && !aligned16(slab)))
    return HF_ERR_ALIGN;
  const int W = h->dtype == HF_F32 ? 4 : 2;
  if (n_store > 0 && (slab_stride % W) != 0) return HF_ERR_ALIGN;
  h->x = x; h->r = r; h->p = p; h->b = b; h->minv = minv;
  h->precond = precond;
  h->max_iter = max_iter;
  h->tol = tol; h->atol = atol;
  h->martens = martens;
  h->store_iters = store_iters; h->n_store = n_store;
  h->slab = slab; h->slab_stride = slab_stride;
  h->store_x0 = store_x0 ? 1 : 0;
  h->m_hist = martens ? m_hist : nullptr;
  h->begun = 1; h->inited = 0; h->finished = 0;
  h->t_count = 0;
  h->h_flag[0] = 0;
  h->h_flag[1] = 0;
  return HF_OK;
}

template <typename T>
static int init_impl(hf_pcg* h, const void* Ax0, int slot0, hipStream_t s) {
  const int g = grid_for(h, 1);
  double* part = part_ptr(h, 0);
  T* snap = slot0 ? (T*)h->slab : nullptr;
#define HF_LAUNCH_INIT(MODE)                                                                   \
  hipLaunchKernelGGL((k_init<T, MODE>), dim3(g), dim3(BLOCK), 0, s, part, h->grid_cap,         \
                     (const T*)h->x, (T*)h->r, (T*)h->p, (const T*)Ax0, (const T*)h->b,        \
                     (const T*)h->minv, snap, (long long)h->n)
  switch (h->precond) {
    case HF_M_NONE: HF_LAUNCH_INIT(HF_M_NONE); break;
    case HF_M_DIAG: HF_LAUNCH_INIT(HF_M_DIAG); break;
    default: HF_LAUNCH_INIT(HF_M_EXTERNAL); break;
  }
#undef HF_LAUNCH_INIT
  HF_HIP(hipGetLastError());
  if (h->precond != HF_M_EXTERNAL) {
    hipLaunchKernelGGL((k_init_finalize<T>), dim3(1), dim3(BLOCK), 0, s, h->d_state, part, g,
                       h->grid_cap, h->tol, h->atol, (T*)h->m_hist,
                       (const long long*)h->store_iters, (long long)h->n_store, h->d_flag);
    HF_HIP(hipGetLastError());
    h->inited = 1;
  }
  return HF_OK;
}

int hf_pcg_init(hf_pcg_t* h, const void* Ax0, void* stream) {
  if (!h || !Ax0) return HF_ERR_ARG;
  if (!h->begun) return HF_ERR_STATE;
  if (!aligned16(Ax0)) return HF_ERR_ALIGN;
  const int slot0 = h->store_x0;
  return h->dtype == HF_F32 ? init_impl<float>(h, Ax0, slot0, (hipStream_t)stream)
                            : init_impl<double>(h, Ax0, slot0, (hipStream_t)stream);
}

int hf_pcg_init_external(hf_pcg_t* h, const void* y, void* stream) {
  if (!h || !y) return HF_ERR_ARG;
  if (!h->begun || h->precond != HF_M_EXTERNAL) return HF_ERR_STATE;
  if (!aligned16(y)) return HF_ERR_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  const int g = grid_for(h, 1);
  double* part = part_ptr(h, 0);
  if (h->dtype == HF_F32) {
    hipLaunchKernelGGL((k_init_external<float>), dim3(g), dim3(BLOCK), 0, s, part, h->grid_cap,
                       (const float*)h->r, (float*)h->p, (const float*)y, (long long)h->n);
    hipLaunchKernelGGL((k_init_finalize<float>), dim3(1), dim3(BLOCK), 0, s, h->d_state, part, g,
                       h->grid_cap, h->tol, h->atol, (float*)h->m_hist,
                       (const long long*)h->store_iters, (long long)h->n_store, h->d_flag);
  } else {
    hipLaunchKernelGGL((k_init_external<double>), dim3(g), dim3(BLOCK), 0, s, part, h->grid_cap,
                       (const double*)h->r, (double*)h->p, (const double*)y, (long long)h->n);
    hipLaunchKernelGGL((k_init_finalize<double>), dim3(1), dim3(BLOCK), 0, s, h->d_state, part,
                       g, h->grid_cap, h->tol, h->atol, (double*)h->m_hist,
                       (const long long*)h->store_iters, (long long)h->n_store, h->d_flag);
  }
  HF_HIP(hipGetLastError());
  h->inited = 1;
  return HF_OK;
}

// unroll factors (x 16-B vectors per lane and stream); all variants from 1..8 and
// 2..16 blocks per CU measured within 3 % of each other on MI355X -- the kernels sit
// on the memory system's plateau
#ifndef HF_U1
#define HF_U1 2
#endif
#ifndef HF_U2
#define HF_U2 2
#endif
#ifndef HF_U3
#define HF_U3 2
#endif
constexpr int U1 = HF_U1, U2 = HF_U2, U3 = HF_U3;

// ---- launch descriptors -----------------------------------------------------
// One description of a K1/K2/K3 launch (function, grid, argument values) serves both
// the direct launch (hipLaunchKernel) and the kernel nodes of the per-iteration
// hipGraph (hf_pcg_graph_*), whose arguments are refreshed per solve.
struct KLaunch {
  const void* func;
  int grid;
  void* args[20];
  // argument storage
  DevState* st;
  const double *pa, *pb;
  double* pw;
  int i0, i1, i2;
  void *v0, *v1, *v2, *v3, *v4, *v5, *v6, *v7;
  float lam_f;
  double lam_d;
  long long l0, l1, l2;
  int* flag;
};

template <typename T>
static void* lam_slot(KLaunch& k, double damping);
template <> void* lam_slot<float>(KLaunch& k, double damping) { k.lam_f = (float)damping; return &k.lam_f; }
template <> void* lam_slot<double>(KLaunch& k, double damping) { k.lam_d = damping; return &k.lam_d; }

template <typename T>
static void build_k1(hf_pcg* h, const void* Bp, double damping, KLaunch& k) {
  k.func = nt_streams(h) ? (const void*)&k_curvature<T, U1, true> : (const void*)&k_curvature<T, U1, false>;
  k.grid = grid_for(h, U1);
  k.st = h->d_state; k.pw = part_ptr(h, 1); k.i0 = h->grid_cap;
  k.v0 = h->p; k.v1 = const_cast<void*>(Bp); k.i1 = damping != 0.0 ? 1 : 0; k.l0 = h->n;
  void* a[] = {&k.st, &k.pw, &k.i0, &k.v0, &k.v1, lam_slot<T>(k, damping), &k.i1, &k.l0};
  memcpy(k.args, a, sizeof(a));
}

template <typename T>
static void build_k2(hf_pcg* h, const void* Bp, double damping, KLaunch& k) {
  const bool nt = nt_streams(h);
  switch (h->precond) {
    case HF_M_NONE: k.func = nt ? (const void*)&k_update_xr<T, HF_M_NONE, U2, true> : (const void*)&k_update_xr<T, HF_M_NONE, U2, false>; break;
    case HF_M_DIAG: k.func = nt ? (const void*)&k_update_xr<T, HF_M_DIAG, U2, true> : (const void*)&k_update_xr<T, HF_M_DIAG, U2, false>; break;
    default: k.func = nt ? (const void*)&k_update_xr<T, HF_M_EXTERNAL, U2, true> : (const void*)&k_update_xr<T, HF_M_EXTERNAL, U2, false>; break;
  }
  k.grid = grid_for(h, U2);
  k.st = h->d_state; k.pa = part_ptr(h, 1); k.pw = part_ptr(h, 2);
  k.i0 = grid_for(h, U1); k.i1 = h->grid_cap;
  k.v0 = h->x; k.v1 = h->r; k.v2 = h->p; k.v3 = const_cast<void*>(Bp);
  k.v4 = const_cast<void*>(h->b); k.v5 = const_cast<void*>(h->minv);
  k.i2 = damping != 0.0 ? 1 : 0;
  k.v6 = const_cast<int64_t*>(h->store_iters); k.l0 = h->n_store; k.v7 = h->slab;
  k.l1 = h->slab_stride; k.l2 = h->n;
  void* a[] = {&k.st, &k.pa, &k.pw, &k.i0, &k.i1, &k.v0, &k.v1, &k.v2, &k.v3, &k.v4, &k.v5,
               lam_slot<T>(k, damping), &k.i2, &k.v6, &k.l0, &k.v7, &k.l1, &k.l2};
  memcpy(k.args, a, sizeof(a));
}

template <typename T>
static void build_k3(hf_pcg* h, const void* yext, KLaunch& k) {
  const bool nt = h->n >= HF_K3_NT_MIN;  // (see HF_LD: below that K3's operands are on-die)
  switch (h->precond) {
    case HF_M_NONE: k.func = nt ? (const void*)&k_update_p<T, HF_M_NONE, U3, true> : (const void*)&k_update_p<T, HF_M_NONE, U3, false>; break;
    case HF_M_DIAG: k.func = nt ? (const void*)&k_update_p<T, HF_M_DIAG, U3, true> : (const void*)&k_update_p<T, HF_M_DIAG, U3, false>; break;
    default: k.func = nt ? (const void*)&k_update_p<T, HF_M_EXTERNAL, U3, true> : (const void*)&k_update_p<T, HF_M_EXTERNAL, U3, false>; break;
  }
  k.grid = grid_for(h, U3);
  k.st = h->d_state; k.pa = part_ptr(h, 2); k.pb = part_ptr(h, 0);
  k.i0 = grid_for(h, U2); k.i1 = h->grid_cap;
  k.v0 = h->r; k.v1 = h->p; k.v2 = const_cast<void*>(h->minv); k.v3 = const_cast<void*>(yext);
  k.v4 = h->m_hist; k.l0 = h->max_iter; k.flag = h->d_flag; k.l1 = h->n;
  void* a[] = {&k.st, &k.pa, &k.pb, &k.i0, &k.i1, &k.v0, &k.v1, &k.v2, &k.v3, &k.v4, &k.l0,
               &k.flag, &k.l1};
  memcpy(k.args, a, sizeof(a));
}

static void build_iteration(hf_pcg* h, const void* Bp, double damping, KLaunch (&k)[3]) {
  if (h->dtype == HF_F32) {
    build_k1<float>(h, Bp, damping, k[0]); build_k2<float>(h, Bp, damping, k[1]);
    build_k3<float>(h, nullptr, k[2]);
  } else {
    build_k1<double>(h, Bp, damping, k[0]); build_k2<double>(h, Bp, damping, k[1]);
    build_k3<double>(h, nullptr, k[2]);
  }
}

static int launch(const KLaunch& k, hipStream_t s) {
  HF_HIP(hipLaunchKernel(k.func, dim3(k.grid), dim3(BLOCK), const_cast<void**>(k.args), 0, s));
  return HF_OK;
}

template <typename T>
static int curvature_impl(hf_pcg* h, const void* Bp, double damping, hipStream_t s) {
  KLaunch k;
  build_k1<T>(h, Bp, damping, k);
  return launch(k, s);
}

template <typename T>
static int update_xr_impl(hf_pcg* h, const void* Bp, double damping, hipStream_t s) {
  KLaunch k;
  build_k2<T>(h, Bp, damping, k);
  return launch(k, s);
}

template <typename T>
static int update_p_impl(hf_pcg* h, const void* yext, hipStream_t s) {
  if (h->precond == HF_M_EXTERNAL) {
    // same grid as K2 so that part2 and part3 hold the same number of partials
    hipLaunchKernelGGL((k_dot_ry<T, U2>), dim3(grid_for(h, U2)), dim3(BLOCK), 0, s, h->d_state,
                       part_ptr(h, 0), h->grid_cap, (const T*)h->r, (const T*)yext, (long long)h->n);
    HF_HIP(hipGetLastError());
  }
  KLaunch k;
  build_k3<T>(h, yext, k);
  return launch(k, s);
}

int hf_pcg_curvature(hf_pcg_t* h, const void* Bp, double damping, void* stream) {
  if (!h || !Bp) return HF_ERR_ARG;
  if (!h->inited) return HF_ERR_STATE;
  if (!aligned16(Bp)) return HF_ERR_ALIGN;
  return h->dtype == HF_F32 ? curvature_impl<float>(h, Bp, damping, (hipStream_t)stream)
                            : curvature_impl<double>(h, Bp, damping, (hipStream_t)stream);
}

int hf_pcg_update_xr(hf_pcg_t* h, const void* Bp, double damping, void* stream) {
  if (!h || !Bp) return HF_ERR_ARG;
  if (!h->inited) return HF_ERR_STATE;
  if (!aligned16(Bp)) return HF_ERR_ALIGN;
  return h->dtype == HF_F32 ? update_xr_impl<float>(h, Bp, damping, (hipStream_t)stream)
                            : update_xr_impl<double>(h, Bp, damping, (hipStream_t)stream);
}

int hf_pcg_update_p(hf_pcg_t* h, const void* y_external, void* stream) {
  if (!h) return HF_ERR_ARG;
  if (!h->inited) return HF_ERR_STATE;
  if (h->precond == HF_M_EXTERNAL && (!y_external || !aligned16(y_external)))
    return y_external ? HF_ERR_ALIGN : HF_ERR_ARG;
  return h->dtype == HF_F32 ? update_p_impl<float>(h, y_external, (hipStream_t)stream)
                            : update_p_impl<double>(h, y_external, (hipStream_t)stream);
}

int hf_pcg_iterate(hf_pcg_t* h, const void* Bp, double damping, void* stream) {
  if (!h || !Bp) return HF_ERR_ARG;
  if (!h->inited || h->precond == HF_M_EXTERNAL) return HF_ERR_STATE;
  if (!aligned16(Bp)) return HF_ERR_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  const bool t = h->timing && h->t_count < TIMING_CAP;
  hipEvent_t* ev = t ? h->ev + 4 * h->t_count : nullptr;
  int rc;
  if (t) HF_HIP(hipEventRecord(ev[0], s));
  rc = h->dtype == HF_F32 ? curvature_impl<float>(h, Bp, damping, s)
                          : curvature_impl<double>(h, Bp, damping, s);
  if (rc) return rc;
  if (t) HF_HIP(hipEventRecord(ev[1], s));
  rc = h->dtype == HF_F32 ? update_xr_impl<float>(h, Bp, damping, s)
                          : update_xr_impl<double>(h, Bp, damping, s);
  if (rc) return rc;
  if (t) HF_HIP(hipEventRecord(ev[2], s));
  rc = h->dtype == HF_F32 ? update_p_impl<float>(h, nullptr, s)
                          : update_p_impl<double>(h, nullptr, s);
  if (rc) return rc;
  if (t) {
    HF_HIP(hipEventRecord(ev[3], s));
    h->t_count++;
  }
  return HF_OK;
}

// ---- one hipGraph per PCG iteration ---------------------------------------------
// [curvature product (a captured graph of the caller, cloned)] -> K1 -> K2 -> K3 as ONE
// graph launch per iteration.  The K1-K3 nodes are explicit kernel nodes whose
// arguments are refreshed per solve (hf_pcg_graph_update), so the caller's
// per-solve vectors (x, b, snapshot slab, m_hist ...) need not be persistent.  A second
// executable of the same graph carries event-record nodes around K1/K2/K3: the host
// launches it for a sample of the iterations to time the kernels without touching the
// others.
struct hf_pcg_graph {
  hf_pcg* h;
  hipGraph_t graph[2];
  hipGraphExec_t exec[2];
  hipGraphNode_t knode[2][3];
  hipEvent_t ev[4];
  int have_events;
  int timed_in_flight;
};

namespace {
int graph_leaves(hipGraph_t g, std::vector<hipGraphNode_t>& leaves) {
  size_t nn = 0, ne = 0;
  HF_HIP(hipGraphGetNodes(g, nullptr, &nn));
  std::vector<hipGraphNode_t> nodes(nn);
  if (nn) HF_HIP(hipGraphGetNodes(g, nodes.data(), &nn));
  HF_HIP(hipGraphGetEdges(g, nullptr, nullptr, &ne));
  std::vector<hipGraphNode_t> from(ne), to(ne);
  if (ne) HF_HIP(hipGraphGetEdges(g, from.data(), to.data(), &ne));
  leaves.clear();
  for (size_t i = 0; i < nn; ++i) {
    bool has_out = false;
    for (size_t e = 0; e < ne && !has_out; ++e) has_out = from[e] == nodes[i];
    if (!has_out) leaves.push_back(nodes[i]);
  }
  return HF_OK;
}

hipKernelNodeParams node_params(const KLaunch& k) {
  hipKernelNodeParams p;
  memset(&p, 0, sizeof(p));
  p.func = const_cast<void*>(k.func);
  p.gridDim = dim3(k.grid);
  p.blockDim = dim3(BLOCK);
  p.sharedMemBytes = 0;
  p.kernelParams = const_cast<void**>(k.args);
  p.extra = nullptr;
  return p;
}

int build_iteration_graph(hf_pcg_graph* g, int which, hipGraph_t product, const KLaunch (&k)[3]) {
  const bool timed = which == 1;
  if (product) HF_HIP(hipGraphClone(&g->graph[which], product));
  else HF_HIP(hipGraphCreate(&g->graph[which], 0));
  std::vector<hipGraphNode_t> deps;
  const int rc = graph_leaves(g->graph[which], deps);
  if (rc) return rc;
  for (int i = 0; i < 3; ++i) {
    if (timed) {
      hipGraphNode_t e;
      HF_HIP(hipGraphAddEventRecordNode(&e, g->graph[which], deps.data(), deps.size(), g->ev[i]));
      deps.assign(1, e);
    }
    hipKernelNodeParams p = node_params(k[i]);
    HF_HIP(hipGraphAddKernelNode(&g->knode[which][i], g->graph[which], deps.data(), deps.size(), &p));
    deps.assign(1, g->knode[which][i]);
  }
  if (timed) {
    hipGraphNode_t e;
    HF_HIP(hipGraphAddEventRecordNode(&e, g->graph[which], deps.data(), deps.size(), g->ev[3]));
  }
  HF_HIP(hipGraphInstantiate(&g->exec[which], g->graph[which], nullptr, nullptr, 0));
  return HF_OK;
}
}  // namespace

int hf_pcg_graph_create(hf_pcg_graph_t** out, hf_pcg_t* h, void* product_graph, const void* Bp,
                        double damping, int with_timing) {
  if (!out || !h || !Bp) return HF_ERR_ARG;
  if (!h->begun || h->precond == HF_M_EXTERNAL) return HF_ERR_STATE;
  if (!aligned16(Bp)) return HF_ERR_ALIGN;
  hf_pcg_graph* g = new (std::nothrow) hf_pcg_graph();
  if (!g) return HF_ERR_ARG;
  memset(g, 0, sizeof(*g));
  g->h = h;
  KLaunch k[3];
  build_iteration(h, Bp, damping, k);
  int rc = build_iteration_graph(g, 0, (hipGraph_t)product_graph, k);
  if (!rc && with_timing) {
    for (int i = 0; i < 4 && !rc; ++i) rc = (int)hipEventCreate(&g->ev[i]);
    g->have_events = 1;
    if (!rc) rc = build_iteration_graph(g, 1, (hipGraph_t)product_graph, k);
  }
  if (rc) { hf_pcg_graph_destroy(g); return rc; }
  *out = g;
  return HF_OK;
}

int hf_pcg_graph_update(hf_pcg_graph_t* g, const void* Bp, double damping) {
  if (!g || !Bp) return HF_ERR_ARG;
  hf_pcg* h = g->h;
  if (!h->begun || h->precond == HF_M_EXTERNAL) return HF_ERR_STATE;
  if (!aligned16(Bp)) return HF_ERR_ALIGN;
  KLaunch k[3];
  build_iteration(h, Bp, damping, k);
  for (int w = 0; w < 2; ++w) {
    if (!g->exec[w]) continue;
    for (int i = 0; i < 3; ++i) {
      hipKernelNodeParams p = node_params(k[i]);
      HF_HIP(hipGraphExecKernelNodeSetParams(g->exec[w], g->knode[w][i], &p));
    }
  }
  return HF_OK;
}

int hf_pcg_graph_launch(hf_pcg_graph_t* g, int timed, void* stream) {
  if (!g) return HF_ERR_ARG;
  if (!g->h->inited) return HF_ERR_STATE;
  const int w = (timed && g->exec[1]) ? 1 : 0;
  HF_HIP(hipGraphLaunch(g->exec[w], (hipStream_t)stream));
  if (w == 1) g->timed_in_flight = 1;
  return HF_OK;
}

int hf_pcg_graph_collect_timing(hf_pcg_graph_t* g) {
  if (!g) return HF_ERR_ARG;
  if (!g->timed_in_flight) return HF_OK;
  HF_HIP(hipEventSynchronize(g->ev[3]));
  for (int i = 0; i < 3; ++i) {
    float t = 0;
    HF_HIP(hipEventElapsedTime(&t, g->ev[i], g->ev[i + 1]));
    g->h->g_ms[i] += t;
  }
  g->h->g_count++;
  g->timed_in_flight = 0;
  return HF_OK;
}

int hf_pcg_graph_destroy(hf_pcg_graph_t* g) {
  if (!g) return HF_OK;
  for (int w = 0; w < 2; ++w) {
    if (g->exec[w]) (void)hipGraphExecDestroy(g->exec[w]);
    if (g->graph[w]) (void)hipGraphDestroy(g->graph[w]);
  }
  if (g->have_events)
    for (int i = 0; i < 4; ++i)
      if (g->ev[i]) (void)hipEventDestroy(g->ev[i]);
  delete g;
  return HF_OK;
}

static void fill_status(const hf_pcg* h, const DevState* st, hf_pcg_status* out) {
  out->done = st->done;
  out->reason = st->done;
  out->n_iters = st->n_iters;
  out->iter_next = st->iter_next;
  out->nonpos_count = st->nonpos_count;
  out->last_alpha = st->last_alpha;
  out->last_beta = st->last_beta;
  out->last_pAp = st->last_pAp;
  out->last_res_norm = st->last_res_norm;
  out->res_bound = st->res_bound;
  out->n_stored = st->slot_next;
  (void)h;
}

int hf_pcg_poll(hf_pcg_t* h, hf_pcg_status* out) {
  if (!h || !out) return HF_ERR_ARG;
  memset(out, 0, sizeof(*out));
  const int f = *(volatile int*)h->h_flag;
  out->done = f;
  out->reason = f;
  out->n_iters = f ? ((volatile int*)h->h_flag)[1] : 0;  // terminating iteration
  return HF_OK;
}

int hf_pcg_finish(hf_pcg_t* h, hf_pcg_status* out, void* stream) {
  if (!h || !out) return HF_ERR_ARG;
  if (!h->inited) return HF_ERR_STATE;
  hipStream_t s = (hipStream_t)stream;
  HF_HIP(hipMemcpyAsync(h->h_state, h->d_state, sizeof(DevState), hipMemcpyDeviceToHost, s));
  HF_HIP(hipStreamSynchronize(s));
  fill_status(h, h->h_state, out);
  h->finished = 1;
  return HF_OK;
}

int hf_pcg_read_nonpos(hf_pcg_t* h, int64_t* iters, double* values, int cap) {
  if (!h || !iters || !values || cap < 0) return HF_ERR_ARG;
  if (!h->finished) return HF_ERR_STATE;
  int64_t c = h->h_state->nonpos_count;
  if (c > NP_CAP) c = NP_CAP;
  if (c > cap) c = cap;
  for (int64_t i = 0; i < c; ++i) {
    iters[i] = h->h_state->nonpos_iter[i];
    values[i] = h->h_state->nonpos_val[i];
  }
  return (int)c;
}

int hf_pcg_timing_enable(hf_pcg_t* h, int enable) {
  if (!h) return HF_ERR_ARG;
  if (enable && !h->ev) {
    h->ev = new (std::nothrow) hipEvent_t[4 * TIMING_CAP];
    if (!h->ev) return HF_ERR_ARG;
    for (int i = 0; i < 4 * TIMING_CAP; ++i) HF_HIP(hipEventCreate(&h->ev[i]));
  }
  h->timing = enable ? 1 : 0;
  h->t_count = 0;
  h->g_ms[0] = h->g_ms[1] = h->g_ms[2] = 0.0;
  h->g_count = 0;
  return HF_OK;
}

int hf_pcg_timing_read(hf_pcg_t* h, double* ms_k1, double* ms_k2, double* ms_k3,
                       int64_t* n_recorded) {
  if (!h || !ms_k1 || !ms_k2 || !ms_k3 || !n_recorded) return HF_ERR_ARG;
  double a = 0, b = 0, c = 0;
  for (int64_t i = 0; i < h->t_count; ++i) {
    float t = 0;
    hipEvent_t* ev = h->ev + 4 * i;
    HF_HIP(hipEventElapsedTime(&t, ev[0], ev[1])); a += t;
    HF_HIP(hipEventElapsedTime(&t, ev[1], ev[2])); b += t;
    HF_HIP(hipEventElapsedTime(&t, ev[2], ev[3])); c += t;
  }
  a += h->g_ms[0]; b += h->g_ms[1]; c += h->g_ms[2];  // sampled hf_pcg_graph launches
  const int64_t cnt = h->t_count + h->g_count;
  const double k = cnt > 0 ? 1.0 / (double)cnt : 0.0;
  *ms_k1 = a * k; *ms_k2 = b * k; *ms_k3 = c * k;
  *n_recorded = cnt;
  return HF_OK;
}

int hf_precond_build(void* minv, const void* diag, double damping, double exponent, int64_t n,
                     int dtype, void* stream) {
  if (!minv || !diag || n <= 0) return HF_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == HF_F32)
    hipLaunchKernelGGL((k_precond_build<float>), dim3(small_grid(n)), dim3(BLOCK), 0, s,
                       (float*)minv, (const float*)diag, (float)damping, (float)(-exponent),
                       (long long)n);
  else if (dtype == HF_F64)
    hipLaunchKernelGGL((k_precond_build<double>), dim3(small_grid(n)), dim3(BLOCK), 0, s,
                       (double*)minv, (const double*)diag, damping, -exponent, (long long)n);
  else
    return HF_ERR_ARG;
  HF_HIP(hipGetLastError());
  return HF_OK;
}

int hf_axpy_out(void* out, const void* a, const void* sv, double alpha, int64_t n, int dtype,
                void* stream) {
  if (!out || !a || !sv || n <= 0) return HF_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int vec_ok = aligned16(out) && aligned16(a) && aligned16(sv);
  if (dtype == HF_F32)
    hipLaunchKernelGGL((k_axpy_out<float>), dim3(small_grid(n / 4 + 1)), dim3(BLOCK), 0, s,
                       (float*)out, (const float*)a, (const float*)sv, (float)alpha,
                       (long long)n, vec_ok);
  else if (dtype == HF_F64)
    hipLaunchKernelGGL((k_axpy_out<double>), dim3(small_grid(n / 2 + 1)), dim3(BLOCK), 0, s,
                       (double*)out, (const double*)a, (const double*)sv, alpha, (long long)n,
                       vec_ok);
  else
    return HF_ERR_ARG;
  HF_HIP(hipGetLastError());
  return HF_OK;
}

