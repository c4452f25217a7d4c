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

#include "hf_pcg.h"
#include "hf_unpack.h"

namespace {

constexpr int BLOCK = 256;          // 4 waves of 64
constexpr int WAVES = BLOCK / 64;
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

using hf_shared::VecOf;
using hf_shared::VU;

// ---------------------------------------------------------------------------
// reductions: 64-lane __shfl_down tree -> LDS partial per wave -> fixed-order sum
// ---------------------------------------------------------------------------
template <int K, int NW = WAVES>
__device__ __forceinline__ void block_allreduce(double (&v)[K], double* lds /*K*NW*/) {
#pragma unroll
  for (int k = 0; k < K; ++k) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_down(v[k], off, 64);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) lds[k * NW + wave] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < K; ++k) {
    double s = lds[k * NW];
#pragma unroll
    for (int w = 1; w < NW; ++w) s += lds[k * NW + w];
    v[k] = s;
  }
  __syncthreads();
}

// Every block re-reduces the previous kernel's per-block partials (layout
// part[k*stride + block]) in the same order.
template <int K>
__device__ __forceinline__ void reduce_partials(const double* __restrict__ part, int nparts,
                                                int stride, double (&out)[K], double* lds) {
#pragma unroll
  for (int k = 0; k < K; ++k) out[k] = 0.0;
  for (int i = threadIdx.x; i < nparts; i += BLOCK) {
#pragma unroll
    for (int k = 0; k < K; ++k) out[k] += part[k * stride + i];
  }
  block_allreduce<K>(out, lds);
}

template <int K>
__device__ __forceinline__ void write_partials(double* __restrict__ part, int stride,
                                               double (&v)[K], double* lds) {
  block_allreduce<K>(v, lds);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) part[k * stride + blockIdx.x] = v[k];
  }
}

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
// of >= HF_PCG_NT_MIN elements, default 16 M = six fp32 vectors of 1.5 x the 256 MiB Infinity Cache).  Beyond the
// cache nothing these kernels read survives until its next use, and non-temporal loads / stores stream faster
// (profiles/r04_pcg_nt_variants.jsonl, N = 100 M: K1 151.8 -> 137.2, K2 466.9 -> 440.0 us; all three kernels 0.70 ->
// 0.77 of 8 TB/s; N = 25.6 M: 0.71 -> 0.79).  At the ResNet-18 size (six vectors = 256 MiB, the cache's edge) the
// same-box A/B of the whole bench is inside its own noise (+1.8 % / +1.8 % in one batch, -3.0 % / +1.5 % in the
// next) while K2 / K3 themselves get ~1 us slower: default policy there, and for small vectors (All-CNN-C: 5.5 MB,
// L2-resident between kernels).
// NT covers: K1 both read streams; K2 the x / b / Bp loads and the x store (r and p are re-read by K3 right
// after).  K3 keeps the default policy: its r was written by K2 a moment ago and is still on-die (a non-temporal
// r load cost K3 24.8 -> 26.9 us at N = 11.2 M).
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
      vp[u].v = reinterpret_cast<const V*>(p)[i];                           \
      if (MODE == HF_M_DIAG) vm[u].v = reinterpret_cast<const V*>(minv)[i]; \
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
        reinterpret_cast<V*>(p)[i] = vp[u].v;
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

// ---------------------------------------------------------------------------
// multi-tensor gather (pointer table passed by value)
// ---------------------------------------------------------------------------
using hf_shared::PACK_MAXT;
using hf_shared::PACK_CHUNK;
struct PackArgs {
  const void* src[PACK_MAXT];
  long long dst_off[PACK_MAXT];
  long long numel[PACK_MAXT];
  int blk_start[PACK_MAXT + 1];
  // channels_last 4-D sources [O, I, H, W] stored as (O, H, W, I): inner channel count I
  // and HW = H*W; 0 = plain contiguous.  The gather un-permutes while it copies.
  int perm_I[PACK_MAXT];
  int perm_HW[PACK_MAXT];
  int chunk[PACK_MAXT];  // elements per block of tensor t (a whole number of [I, HW] slabs when tiled)
  // split-K partial results: the source is the SUM of nsplit[t] arrays, split_stride[t] elements apart
  // (the weight gradients of hf_conv2d_nhwc_*_slabs: combined here, in split order, while gathering)
  int nsplit[PACK_MAXT];
  long long split_stride[PACK_MAXT];
  // permuted sources only, HW <= 16: bit hw set = kernel tap hw can meet data; the other taps'
  // gradients are structurally zero (3x3 kernels on 1x1 / 2x2 maps) and are written as zeros
  // without being read.  0 = every tap is read.
  unsigned short live[PACK_MAXT];
  int nt;
};
constexpr int TILE_BYTES = 32768;  // LDS staging of the layout-permuting paths

template <typename T, int OP>
__device__ __forceinline__ T pack_op(T d, T s, T scale) {
  if (OP == 0) return (T)(scale * s);
  const T g = (T)(scale * s);
  return d + (T)(g * g);
}

template <typename T, int OP>
__global__ __launch_bounds__(BLOCK) void k_pack(T* __restrict__ dst, const PackArgs a, T scale) {
  constexpr int W = VecOf<T>::W;
  typedef typename VecOf<T>::type V;
  constexpr unsigned TILE = TILE_BYTES / sizeof(T);
  __shared__ __attribute__((aligned(16))) T tile[TILE];  // staging of the layout-permuting paths
  // binary search: tensor t with blk_start[t] <= blockIdx.x < blk_start[t+1]
  int lo = 0, hi = a.nt;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (a.blk_start[mid] <= (int)blockIdx.x) lo = mid; else hi = mid;
  }
  const T* __restrict__ src = reinterpret_cast<const T*>(a.src[lo]);
  const int nsp = a.nsplit[lo];
  const long long numel = a.numel[lo];
  const long long j0 = (long long)(blockIdx.x - a.blk_start[lo]) * a.chunk[lo];
  const long long j1 = (j0 + a.chunk[lo] < numel) ? j0 + a.chunk[lo] : numel;
  T* __restrict__ out = dst + a.dst_off[lo];
  if (nsp > 1) {
    // source = sum of nsp split-K slabs (the weight gradients of layers whose reduction had to
    // be split, the BatchNorm adjoint's per-row-block sums); combined in split order.  Walked
    // in SOURCE order (coalesced loads, 4 elements x 2 slabs in flight per lane: these blocks
    // are latency-bound), un-permuted on the store side.
    const long long sps = a.split_stride[lo];
    const unsigned I = (unsigned)a.perm_I[lo], HW = (unsigned)a.perm_HW[lo], slab = I * HW;
    const unsigned live = a.live[lo];
    if (sizeof(T) == 4 && (((uintptr_t)src) & 15) == 0 && (sps & 3) == 0 && (j0 & 3) == 0 && (numel & 3) == 0 &&
        (I & 3) == 0) {
      // 16-byte loads: one quad of consecutive source elements per lane and pass (a quad never leaves its
      // (o, hw) row: I % 4 == 0), eight slabs in flight; dword loads moved these 50 MB at 3 TB/s
      const unsigned j1u = (unsigned)j1;
      // whole slabs per block (host: chunk = a few slabs): the permuted order is assembled in LDS and leaves as
      // 16-byte stores -- four 4-byte stores per lane, 144 bytes apart across the lanes, cost more L2
      // transactions than the loads they follow
      const bool staged = I > 0 && (unsigned)a.chunk[lo] % slab == 0 && (unsigned)a.chunk[lo] <= TILE &&
                          (((uintptr_t)(out + j0)) & 15) == 0;
      for (unsigned e = (unsigned)j0 + threadIdx.x * 4; e < j1u; e += BLOCK * 4) {
        unsigned jd = e, step = 1;  // destination of the quad's first element, distance between its elements
        bool rd = true;
        if (I > 0) {
          const unsigned o = e / slab, rem = e - o * slab;
          const unsigned hw = rem / I, i = rem - hw * I;
          jd = o * slab + i * HW + hw;
          step = HW;
          if (live) rd = (live >> hw) & 1u;
        }
        VU<T> acc;
#pragma unroll
        for (int c = 0; c < W; ++c) acc.e[c] = (T)0;
        if (rd) acc.v = *reinterpret_cast<const V*>(src + e);
        for (int sp = 1; sp < nsp; sp += 8) {
          VU<T> tt[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int c = 0; c < W; ++c) tt[u].e[c] = (T)0;
            if (rd && sp + u < nsp) tt[u].v = *reinterpret_cast<const V*>(src + e + (long long)(sp + u) * sps);
          }
#pragma unroll
          for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < W; ++c) acc.e[c] += tt[u].e[c];
        }
        if (staged) {
          // Lanes of a wave hold consecutive input-channel quads of one (o, hw) row: their LDS words are
          // 4*HW = 36 apart for a 3x3 kernel, i.e. lanes l, l+16, l+32, l+48 met in one bank (4-way conflicts:
          // two thirds of this kernel's LDS cycles, profiles/r03_engine_kernel_counters.json).  Their word
          // indices differ by multiples of 9*64, so bits 6..7 tell them apart: XOR those into the position
          // INSIDE the 16-byte quad -- quads stay whole and aligned for the 16-byte reads below, which undo
          // the swap in registers.
#pragma unroll
          for (int c = 0; c < W; ++c) {
            const unsigned q = jd - (unsigned)j0 + c * step;
            tile[(q & ~3u) | ((q ^ (q >> 6)) & 3u)] = acc.e[c];
          }
        } else if (I == 0 && OP == 0 && (((uintptr_t)(out + jd)) & 15) == 0) {
#pragma unroll
          for (int c = 0; c < W; ++c) acc.e[c] = pack_op<T, OP>((T)0, acc.e[c], scale);
          *reinterpret_cast<V*>(out + jd) = acc.v;
        } else {
#pragma unroll
          for (int c = 0; c < W; ++c) out[jd + c * step] = pack_op<T, OP>(out[jd + c * step], acc.e[c], scale);
        }
      }
      if (staged) {
        __syncthreads();
        const unsigned len = j1u - (unsigned)j0;  // (a multiple of 4: whole slabs, I % 4 == 0)
        for (unsigned t = threadIdx.x * 4; t < len; t += BLOCK * 4) {
          VU<T> v, d;
          v.v = *reinterpret_cast<const V*>(tile + t);
          if constexpr (W == 4) {  // undo the in-quad swap of the staging stores (sw = bits 6..7 of the word index)
            const unsigned sw = (t >> 6) & 3u;
            T e0 = v.e[0], e1 = v.e[1], e2 = v.e[2], e3 = v.e[3];
            if (sw & 1u) { T x0 = e0; e0 = e1; e1 = x0; T x2 = e2; e2 = e3; e3 = x2; }
            if (sw & 2u) { T x0 = e0; e0 = e2; e2 = x0; T x1 = e1; e1 = e3; e3 = x1; }
            v.e[0] = e0; v.e[1] = e1; v.e[2] = e2; v.e[3] = e3;
          }
          if (OP == 1) d.v = *reinterpret_cast<const V*>(out + j0 + t);
#pragma unroll
          for (int c = 0; c < W; ++c) v.e[c] = pack_op<T, OP>(OP == 1 ? d.e[c] : (T)0, v.e[c], scale);
          *reinterpret_cast<V*>(out + j0 + t) = v.v;
        }
      }
      return;
    }
    constexpr int E = 4;
    // (element indices of one tensor fit 32 bits -- checked on the host: the per-element divisions below are
    // 32-bit, a 64-bit division is ~5x the instructions and these blocks were VALU-bound on them)
    const unsigned j1u = (unsigned)j1;
    for (unsigned base = (unsigned)j0 + threadIdx.x; base < j1u; base += BLOCK * E) {
      T acc[E];
      unsigned e[E];
      bool rd[E];  // inside the tensor and not a structurally-zero tap
#pragma unroll
      for (int k = 0; k < E; ++k) {
        e[k] = base + (unsigned)k * BLOCK;
        rd[k] = e[k] < j1u;
        if (live && rd[k]) rd[k] = (live >> ((e[k] % slab) / I)) & 1u;
        acc[k] = rd[k] ? src[e[k]] : (T)0;
      }
      int sp = 1;
      // (weight gradients of large-map layers arrive as up to 128 slabs: eight slabs x E elements in
      // flight per lane, added in split order -- two at a time cost one round trip per pair)
      for (; sp + 8 <= nsp; sp += 8) {
        T tt[8][E];
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int k = 0; k < E; ++k) tt[u][k] = rd[k] ? src[e[k] + (long long)(sp + u) * sps] : (T)0;
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int k = 0; k < E; ++k) acc[k] += tt[u][k];
      }
      if (sp < nsp) {
        // the last (partial) batch, predicated: all its loads in flight at once (pairs cost a round trip each)
        T tt[8][E];
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int k = 0; k < E; ++k) tt[u][k] = (rd[k] && sp + u < nsp) ? src[e[k] + (long long)(sp + u) * sps] : (T)0;
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int k = 0; k < E; ++k) acc[k] += tt[u][k];
      }
#pragma unroll
      for (int k = 0; k < E; ++k) {
        if (e[k] >= j1u) continue;
        unsigned j = e[k];
        if (I > 0) {  // source (o, hw, i) -> destination (o, i, hw)
          const unsigned o = j / slab;
          const unsigned rem = j - o * slab;
          const unsigned hw = rem / I, i = rem - hw * I;
          j = o * slab + i * HW + hw;
        }
        out[j] = pack_op<T, OP>(out[j], acc[k], scale);
      }
    }
    return;
  }
  if (a.perm_I[lo] > 0 && a.live[lo] != 0) {
    // mostly structural zeros (a 3x3 kernel on a 1x1 map: 8 of 9 entries): walk the DESTINATION
    // in 16-byte vectors, fetch only the live entries (dst (o, i, hw) <- src (o, hw, i)); no LDS
    // staging, no barrier -- the block is a stream of vector stores
    const unsigned I = (unsigned)a.perm_I[lo], HW = (unsigned)a.perm_HW[lo], slab = I * HW;
    const unsigned live = a.live[lo];
    const bool al = (((uintptr_t)(out + j0)) & 15) == 0;
    const unsigned j1u = (unsigned)j1;
    if (OP == 0 && al && ((unsigned)a.chunk[lo] % slab) == 0 && ((j1u - (unsigned)j0) & (W - 1)) == 0) {
      // whole (o) slabs per block: a pure stream of zero vectors over the chunk (no index arithmetic: the
      // per-element divisions of the walk below held this 33 MB store stream at 2.3 TB/s), then, behind a
      // barrier, the live taps' values on top -- source order, coalesced reads, 4-byte stores into lines this
      // workgroup has just written
      VU<T> z;
#pragma unroll
      for (int c = 0; c < W; ++c) z.e[c] = (T)0;
      for (unsigned j = (unsigned)j0 + threadIdx.x * W; j < j1u; j += BLOCK * W) *reinterpret_cast<V*>(out + j) = z.v;
      __syncthreads();  // (s_waitcnt vmcnt(0) + barrier: the zeros are acknowledged before any value store is issued)
      const unsigned o0 = (unsigned)j0 / slab, no = (j1u - (unsigned)j0) / slab;
      const unsigned nl = (unsigned)__popc(live);
      const unsigned per_o = nl * I, total = no * per_o;
      for (unsigned q = threadIdx.x; q < total; q += BLOCK) {
        const unsigned ol = q / per_o, rem = q - ol * per_o;
        const unsigned l = rem / I, i = rem - l * I;
        unsigned hw = 0, seen = 0;  // the l-th live tap (registers only: an indexed local array would go to scratch)
#pragma unroll
        for (unsigned t = 0; t < 16; ++t) {
          const unsigned bit = (live >> t) & 1u;
          hw = (bit && seen == l) ? t : hw;
          seen += bit;
        }
        const unsigned ob = (o0 + ol) * slab;
        out[ob + i * HW + hw] = pack_op<T, OP>((T)0, src[ob + hw * I + i], scale);
      }
      return;
    }
    for (unsigned j = (unsigned)j0 + threadIdx.x * W; j < j1u; j += BLOCK * W) {
      VU<T> v;
#pragma unroll
      for (int c = 0; c < W; ++c) {
        const unsigned jj = j + c;
        const unsigned o = jj / slab;
        const unsigned rem = jj - o * slab;
        const unsigned i = rem / HW, hw = rem - i * HW;
        v.e[c] = (jj < j1u && ((live >> hw) & 1u)) ? src[o * slab + hw * I + i] : (T)0;
      }
      if (OP == 0 && al && j + W <= j1u) {
#pragma unroll
        for (int c = 0; c < W; ++c) v.e[c] = pack_op<T, OP>((T)0, v.e[c], scale);
        *reinterpret_cast<V*>(out + j) = v.v;
      } else {
#pragma unroll
        for (int c = 0; c < W; ++c)
          if (j + c < j1u) out[j + c] = pack_op<T, OP>(out[j + c], v.e[c], scale);
      }
    }
    return;
  }
  if (a.perm_I[lo] > 0) {
    // dst index j = (o*I + i)*HW + hw   <-   src index (o*HW + hw)*I + i
    const unsigned I = (unsigned)a.perm_I[lo], HW = (unsigned)a.perm_HW[lo], slab = I * HW;
    if ((unsigned)a.chunk[lo] % slab == 0 && (unsigned)a.chunk[lo] / slab * (slab + HW) <= TILE) {
      // whole slabs per block: read them contiguously into LDS (rows of I padded to I+1
      // against bank conflicts), write the permuted order contiguously
      const unsigned len = (unsigned)(j1 - j0);
      for (unsigned t = threadIdx.x; t < len; t += BLOCK) {
        const unsigned row = t / I;  // (o_local*HW + hw)
        tile[row * (I + 1) + (t - row * I)] = src[j0 + t];
      }
      __syncthreads();
      for (unsigned t = threadIdx.x; t < len; t += BLOCK) {
        const unsigned ol = t / slab, rem = t - ol * slab;
        const unsigned i = rem / HW, hw = rem - i * HW;
        out[j0 + t] = pack_op<T, OP>(out[j0 + t], tile[(ol * HW + hw) * (I + 1) + i], scale);
      }
      return;
    }
    for (long long j = j0 + threadIdx.x; j < j1; j += BLOCK) {
      const long long o = j / slab;
      const unsigned rem = (unsigned)(j - o * slab);
      const unsigned i = rem / HW, hw = rem - i * HW;
      out[j] = pack_op<T, OP>(out[j], src[o * slab + (long long)hw * I + i], scale);
    }
    return;
  }
  const bool vec_ok = ((((uintptr_t)src) | ((uintptr_t)out)) & 15) == 0;
  if (vec_ok) {
    const long long v0 = j0 / W, v1 = j1 / W;
    for (long long i = v0 + threadIdx.x; i < v1; i += BLOCK) {
      VU<T> s, d;
      s.v = reinterpret_cast<const V*>(src)[i];
      if (OP == 1) d.v = reinterpret_cast<const V*>(out)[i];
#pragma unroll
      for (int c = 0; c < W; ++c) d.e[c] = pack_op<T, OP>(d.e[c], s.e[c], scale);
      reinterpret_cast<V*>(out)[i] = d.v;
    }
    for (long long j = v1 * W + threadIdx.x; j < j1; j += BLOCK)
      out[j] = pack_op<T, OP>(out[j], src[j], scale);
  } else {
    for (long long j = j0 + threadIdx.x; j < j1; j += BLOCK)
      out[j] = pack_op<T, OP>(out[j], src[j], scale);
  }
}

// Multi-tensor scatter for the tangent sweep (inverse of the gather above): argument block and device body in
// hf_unpack.h (shared with hf_conv.hip).
using hf_shared::UnpackArgs;

template <typename T>
__global__ __launch_bounds__(BLOCK) void k_unpack_tangent(const T* __restrict__ src_base, const UnpackArgs a) {
  __shared__ T tile[hf_shared::TT * (hf_shared::TT + 1)];  // (transposed copies only)
  if (hf_shared::unpack_transposed_block<T>(src_base, a, blockIdx.x, tile)) return;
  hf_shared::unpack_block<T>(src_base, a, blockIdx.x);
}

// Compaction of a flat parameter-space vector to its entries that can be non-zero, and back
// (data-parallel products: only those travel through the all-reduce).  The vector is a sequence
// of segments: dense ones, and conv weights [O, I, H*W] of which only the kernel taps in `mask`
// are live (period HW, nl = popcount(mask) live entries per period, pos[] their tap indices).
constexpr int LIVE_MAXS = 24;
constexpr int LIVE_CHUNK = BLOCK * 8;  // compact entries per block
struct LiveSegs {
  long long full_off[LIVE_MAXS];
  long long comp_off[LIVE_MAXS + 1];  // compact offsets; [ns] = total
  int blk_start[LIVE_MAXS + 1];       // blocks never straddle segments: the segment look-up is per block
  int hw[LIVE_MAXS];                  // 0: dense
  int nl[LIVE_MAXS];
  int pos[LIVE_MAXS][16];             // (ints: a byte table in the kernel arguments is read with vector loads)
  int ns;
};

// SCATTER = false: comp[k] = full[index(k)];  true: full[index(k)] = comp[k]
template <typename T, bool SCATTER>
__global__ __launch_bounds__(BLOCK) void k_live_copy(T* __restrict__ full, T* __restrict__ comp,
                                                     const LiveSegs a) {
  int lo = 0, hi = a.ns;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (a.blk_start[mid] <= (int)blockIdx.x) lo = mid; else hi = mid;
  }
  const long long len = a.comp_off[lo + 1] - a.comp_off[lo];
  const long long r0 = (long long)((int)blockIdx.x - a.blk_start[lo]) * LIVE_CHUNK;
  const unsigned cnt = (unsigned)(len - r0 < LIVE_CHUNK ? len - r0 : LIVE_CHUNK);
  T* __restrict__ c = comp + a.comp_off[lo] + r0;
  const unsigned hw = (unsigned)a.hw[lo], nl = (unsigned)a.nl[lo];
  if (hw == 0) {
    T* __restrict__ f = full + a.full_off[lo] + r0;
    constexpr int W = VecOf<T>::W;
    typedef typename VecOf<T>::type V;
    if ((((uintptr_t)f | (uintptr_t)c) & 15) == 0) {
      const unsigned nv = cnt / W;
      for (unsigned i = threadIdx.x; i < nv; i += BLOCK) {
        if (SCATTER) reinterpret_cast<V*>(f)[i] = reinterpret_cast<const V*>(c)[i];
        else reinterpret_cast<V*>(c)[i] = reinterpret_cast<const V*>(f)[i];
      }
      for (unsigned i = nv * W + threadIdx.x; i < cnt; i += BLOCK) {
        if (SCATTER) f[i] = c[i]; else c[i] = f[i];
      }
    } else {
      for (unsigned i = threadIdx.x; i < cnt; i += BLOCK) {
        if (SCATTER) f[i] = c[i]; else c[i] = f[i];
      }
    }
    return;
  }
  // periodic: compact entry r = g*nl + l  <->  full entry g*hw + pos[l]
  T* __restrict__ f = full + a.full_off[lo];
  const unsigned r0u = (unsigned)r0;  // (one segment's compact length fits 32 bits: host check)
  if (nl == 1) {
    f += (size_t)r0u * hw + a.pos[lo][0];
    for (unsigned i = threadIdx.x; i < cnt; i += BLOCK) {
      if (SCATTER) f[(size_t)i * hw] = c[i]; else c[i] = f[(size_t)i * hw];
    }
    return;
  }
  for (unsigned i = threadIdx.x; i < cnt; i += BLOCK) {
    const unsigned r = r0u + i, g = r / nl, l = r - g * nl;
    T* q = f + (size_t)g * hw + a.pos[lo][l];
    if (SCATTER) *q = c[i]; else c[i] = *q;
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

// ---------------------------------------------------------------------------
// eval-mode BatchNorm (+ residual add, + ReLU) as a per-channel affine map, fused
// (curvature-product path).   xhat = (x - mean[c]) * rstd[c]
//   k_chan_affine     : t = a*(w[c]*rstd[c]) + xhat*q[c] + r[c] + add   (each term optional)
//                       out = relu_self ? max(t, 0) : (mask_src ? (mask_src > 0 ? t : 0) : t)
//       forward  y = act(xhat*w + b + res)      (q = w, r = b, add = res, relu_self)
//       tangent / transpose of the backward map (a = v_gx, q = v_gw, r = v_gb,
//                                                add = v_gres, mask_src = y)
//   k_chan_affine_bwd : g = mask_src ? gy*(mask_src > 0) : gy
//                       gx = g*w[c]*rstd[c] ; gw[c] = sum g*xhat ; gb[c] = sum g ; gres = g
// One launch each instead of the ~16 small ATen kernels autograd's generic
// double-backward of batch_norm (+2 for the add, +2 for the ReLU) issues per layer
// and product.  NCHW-contiguous.
// ---------------------------------------------------------------------------
// I = unsigned (tensors < 2^31 elements: 32-bit index arithmetic, the per-element
// division is what these tiny kernels spend their time on) or long long.
template <typename T, typename I>
__device__ __forceinline__ void chan_affine_body(
    T* __restrict__ out, const T* __restrict__ a, const T* __restrict__ x,
    const T* __restrict__ mean, const T* __restrict__ rstd, const T* __restrict__ w,
    const T* __restrict__ q, const T* __restrict__ r, const T* __restrict__ add,
    const T* __restrict__ mask_src, int relu_self, I total, I C, I HW, int nhwc, I out_ld,
    I add_ld, int a_splits, long long a_slab, I bid, I nblocks) {
  // out_ld / add_ld != 0: that operand is the first-C-channels slice of a wider buffer --
  // NHWC: element (row, c) at row*ld + c; NCHW: (n, c, hw) at n*ld + c*HW + hw.
  const I CHW = C * HW;
  for (I i = bid * BLOCK + threadIdx.x; i < total; i += nblocks * BLOCK) {
    const I c = (nhwc || HW == 1 ? i : i / HW) % C;  // NHWC: the channel is the fastest index
    const T rs = rstd ? rstd[c] : (T)1;  // (no BatchNorm: conv + bias layers of plain stacks)
    T acc = (T)0;
    if (a) {
      T av = a[i];
      for (int sp = 1; sp < a_splits; sp += 8) {  // split-K slabs: eight loads in flight, summed in split order
        T t8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t8[u] = a[(long long)(sp + u < a_splits ? sp + u : 0) * a_slab + i];
#pragma unroll
        for (int u = 0; u < 8; ++u) av += sp + u < a_splits ? t8[u] : (T)0;
      }
      acc = av * ((w ? w[c] : (T)1) * rs);
    }
    if (q) acc += ((x[i] - mean[c]) * rs) * q[c];
    if (r) acc += r[c];
    I outer = 0;
    if (out_ld | add_ld) outer = nhwc ? i / C : i / CHW;  // row resp. sample
    if (add) acc += add[add_ld ? i + outer * (add_ld - (nhwc ? C : CHW)) : i];
    if (relu_self) acc = acc > (T)0 ? acc : (T)0;
    else if (mask_src) acc = mask_src[i] > (T)0 ? acc : (T)0;
    out[out_ld ? i + outer * (out_ld - (nhwc ? C : CHW)) : i] = acc;
  }
}

// fp32 NHWC, C % 4 == 0, every operand 16-byte aligned: one 16-byte channel quad per thread, and EVERY load
// of the quad -- per-channel vectors, x / add / mask, up to 17 split-K slabs -- issued before the first use.
// These launches move a few MB and take ~5 us: what they cost is dependent round trips (~0.6 us each from the
// memory-side cache the producer's slabs sit in), not bytes; the scalar walk above paid one per slab batch of
// eight, one for w[c], one for x / q / r, one for add, one for the mask.  Same expressions, same order of
// additions: bitwise the scalar walk's results.
struct alignas(16) F4 { float e[4]; };

__device__ __forceinline__ F4 ld4(const float* p) { return *reinterpret_cast<const F4*>(p); }

__device__ __forceinline__ void chan_affine_v4_body(
    float* __restrict__ out, const float* __restrict__ a, const float* __restrict__ x,
    const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ w,
    const float* __restrict__ q, const float* __restrict__ r, const float* __restrict__ add,
    const float* __restrict__ mask_src, int relu_self, unsigned total, unsigned C, unsigned out_ld,
    unsigned add_ld, int a_splits, long long a_slab, unsigned bid, unsigned nblocks) {
  const unsigned quads = total >> 2;
  for (unsigned v = bid * BLOCK + threadIdx.x; v < quads; v += nblocks * BLOCK) {
    const unsigned i = v << 2;
    const unsigned row = i / C, c = i - row * C;
    F4 rs4, w4, q4, r4, mu4, xv, addv, mv, av;
    if (rstd) rs4 = ld4(rstd + c);
    if (w) w4 = ld4(w + c);
    if (q) { q4 = ld4(q + c); mu4 = ld4(mean + c); xv = ld4(x + i); }
    if (r) r4 = ld4(r + c);
    if (add) addv = ld4(add + (add_ld ? row * add_ld + c : i));
    if (mask_src && !relu_self) mv = ld4(mask_src + i);
    if (a) {
      av = ld4(a + i);
      for (int sp = 1; sp < a_splits; sp += 16) {  // split-K slabs: sixteen loads in flight, summed in split order
        F4 t[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) t[u] = ld4(a + (long long)(sp + u < a_splits ? sp + u : 0) * a_slab + i);
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
          for (int k = 0; k < 4; ++k) av.e[k] += sp + u < a_splits ? t[u].e[k] : 0.f;
      }
    }
    F4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float rs = rstd ? rs4.e[k] : 1.f;
      float acc = 0.f;
      if (a) acc = av.e[k] * ((w ? w4.e[k] : 1.f) * rs);
      if (q) acc += ((xv.e[k] - mu4.e[k]) * rs) * q4.e[k];
      if (r) acc += r4.e[k];
      if (add) acc += addv.e[k];
      if (relu_self) acc = acc > 0.f ? acc : 0.f;
      else if (mask_src) acc = mv.e[k] > 0.f ? acc : 0.f;
      o.e[k] = acc;
    }
    *reinterpret_cast<F4*>(out + (out_ld ? row * out_ld + c : i)) = o;
  }
}

__global__ __launch_bounds__(BLOCK) void k_chan_affine_v4(
    float* __restrict__ out, const float* __restrict__ a, const float* __restrict__ x,
    const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ w,
    const float* __restrict__ q, const float* __restrict__ r, const float* __restrict__ add,
    const float* __restrict__ mask_src, int relu_self, unsigned total, unsigned C, unsigned out_ld,
    unsigned add_ld, int a_splits, long long a_slab) {
  chan_affine_v4_body(out, a, x, mean, rstd, w, q, r, add, mask_src, relu_self, total, C, out_ld, add_ld, a_splits,
                      a_slab, blockIdx.x, gridDim.x);
}

template <typename T, typename I>
__global__ __launch_bounds__(BLOCK) void k_chan_affine(
    T* __restrict__ out, const T* __restrict__ a, const T* __restrict__ x,
    const T* __restrict__ mean, const T* __restrict__ rstd, const T* __restrict__ w,
    const T* __restrict__ q, const T* __restrict__ r, const T* __restrict__ add,
    const T* __restrict__ mask_src, int relu_self, I total, I C, I HW, int nhwc, I out_ld,
    I add_ld, int a_splits, long long a_slab) {
  chan_affine_body<T, I>(out, a, x, mean, rstd, w, q, r, add, mask_src, relu_self, total, C, HW, nhwc, out_ld,
                         add_ld, a_splits, a_slab, (I)blockIdx.x, (I)gridDim.x);
}

// Two independent layers (a residual block's first BatchNorm and its downsample branch's) in ONE
// launch: the first `blocks_a` workgroups run problem A.  fp32, 32-bit indices.
struct AffArgs {
  float* out;
  const float *a, *x, *mean, *rstd, *w, *q, *r, *add, *mask_src;
  int relu_self;
  unsigned total, C, HW;
  int nhwc;
  unsigned out_ld, add_ld;
  int a_splits;
  long long a_slab;
  int vec4;  // eligible for the quad-per-thread walk (alignment checked on the host)
};

__global__ __launch_bounds__(BLOCK) void k_chan_affine_pair(const AffArgs A, const AffArgs B, unsigned blocks_a) {
  const bool first = blockIdx.x < blocks_a;
  const AffArgs& p = first ? A : B;
  if (p.vec4) {
    chan_affine_v4_body(p.out, p.a, p.x, p.mean, p.rstd, p.w, p.q, p.r, p.add, p.mask_src, p.relu_self, p.total,
                        p.C, p.out_ld, p.add_ld, p.a_splits, p.a_slab, first ? blockIdx.x : blockIdx.x - blocks_a,
                        first ? blocks_a : gridDim.x - blocks_a);
    return;
  }
  chan_affine_body<float, unsigned>(p.out, p.a, p.x, p.mean, p.rstd, p.w, p.q, p.r, p.add, p.mask_src, p.relu_self,
                                    p.total, p.C, p.HW, p.nhwc, p.out_ld, p.add_ld, p.a_splits, p.a_slab,
                                    first ? blockIdx.x : blockIdx.x - blocks_a,
                                    first ? blocks_a : gridDim.x - blocks_a);
}

// One channel per GROUP of TPC threads (TPC = 64: one wave per channel, 4 channels
// per block, no LDS / barrier -- for the late layers where a channel has only
// N*HW <= 256 elements; TPC = 256: one block per channel).
template <typename T, typename I, int TPC>
__global__ __launch_bounds__(BLOCK) void k_chan_affine_bwd(
    T* __restrict__ gx, T* __restrict__ gw, T* __restrict__ gb, T* __restrict__ gres,
    const T* __restrict__ gy, const T* __restrict__ gy2, const T* __restrict__ x,
    const T* __restrict__ mean, const T* __restrict__ rstd, const T* __restrict__ w,
    const T* __restrict__ mask_src, I N, I C, I HW, int s1 = 1, long long l1 = 0, int s2 = 1,
    long long l2 = 0) {
  __shared__ double lds[2 * WAVES];
  constexpr int GROUPS = BLOCK / TPC;
  const I c = (I)blockIdx.x * GROUPS + threadIdx.x / TPC;
  const int lane = threadIdx.x % TPC;
  const bool live = c < C;
  double acc[2] = {0.0, 0.0};
  if (live) {
    const T rs = rstd ? rstd[c] : (T)1, mu = mean ? mean[c] : (T)0;
    const T s = (w ? w[c] : (T)1) * rs;
    const I per = N * HW;
    if (per <= (I)TPC) {
      // at most ONE element per lane (the 32-row maps of the last stage, whose cotangents arrive as ~32 slabs):
      // sixteen slabs in flight per pass -- one at a time is a dependent round trip per slab, ~5 us per launch
      if ((I)lane < per) {
        const I n = HW == 1 ? (I)lane : (I)lane / HW;
        const I idx = (n * C + c) * HW + ((I)lane - n * HW);
        T g = gy[idx], h = gy2 ? gy2[idx] : (T)0;
        const T m = mask_src ? mask_src[idx] : (T)1, xv = x ? x[idx] : (T)0;
        for (int sp = 1; sp < s1; sp += 16) {
          T v[16];
#pragma unroll
          for (int u = 0; u < 16; ++u) v[u] = gy[(long long)(sp + u < s1 ? sp + u : 0) * l1 + idx];
#pragma unroll
          for (int u = 0; u < 16; ++u) g += sp + u < s1 ? v[u] : (T)0;
        }
        if (gy2) {
          for (int sp = 1; sp < s2; sp += 16) {
            T v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = gy2[(long long)(sp + u < s2 ? sp + u : 0) * l2 + idx];
#pragma unroll
            for (int u = 0; u < 16; ++u) h += sp + u < s2 ? v[u] : (T)0;
          }
        }
        T gg = g;
        if (gy2) gg = gg + h;
        if (mask_src) gg = m > (T)0 ? gg : (T)0;
        if (gx) gx[idx] = gg * s;
        if (gres) gres[idx] = gg;
        if (x) acc[0] += (double)gg * (double)(T)((xv - mu) * rs);
        acc[1] += (double)gg;
      }
    } else {
    // ITER elements per thread with all loads issued before the first use (latency-bound)
    constexpr int ITER = 8;
    for (I e0 = lane; e0 < per; e0 += (I)TPC * ITER) {
      I idx[ITER];
      T g[ITER], h[ITER], xv[ITER], m[ITER];
#pragma unroll
      for (int t = 0; t < ITER; ++t) {
        const I e = e0 + (I)t * TPC;
        idx[t] = 0;
        if (e < per) {
          const I n = HW == 1 ? e : e / HW;
          idx[t] = (n * C + c) * HW + (e - n * HW);
          g[t] = gy[idx[t]];
          if (gy2) h[t] = gy2[idx[t]];
          if (mask_src) m[t] = mask_src[idx[t]];
          if (x) xv[t] = x[idx[t]];
        }
      }
      // split-K slabs, added in split order (batching eight slabs of every element per pass was measured: no
      // gain on these 32-row maps, 252 instead of 58 VGPRs)
#pragma unroll
      for (int t = 0; t < ITER; ++t) {
        if (e0 + (I)t * TPC < per) {
          for (int sp = 1; sp < s1; ++sp) g[t] += gy[(long long)sp * l1 + idx[t]];
          if (gy2)
            for (int sp = 1; sp < s2; ++sp) h[t] += gy2[(long long)sp * l2 + idx[t]];
        }
      }
#pragma unroll
      for (int t = 0; t < ITER; ++t) {
        if (e0 + (I)t * TPC < per) {
          T gg = g[t];
          if (gy2) gg = gg + h[t];  // the cotangents of the output's two consumers
          if (mask_src) gg = m[t] > (T)0 ? gg : (T)0;
          if (gx) gx[idx[t]] = gg * s;
          if (gres) gres[idx[t]] = gg;
          if (x) acc[0] += (double)gg * (double)(T)((xv[t] - mu) * rs);
          acc[1] += (double)gg;
        }
      }
    }
    }
  }
  if (TPC == 64) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      acc[0] += __shfl_down(acc[0], off, 64);
      acc[1] += __shfl_down(acc[1], off, 64);
    }
  } else {
    block_allreduce<2>(acc, lds);
  }
  if (live && lane == 0) {
    if (gw) gw[c] = (T)acc[0];
    if (gb) gb[c] = (T)acc[1];
  }
}

// first + slabs 1..n-1 of a W-wide column, eight loads in flight, added in split order
template <typename T, typename Col, int W>
__device__ __forceinline__ Col slab_sum(Col first, const T* p, int n, long long stride) {
  // batches of eight loads, ALL in flight before the first add; the last batch is predicated
  // instead of a one-by-one tail (a tail of dependent load-add pairs costs a round trip each)
  for (int sp = 1; sp < n; sp += 8) {
    Col v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int q = sp + u < n ? sp + u : 0;  // slab 0 is valid memory; its value is discarded
      v[u] = *reinterpret_cast<const Col*>(p + (long long)q * stride);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (sp + u < n) {
#pragma unroll
        for (int k = 0; k < W; ++k) first.e[k] += v[u].e[k];
      }
    }
  }
  return first;
}

// NHWC variant: element (row, c) at row*C + c, row = n*HW + hw.  A block owns W
// adjacent channels (W = 4: one 16-byte column) and spreads the rows over its 256
// threads, so the per-channel sums need no cross-block step (deterministic, no
// workspace) and C/W blocks are in flight; the tensors of this path are a few MB
// and L2-resident, the strided 16-byte reads cost less than a second launch would.
template <typename T, typename I, int W, int BS>
__global__ __launch_bounds__(BS) void k_chan_affine_bwd_nhwc(
    T* __restrict__ gx, T* __restrict__ gw, T* __restrict__ gb, T* __restrict__ gres,
    const T* __restrict__ gy, const T* __restrict__ gy2, const T* __restrict__ x,
    const T* __restrict__ mean, const T* __restrict__ rstd, const T* __restrict__ w,
    const T* __restrict__ mask_src, I rows, I C, int s1 = 1, long long l1 = 0, int s2 = 1,
    long long l2 = 0, int row_blocks = 1) {
  __shared__ double lds[2 * W * (BS / 64)];
  struct alignas(sizeof(T) * W) Col { T e[W]; };
  // row_blocks > 1: block (q, rb) owns channel column q and the rb-th share of the rows and
  // writes its per-channel partial sums to gw/gb + rb*C (hf_pack_ex adds the shares up)
  const I cq = (I)blockIdx.x % (C / W), rb = (I)blockIdx.x / (C / W);
  const I c0 = cq * W;
  const I rpb = (rows + (I)row_blocks - 1) / (I)row_blocks;
  const I row_lo = rb * rpb, row_hi = (row_lo + rpb < rows) ? row_lo + rpb : rows;
  T rs[W], mu[W], sc[W];
#pragma unroll
  for (int k = 0; k < W; ++k) {
    rs[k] = rstd ? rstd[c0 + k] : (T)1;
    mu[k] = mean ? mean[c0 + k] : (T)0;
    sc[k] = (w ? w[c0 + k] : (T)1) * rs[k];
  }
  double acc[2 * W];
#pragma unroll
  for (int k = 0; k < 2 * W; ++k) acc[k] = 0.0;
  // rows are visited ITER at a time with all loads issued before the first use: these
  // activation-sized kernels are latency-bound, one round trip per 8 rows instead of one each
  constexpr int ITER = 8;
  for (I r0 = row_lo + threadIdx.x; r0 < row_hi; r0 += (I)BS * ITER) {
    Col g[ITER], h[ITER], xv[ITER], m[ITER];
#pragma unroll
    for (int t = 0; t < ITER; ++t) {
      const I r = r0 + (I)t * BS;
      if (r < row_hi) {
        const I idx = r * C + c0;
        // every first load is issued before any slab is summed (a sum waits for its loads)
        g[t] = *reinterpret_cast<const Col*>(gy + idx);
        if (gy2) h[t] = *reinterpret_cast<const Col*>(gy2 + idx);
        if (x) xv[t] = *reinterpret_cast<const Col*>(x + idx);
        if (mask_src) m[t] = *reinterpret_cast<const Col*>(mask_src + idx);
      }
    }
    if (s1 > 1 || s2 > 1) {
#pragma unroll
      for (int t = 0; t < ITER; ++t) {
        const I r = r0 + (I)t * BS;
        if (r < row_hi) {
          const I idx = r * C + c0;
          if (s1 > 1) g[t] = slab_sum<T, Col, W>(g[t], gy + idx, s1, l1);  // split-K slabs, in split order
          if (gy2 && s2 > 1) h[t] = slab_sum<T, Col, W>(h[t], gy2 + idx, s2, l2);
        }
      }
    }
#pragma unroll
    for (int t = 0; t < ITER; ++t) {
      const I r = r0 + (I)t * BS;
      if (r < row_hi) {
        const I idx = r * C + c0;
        if (gy2) {
#pragma unroll
          for (int k = 0; k < W; ++k) g[t].e[k] = g[t].e[k] + h[t].e[k];
        }
        if (mask_src) {
#pragma unroll
          for (int k = 0; k < W; ++k) g[t].e[k] = m[t].e[k] > (T)0 ? g[t].e[k] : (T)0;
        }
        if (gres) *reinterpret_cast<Col*>(gres + idx) = g[t];
        if (gx) {
          Col o;
#pragma unroll
          for (int k = 0; k < W; ++k) o.e[k] = g[t].e[k] * sc[k];
          *reinterpret_cast<Col*>(gx + idx) = o;
        }
#pragma unroll
        for (int k = 0; k < W; ++k) {
          if (x) acc[2 * k] += (double)g[t].e[k] * (double)(T)((xv[t].e[k] - mu[k]) * rs[k]);
          acc[2 * k + 1] += (double)g[t].e[k];
        }
      }
    }
  }
  block_allreduce<2 * W, BS / 64>(acc, lds);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < W; ++k) {
      if (gw) gw[rb * C + c0 + k] = (T)acc[2 * k];
      if (gb) gb[rb * C + c0 + k] = (T)acc[2 * k + 1];
    }
  }
}

// BatchNorm adjoint, NHWC fp32, ROW-MAJOR thread map: thread (ty, tx) owns the 16-byte channel
// column tx of rows ty, ty + RP, ... of its block's row share, so that a wave reads whole
// contiguous rows (the column-per-block kernel above reads 16 bytes every C*4 bytes: one cache
// line per lane).  Per-channel sums: per thread over its rows, then over ty through LDS in a
// fixed order; every block writes its partial sums to gw / gb + blockIdx.x*C (hf_pack_ex adds
// the row shares up).  Cotangents may arrive as split-K slabs.
__device__ __forceinline__ void bn_adjoint_rows_body(
    float* __restrict__ gx, float* __restrict__ gw, float* __restrict__ gb, float* __restrict__ gres,
    const float* __restrict__ gy, int s1, long long l1, const float* __restrict__ gy2, int s2, long long l2,
    const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
    const float* __restrict__ w, const float* __restrict__ mask_src, unsigned rows, unsigned C,
    unsigned rows_per_block, unsigned bid, double* red, const bool publish = false) {
  struct alignas(16) Col { float e[4]; };
  const unsigned quads = C / 4, RP = BLOCK / quads;
  const unsigned tx = threadIdx.x % quads, ty = threadIdx.x / quads;
  const unsigned c0 = tx * 4;
  const bool live = ty < RP;
  float rs[4], mu[4], sc[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    rs[k] = rstd ? rstd[c0 + k] : 1.f;
    mu[k] = mean ? mean[c0 + k] : 0.f;
    sc[k] = (w ? w[c0 + k] : 1.f) * rs[k];
  }
  double acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = 0.0;
  const unsigned row_lo = bid * rows_per_block;
  const unsigned row_hi = row_lo + rows_per_block < rows ? row_lo + rows_per_block : rows;
  if (live) {
    // two rows per pass: their first loads and their slab batches are all in flight together
    for (unsigned r = row_lo + ty; r < row_hi; r += 2 * RP) {
      const bool two = r + RP < row_hi;
      const unsigned idx0 = r * C + c0, idx1 = (two ? r + RP : r) * C + c0;
      Col g0 = *reinterpret_cast<const Col*>(gy + idx0), g1 = *reinterpret_cast<const Col*>(gy + idx1);
      Col h0, h1, x0, x1, m0, m1;
      if (gy2) { h0 = *reinterpret_cast<const Col*>(gy2 + idx0); h1 = *reinterpret_cast<const Col*>(gy2 + idx1); }
      if (x) { x0 = *reinterpret_cast<const Col*>(x + idx0); x1 = *reinterpret_cast<const Col*>(x + idx1); }
      if (mask_src) {
        m0 = *reinterpret_cast<const Col*>(mask_src + idx0);
        m1 = *reinterpret_cast<const Col*>(mask_src + idx1);
      }
      // split-K slabs of both rows and both cotangents: one loop, 8 slabs x up to 4 columns in flight per
      // pass (each column still adds its slabs in split order: bitwise the one-column-at-a-time sums, which
      // cost a round trip per column and batch)
      // (sixteen slabs x two columns per pass was measured slower: 272 VGPRs; the first batch as straight-line code
      // behind the row loads with its first addition pinned behind its last load -- 43 loads before the first wait, 256
      // VGPRs -- measured +0.5 % on ResNet-18, -1.5 % on All-CNN-C's large maps: profiles/r04_rows_straight_rejected.jsonl)
      const int smax = (gy2 && s2 > s1) ? s2 : s1;
      for (int sp = 1; sp < smax; sp += 8) {
        Col vg0[8], vg1[8], vh0[8], vh1[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const long long o1 = (long long)(sp + u < s1 ? sp + u : 0) * l1;
          vg0[u] = *reinterpret_cast<const Col*>(gy + o1 + idx0);
          vg1[u] = *reinterpret_cast<const Col*>(gy + o1 + idx1);
        }
        if (gy2 && s2 > 1) {
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const long long o2 = (long long)(sp + u < s2 ? sp + u : 0) * l2;
            vh0[u] = *reinterpret_cast<const Col*>(gy2 + o2 + idx0);
            vh1[u] = *reinterpret_cast<const Col*>(gy2 + o2 + idx1);
          }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (sp + u < s1) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { g0.e[k] += vg0[u].e[k]; g1.e[k] += vg1[u].e[k]; }
          }
        }
        if (gy2 && s2 > 1) {
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            if (sp + u < s2) {
#pragma unroll
              for (int k = 0; k < 4; ++k) { h0.e[k] += vh0[u].e[k]; h1.e[k] += vh1[u].e[k]; }
            }
          }
        }
      }
      Col o0, o1;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float a0 = gy2 ? g0.e[k] + h0.e[k] : g0.e[k], a1 = gy2 ? g1.e[k] + h1.e[k] : g1.e[k];
        if (mask_src) { a0 = m0.e[k] > 0.f ? a0 : 0.f; a1 = m1.e[k] > 0.f ? a1 : 0.f; }
        g0.e[k] = a0; g1.e[k] = a1;
        o0.e[k] = a0 * sc[k]; o1.e[k] = a1 * sc[k];
        if (x) acc[2 * k] += (double)a0 * (double)(float)((x0.e[k] - mu[k]) * rs[k]);
        acc[2 * k + 1] += (double)a0;
        if (two) {
          if (x) acc[2 * k] += (double)a1 * (double)(float)((x1.e[k] - mu[k]) * rs[k]);
          acc[2 * k + 1] += (double)a1;
        }
      }
      if (gres) { *reinterpret_cast<Col*>(gres + idx0) = g0; if (two) *reinterpret_cast<Col*>(gres + idx1) = g1; }
      if (gx) { *reinterpret_cast<Col*>(gx + idx0) = o0; if (two) *reinterpret_cast<Col*>(gx + idx1) = o1; }
    }
  }
  // cross-row sums: red[k][ty][tx] (consecutive lanes -> consecutive words: no bank conflicts), then
  // 8*quads threads each add one (k, tx) column over ty in a fixed order.  (The first version let the
  // `quads` threads of ty == 0 walk all 8 sums serially: 8*RP dependent LDS reads per thread behind
  // 8-way bank conflicts -- 72 % of this kernel's LDS cycles were conflict cycles,
  // profiles/r03_engine_kernel_counters.json.)  Same summation order, bitwise the same sums.
  if (live) {
#pragma unroll
    for (int k = 0; k < 8; ++k) red[(k * RP + ty) * quads + tx] = acc[k];
  }
  __syncthreads();
  for (unsigned idx = threadIdx.x; idx < 8 * quads; idx += BLOCK) {
    const unsigned k = idx / quads, col = idx - k * quads;
    double sum = 0.0;
    for (unsigned t = 0; t < RP; ++t) sum += red[(k * RP + t) * quads + col];  // fixed order over ty
    float* dst = (k & 1) ? gb : gw;
    if (dst) {
      // publish: write-through (sc1) store -- visible device-wide once drained, no release fence (in-launch readers)
      if (publish) __hip_atomic_store(dst + bid * C + col * 4 + (k >> 1), (float)sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else dst[bid * C + col * 4 + (k >> 1)] = (float)sum;
    }
  }
}

__global__ __launch_bounds__(BLOCK) void k_bn_adjoint_rows(
    float* __restrict__ gx, float* __restrict__ gw, float* __restrict__ gb, float* __restrict__ gres,
    const float* __restrict__ gy, int s1, long long l1, const float* __restrict__ gy2, int s2, long long l2,
    const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
    const float* __restrict__ w, const float* __restrict__ mask_src, unsigned rows, unsigned C,
    unsigned rows_per_block) {
  __shared__ double red[BLOCK * 8];
  bn_adjoint_rows_body(gx, gw, gb, gres, gy, s1, l1, gy2, s2, l2, x, mean, rstd, w, mask_src, rows, C,
                       rows_per_block, blockIdx.x, red);
}

// ---- train-mode BatchNorm: the per-channel finalisation inside the reduction's launch -----------------------
// Every block publishes its partial row, draws a ticket; the LAST arriver re-reads all rows (agent-scope loads,
// fixed order: the same sums whichever block comes last) and writes the per-channel result.  The ticket word
// resets itself.  Replaces one tiny dependent launch (hf_bn_train_coeffs / hf_bn_batch_stats) per layer and sweep.
__device__ __forceinline__ bool last_block_arrives(unsigned* ticket, unsigned* s_last) {
  // (cdna_hip_programming.md, in-launch reduction, write-through form -- as hf_conv.hip's split-K tickets: the
  // partial sums were stored write-through (sc1); every wave drains them, ONE lane draws the ticket, the last
  // arriver acquires once (drops stale lines) and then reads the rows with plain loads.  A release FENCE here
  // instead would write this workgroup's share of the activation-sized outputs back out of L2 first.)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned old = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned last = old == gridDim.x - 1u ? 1u : 0u;
    if (last) {
      __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // next launch
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    *s_last = last;
  }
  __syncthreads();
  return *s_last != 0u;
}
__device__ __forceinline__ float ld_agent(const float* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_agent(const double* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct TrainFinal {
  unsigned* ticket;
  float *q_out, *r_out;
  const float *fw, *vq, *vr;  // the layer's scale (nullable: 1) and the parameter tangents (nullable)
  float inv_m;
};

// Column sums of `nrows` partial rows [nrows, C] (C % 4 == 0, C / 4 <= BLOCK) by ONE workgroup, all threads busy:
// thread (quad tx, group ty) adds rows ty, ty + G, ... (16-byte loads, all in flight), the groups are combined through
// LDS in a fixed order.  out: LDS [C] doubles.  scratch: LDS 4 * BLOCK doubles.
__device__ __forceinline__ void final_column_sums(const float* rows, unsigned nrows, unsigned C, double* scratch,
                                                  double* out) {
  const unsigned quads = C / 4, G = BLOCK / quads;
  const unsigned tx = threadIdx.x % quads, ty = threadIdx.x / quads;
  if (ty < G) {
    double a[4] = {0.0, 0.0, 0.0, 0.0};
    for (unsigned p = ty; p < nrows; p += G) {
      const F4 v = ld4(rows + (size_t)p * C + 4 * tx);
#pragma unroll
      for (int k = 0; k < 4; ++k) a[k] += (double)v.e[k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) scratch[(k * G + ty) * quads + tx] = a[k];
  }
  __syncthreads();
  for (unsigned idx = threadIdx.x; idx < 4 * quads; idx += BLOCK) {
    const unsigned k = idx / quads, col = idx - k * quads;
    double sum = 0.0;
    for (unsigned t = 0; t < G; ++t) sum += scratch[(k * G + t) * quads + col];
    out[col * 4 + k] = sum;
  }
  __syncthreads();
}

// k_bn_adjoint_rows + hf_bn_train_coeffs in one launch (q = vq - w*rstd*S_x/m, r = vr - w*rstd*S_1/m).
__global__ __launch_bounds__(BLOCK) void k_bn_adjoint_rows_train(
    float* __restrict__ gx, float* gw, float* gb, float* __restrict__ gres,
    const float* __restrict__ gy, int s1, long long l1, const float* __restrict__ gy2, int s2, long long l2,
    const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
    const float* __restrict__ w, const float* __restrict__ mask_src, unsigned rows, unsigned C,
    unsigned rows_per_block, const TrainFinal f) {
  __shared__ double red[BLOCK * 8];
  __shared__ double fin[2 * 4 * BLOCK];  // the two finished column sums, C <= 4 * BLOCK channels each
  __shared__ unsigned s_last;
  bn_adjoint_rows_body(gx, gw, gb, gres, gy, s1, l1, gy2, s2, l2, x, mean, rstd, w, mask_src, rows, C,
                       rows_per_block, blockIdx.x, red, true);
  if (!last_block_arrives(f.ticket, &s_last)) return;
  final_column_sums(gw, gridDim.x, C, red, fin);
  final_column_sums(gb, gridDim.x, C, red, fin + 4 * BLOCK);
  for (unsigned c = threadIdx.x; c < C; c += BLOCK) {
    const float k = (f.fw ? f.fw[c] : 1.f) * rstd[c] * f.inv_m;
    f.q_out[c] = (f.vq ? f.vq[c] : 0.f) - k * (float)fin[c];
    f.r_out[c] = (f.vr ? f.vr[c] : 0.f) - k * (float)fin[4 * BLOCK + c];
  }
}

// ---- train-mode BatchNorm: reduction, per-channel finalisation AND the elementwise pass in ONE launch ---------
// A grid-wide barrier between the two passes (every workgroup is resident: the host refuses more row blocks than
// the device has compute units).  `bar`: one zero-initialised 64-bit counter per layer, never reset -- the k-th
// launch waits for k * gridDim.x arrivals (2^64 never wraps; a 32-bit word would after ~9 h of products).
// Arrivals: this workgroup's partial rows were stored write-through and drained (as last_block_arrives); the rows
// of the other workgroups are then read with agent-scope loads -- no acquire fence, which would drop every clean
// line of this XCD's L2 forty times per product.
__device__ __forceinline__ void grid_barrier(unsigned long long* bar) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long old = __hip_atomic_fetch_add(bar, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long target = (old / gridDim.x + 1ull) * gridDim.x;
    // (bounded: ~2^26 polls of >= 64 clocks are seconds -- a launch whose workgroups cannot all become resident ends
    // with wrong sums instead of hanging the device; the host refuses such launches up front)
    for (unsigned spins = 0; spins < (1u << 26) &&
                             __hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target; ++spins)
      __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
}

// final_column_sums with agent-scope loads (the partial rows come from other workgroups of THIS launch).
__device__ __forceinline__ void final_column_sums_agent(const float* rows, unsigned nrows, unsigned C,
                                                        double* scratch, double* out) {
  const unsigned quads = C / 4, G = BLOCK / quads;
  const unsigned tx = threadIdx.x % quads, ty = threadIdx.x / quads;
  if (ty < G) {
    double a[4] = {0.0, 0.0, 0.0, 0.0};
    for (unsigned p0 = ty; p0 < nrows; p0 += 4 * G) {  // four partial rows (16 scalar loads) in flight
      float v[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const unsigned p = p0 + u * G < nrows ? p0 + u * G : p0;
#pragma unroll
        for (int k = 0; k < 4; ++k) v[u][k] = ld_agent(rows + (size_t)p * C + 4 * tx + k);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (p0 + u * G < nrows) {
#pragma unroll
          for (int k = 0; k < 4; ++k) a[k] += (double)v[u][k];
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) scratch[(k * G + ty) * quads + tx] = a[k];
  }
  __syncthreads();
  for (unsigned idx = threadIdx.x; idx < 4 * quads; idx += BLOCK) {
    const unsigned k = idx / quads, col = idx - k * quads;
    double sum = 0.0;
    for (unsigned t = 0; t < G; ++t) sum += scratch[(k * G + t) * quads + col];
    out[col * 4 + k] = sum;
  }
  __syncthreads();
}

struct TrainApply {
  unsigned long long* bar;
  float* out;        // [rows, out_ld] (out_ld == 0: dense)
  const float* add;  // nullable, [rows, add_ld]
  const float* out_mask;  // nullable: out = out_mask > 0 ? t : 0
  unsigned out_ld, add_ld;
  float *q_out, *r_out;  // nullable: the per-channel vectors, written by workgroup 0 (tests)
  const float *fw, *vq, *vr;
  float inv_m;
};

// pass 1 = k_bn_adjoint_rows (g = mask * sum of slabs -> gres, partial sums of g and xhat*g per workgroup);
// barrier; every workgroup adds the partial rows up (fixed order: the same sums in every workgroup) and forms
//   q = vq - fw*rstd*S_x/m,  r = vr - fw*rstd*S_1/m;
// pass 2 = hf_chan_affine_ex on the workgroup's OWN rows, each thread re-reading the g it wrote itself:
//   out = mask(g*(fw*rstd) + xhat*q + r + add).
__global__ __launch_bounds__(BLOCK) void k_bn_rows_train_apply(
    float* gw, float* gb, float* gres, const float* __restrict__ gy, int s1, long long l1,
    const float* __restrict__ gy2, int s2, long long l2, const float* __restrict__ x,
    const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ mask_src,
    unsigned rows, unsigned C, unsigned rows_per_block, const TrainApply f) {
  __shared__ double red[BLOCK * 8];
  __shared__ double fin[2 * 4 * BLOCK];
  __shared__ float qs[4 * BLOCK], rsh[4 * BLOCK];
  bn_adjoint_rows_body(nullptr, gw, gb, gres, gy, s1, l1, gy2, s2, l2, x, mean, rstd, nullptr, mask_src, rows, C,
                       rows_per_block, blockIdx.x, red, true);
  grid_barrier(f.bar);
  final_column_sums_agent(gw, gridDim.x, C, red, fin);
  final_column_sums_agent(gb, gridDim.x, C, red, fin + 4 * BLOCK);
  for (unsigned c = threadIdx.x; c < C; c += BLOCK) {
    const float k = (f.fw ? f.fw[c] : 1.f) * rstd[c] * f.inv_m;
    const float q = (f.vq ? f.vq[c] : 0.f) - k * (float)fin[c];
    const float r = (f.vr ? f.vr[c] : 0.f) - k * (float)fin[4 * BLOCK + c];
    qs[c] = q;
    rsh[c] = r;
    if (blockIdx.x == 0 && f.q_out) { f.q_out[c] = q; f.r_out[c] = r; }
  }
  __syncthreads();
  const unsigned quads = C / 4, RP = BLOCK / quads;
  const unsigned tx = threadIdx.x % quads, ty = threadIdx.x / quads;
  if (ty >= RP) return;
  const unsigned c0 = tx * 4;
  const unsigned row_lo = blockIdx.x * rows_per_block;
  const unsigned row_hi = row_lo + rows_per_block < rows ? row_lo + rows_per_block : rows;
  float rs[4], mu[4], sc[4], q4[4], r4[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    rs[k] = rstd[c0 + k];
    mu[k] = mean[c0 + k];
    sc[k] = (f.fw ? f.fw[c0 + k] : 1.f) * rs[k];
    q4[k] = qs[c0 + k];
    r4[k] = rsh[c0 + k];
  }
  for (unsigned r = row_lo + ty; r < row_hi; r += RP) {
    const unsigned idx = r * C + c0;
    const F4 g = ld4(gres + idx), xv = ld4(x + idx);
    F4 addv, mv, o;
    if (f.add) addv = ld4(f.add + (f.add_ld ? r * f.add_ld + c0 : idx));
    if (f.out_mask) mv = ld4(f.out_mask + idx);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float acc = g.e[k] * sc[k];
      acc += ((xv.e[k] - mu[k]) * rs[k]) * q4[k];
      acc += r4[k];
      if (f.add) acc += addv.e[k];
      if (f.out_mask) acc = mv.e[k] > 0.f ? acc : 0.f;
      o.e[k] = acc;
    }
    *reinterpret_cast<F4*>(f.out + (f.out_ld ? r * f.out_ld + c0 : idx)) = o;
  }
}

// Both column sums in one pass (all loads of both partial-row sets in flight together).  scratch: 8 * BLOCK doubles.
// `between()` runs right after the first batch of partial-row loads is issued: the caller's own independent loads go
// there, so that one round trip covers both.
template <typename Between>
__device__ __forceinline__ void final_column_sums2(const float* __restrict__ rows_a, const float* __restrict__ rows_b,
                                                   unsigned nrows, unsigned C, double* scratch, double* out_a,
                                                   double* out_b, Between&& between) {
  const unsigned quads = C / 4, G = BLOCK / quads;
  const unsigned tx = threadIdx.x % quads, ty = threadIdx.x / quads;
  const bool live = ty < G;
  double a[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  F4 va[4], vb[4];
  auto issue = [&](unsigned p0) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const unsigned p = p0 + u * G < nrows ? p0 + u * G : 0u;
      va[u] = ld4(rows_a + (size_t)p * C + 4 * tx);
      vb[u] = ld4(rows_b + (size_t)p * C + 4 * tx);
    }
  };
  auto add = [&](unsigned p0) {  // four partial rows of each set, added in row order
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (p0 + u * G < nrows) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { a[k] += (double)va[u].e[k]; a[4 + k] += (double)vb[u].e[k]; }
      }
  };
  const bool first = live && ty < nrows;
  issue(ty);  // (unconditional, out-of-range rows read row 0: a branch here makes the compiler shuffle -- and wait for --
              // the loaded registers at its join)
  between();
  if (first) {
    // (pins the first use of the rows BEHIND the caller's loads: without it the compiler adds them up -- and waits
    // for them -- before it issues those)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      asm volatile("" : "+v"(va[u].e[0]), "+v"(va[u].e[1]), "+v"(va[u].e[2]), "+v"(va[u].e[3]) : : "memory");
      asm volatile("" : "+v"(vb[u].e[0]), "+v"(vb[u].e[1]), "+v"(vb[u].e[2]), "+v"(vb[u].e[3]) : : "memory");
    }
    add(ty);
  }
  if (live) {
    for (unsigned p0 = ty + 4 * G; p0 < nrows; p0 += 4 * G) { issue(p0); add(p0); }
#pragma unroll
    for (int k = 0; k < 8; ++k) scratch[(k * G + ty) * quads + tx] = a[k];
  }
  __syncthreads();
  for (unsigned idx = threadIdx.x; idx < 8 * quads; idx += BLOCK) {
    const unsigned k = idx / quads, col = idx - k * quads;
    double sum = 0.0;
    for (unsigned t = 0; t < G; ++t) sum += scratch[(k * G + t) * quads + col];
    (k < 4 ? out_a : out_b)[col * 4 + (k & 3)] = sum;
  }
  __syncthreads();
}

// ---- train-mode BatchNorm: the per-channel finalisation in the CONSUMER's prologue ---------------------------
// k_chan_affine_v4 whose workgroups first add the reduction launch's partial rows up themselves (plain loads: the
// rows come from the PREVIOUS launch; every workgroup the same fixed order, so the same q / r everywhere) --
//   q = vq - w*rstd*S_x/m,  r = vr - w*rstd*S_1/m   (hf_bn_train_coeffs),  then  out = mask(a*(w*rstd) + xhat*q + r + add).
// The redundant sums cost each workgroup one more round trip (nparts * C * 8 bytes out of L2); the finalisation as the
// reduction launch's TAIL (k_bn_adjoint_rows_train) costs a ticket, a drain and a one-workgroup re-read, as its own
// launch (k_bn_train_coeffs) a launch boundary more.
struct AffTrainArgs {
  float* out;
  const float *a, *x, *mean, *rstd, *w, *part_x, *part_1;
  unsigned nparts;
  const float *vq, *vr;
  float inv_m;
  const float *add, *mask_src;
  unsigned total, C, out_ld, add_ld;
  int a_splits;
  long long a_slab;
};

template <bool SLABS, bool ADD, bool MASK>
__device__ __forceinline__ void affine_train_body(const AffTrainArgs& p, unsigned bid, unsigned nblocks,
                                                  double* scratch, double* fin, float* qs, float* rsh) {
  const float* __restrict__ a = p.a;
  const float* __restrict__ x = p.x;
  const float* __restrict__ mean = p.mean;
  const float* __restrict__ rstd = p.rstd;
  const float* __restrict__ w = p.w;
  const float* __restrict__ add = p.add;
  const float* __restrict__ mask_src = p.mask_src;
  float* __restrict__ out = p.out;
  const unsigned total = p.total, C = p.C, out_ld = p.out_ld, add_ld = p.add_ld;
  const int a_splits = p.a_splits;
  const long long a_slab = p.a_slab;
  // this thread's element quad: every load of it issued right behind the first partial-row loads and BEFORE those are
  // added up (none depends on the sums): one round trip for both
  const unsigned quads_total = total >> 2;
  const unsigned v = bid * BLOCK + threadIdx.x;
  const bool have = v < quads_total;
  const unsigned i = v << 2;
  const unsigned row = i / C, c = i - row * C;
  F4 rs4, w4, mu4, xv, addv, mv, av, t[16];
  final_column_sums2(p.part_x, p.part_1, p.nparts, C, scratch, fin, fin + 4 * BLOCK, [&]() {
    // (no run-time branches around these loads -- optional operands are template flags, threads past the end read
    // element 0: at a branch's join the compiler copies the loaded registers, which waits for them right here)
    const unsigned ii = have ? i : 0u, cc = have ? c : 0u, rr = have ? row : 0u;
    rs4 = ld4(rstd + cc);
    mu4 = ld4(mean + cc);
    xv = ld4(x + ii);
    w4 = ld4(w + cc);
    if (ADD) addv = ld4(add + (add_ld ? rr * add_ld + cc : ii));
    if (MASK) mv = ld4(mask_src + ii);
    av = ld4(a + ii);
    if (SLABS) {
#pragma unroll
      for (int u = 0; u < 16; ++u) t[u] = ld4(a + (long long)(1 + u < a_splits ? 1 + u : 0) * a_slab + ii);
    }
  });
  for (unsigned ch = threadIdx.x; ch < C; ch += BLOCK) {
    const float k = w[ch] * rstd[ch] * p.inv_m;
    qs[ch] = (p.vq ? p.vq[ch] : 0.f) - k * (float)fin[ch];
    rsh[ch] = (p.vr ? p.vr[ch] : 0.f) - k * (float)fin[4 * BLOCK + ch];
  }
  __syncthreads();
  if (have) {
    if (SLABS) {  // (slabs in split order, as chan_affine_v4_body)
#pragma unroll
      for (int u = 0; u < 16; ++u)
#pragma unroll
        for (int k = 0; k < 4; ++k) av.e[k] += 1 + u < a_splits ? t[u].e[k] : 0.f;
      for (int sp = 17; sp < a_splits; sp += 16) {
#pragma unroll
        for (int u = 0; u < 16; ++u) t[u] = ld4(a + (long long)(sp + u < a_splits ? sp + u : 0) * a_slab + i);
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
          for (int k = 0; k < 4; ++k) av.e[k] += sp + u < a_splits ? t[u].e[k] : 0.f;
      }
    }
    F4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float rs = rs4.e[k];
      float acc = av.e[k] * (w4.e[k] * rs);
      acc += ((xv.e[k] - mu4.e[k]) * rs) * qs[c + k];
      acc += rsh[c + k];
      if (ADD) acc += addv.e[k];
      if (MASK) acc = mv.e[k] > 0.f ? acc : 0.f;
      o.e[k] = acc;
    }
    *reinterpret_cast<F4*>(out + (out_ld ? row * out_ld + c : i)) = o;
  }
  // (a grid capped below one quad per thread: the rest by the plain walk)
  if (nblocks * BLOCK < quads_total)
    chan_affine_v4_body(out, a, x, mean, rstd, w, qs, rsh, add, mask_src, 0, total, C, out_ld, add_ld, a_splits,
                        a_slab, bid + nblocks, nblocks);
}

template <bool SLABS, bool ADD, bool MASK>
__global__ __launch_bounds__(BLOCK) void k_chan_affine_v4_train(const AffTrainArgs p) {
  __shared__ double scratch[8 * BLOCK];
  __shared__ double fin[2 * 4 * BLOCK];
  __shared__ float qs[4 * BLOCK], rsh[4 * BLOCK];
  affine_train_body<SLABS, ADD, MASK>(p, blockIdx.x, gridDim.x, scratch, fin, qs, rsh);
}

// Two independent train-mode layers (a residual block's first BatchNorm and its downsample branch's) in ONE launch:
// the first `blocks_a` workgroups run problem A.  No residual operand in either (template flags: slabs / mask of A, B).
template <bool SA, bool MA, bool SB, bool MB>
__global__ __launch_bounds__(BLOCK) void k_chan_affine_v4_train_pair(const AffTrainArgs A, const AffTrainArgs B,
                                                                     unsigned blocks_a) {
  __shared__ double scratch[8 * BLOCK];
  __shared__ double fin[2 * 4 * BLOCK];
  __shared__ float qs[4 * BLOCK], rsh[4 * BLOCK];
  if (blockIdx.x < blocks_a) affine_train_body<SA, false, MA>(A, blockIdx.x, blocks_a, scratch, fin, qs, rsh);
  else affine_train_body<SB, false, MB>(B, blockIdx.x - blocks_a, gridDim.x - blocks_a, scratch, fin, qs, rsh);
}

// One-pass batch statistics of a train-mode BatchNorm's forward: sums the convolution's split-K slabs into
// a_out (row-major walk as k_bn_adjoint_rows), per-channel sum a and sum a^2 in fp64 per thread / block /
// (last block) over the blocks; then mean, biased variance = E[a^2] - mean^2 (fp64: 1e-16 * mean^2/var relative,
// far below fp32 for any layer a network can train), rstd, and -- momentum >= 0 -- the running statistics as
// torch.nn.BatchNorm2d's forward moves them.  part: [gridDim.x, 2, C] doubles.
__global__ __launch_bounds__(BLOCK) void k_bn_stats_rows(
    float* __restrict__ a_out, const float* __restrict__ a, int splits, long long slab, double* part,
    unsigned* ticket, float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ run_mean,
    float* __restrict__ run_var, double count, float eps, float momentum, unsigned rows, unsigned C,
    unsigned rows_per_block) {
  struct alignas(16) Col { float e[4]; };
  __shared__ double red[BLOCK * 8];
  __shared__ unsigned s_last;
  const unsigned quads = C / 4, RP = BLOCK / quads;
  const unsigned tx = threadIdx.x % quads, ty = threadIdx.x / quads;
  const unsigned c0 = tx * 4;
  const bool live = ty < RP;
  double acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = 0.0;
  const unsigned row_lo = blockIdx.x * rows_per_block;
  const unsigned row_hi = row_lo + rows_per_block < rows ? row_lo + rows_per_block : rows;
  if (live) {
    for (unsigned r = row_lo + ty; r < row_hi; r += RP) {
      const unsigned idx = r * C + c0;
      Col v = *reinterpret_cast<const Col*>(a + idx);
      for (int sp = 1; sp < splits; sp += 8) {  // eight slabs in flight, added in split order
        Col t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
          t[u] = *reinterpret_cast<const Col*>(a + (long long)(sp + u < splits ? sp + u : 0) * slab + idx);
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int k = 0; k < 4; ++k) v.e[k] += sp + u < splits ? t[u].e[k] : 0.f;
      }
      if (a_out) *reinterpret_cast<Col*>(a_out + idx) = v;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        acc[2 * k] += (double)v.e[k];
        acc[2 * k + 1] += (double)v.e[k] * (double)v.e[k];
      }
    }
  }
  if (live) {
#pragma unroll
    for (int k = 0; k < 8; ++k) red[(k * RP + ty) * quads + tx] = acc[k];
  }
  __syncthreads();
  for (unsigned idx = threadIdx.x; idx < 8 * quads; idx += BLOCK) {
    const unsigned k = idx / quads, col = idx - k * quads;
    double sum = 0.0;
    for (unsigned t = 0; t < RP; ++t) sum += red[(k * RP + t) * quads + col];  // fixed order over ty
    __hip_atomic_store(part + ((size_t)blockIdx.x * 2 + (k & 1)) * C + col * 4 + (k >> 1), sum, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);  // (write-through: see last_block_arrives)
  }
  if (!ticket) return;  // (partial sums only: hf_bn_forward_train adds them up in its prologue)
  if (!last_block_arrives(ticket, &s_last)) return;
  for (unsigned c = threadIdx.x; c < C; c += BLOCK) {
    double s = 0.0, sq = 0.0;
    for (unsigned p0 = 0; p0 < gridDim.x; p0 += 8) {  // eight rows in flight, added in row order
      double ts[8], tq[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const unsigned p = p0 + u < gridDim.x ? p0 + u : p0;
        ts[u] = part[((size_t)p * 2) * C + c];
        tq[u] = part[((size_t)p * 2 + 1) * C + c];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (p0 + u < gridDim.x) { s += ts[u]; sq += tq[u]; }
      }
    }
    const double m = s / count;
    double var = sq / count - m * m;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)m;
    rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (momentum >= 0.f && run_mean && run_var) {
      const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
      run_mean[c] = (float)((1.0 - (double)momentum) * (double)run_mean[c] + (double)momentum * (double)(float)m);
      run_var[c] = (float)((1.0 - (double)momentum) * (double)run_var[c] + (double)momentum * unbiased);
    }
  }
}

// Two independent layers' adjoints in ONE launch (see k_chan_affine_pair).
struct BnAdjArgs {
  float *gx, *gw, *gb, *gres;
  const float* gy;
  int s1;
  long long l1;
  const float* gy2;
  int s2;
  long long l2;
  const float *x, *mean, *rstd, *w, *mask_src;
  unsigned rows, C, rows_per_block;
};

__global__ __launch_bounds__(BLOCK) void k_bn_adjoint_rows_pair(const BnAdjArgs A, const BnAdjArgs B,
                                                                unsigned blocks_a) {
  __shared__ double red[BLOCK * 8];
  const bool first = blockIdx.x < blocks_a;
  const BnAdjArgs& p = first ? A : B;
  bn_adjoint_rows_body(p.gx, p.gw, p.gb, p.gres, p.gy, p.s1, p.l1, p.gy2, p.s2, p.l2, p.x, p.mean, p.rstd, p.w,
                       p.mask_src, p.rows, p.C, p.rows_per_block, first ? blockIdx.x : blockIdx.x - blocks_a, red);
}

// Adjoint pre-pass of a fused eval-BatchNorm(+add+ReLU) layer in NHWC [rows, C], elementwise:
//   g  = (sum_s gyA[s] + sum_s gyB[s]) * [mask_src > 0]      (the two consumers' cotangents,
//                                                              each possibly split-K slabs)
//   g_out = g (the residual branch's cotangent, and what the per-channel sums are taken of)
//   ga_out = g * w[c]*rstd[c]                                 (cotangent of the convolution output)
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_bn_adjoint_pre(
    T* __restrict__ g_out, T* __restrict__ ga_out, const T* __restrict__ gyA, int a_splits,
    long long a_slab, const T* __restrict__ gyB, int b_splits, long long b_slab,
    const T* __restrict__ mask_src, const T* __restrict__ w, const T* __restrict__ rstd,
    unsigned total, unsigned C) {
  for (unsigned i = blockIdx.x * BLOCK + threadIdx.x; i < total; i += gridDim.x * BLOCK) {
    T g = gyA[i];
    for (int sp = 1; sp < a_splits; ++sp) g += gyA[(long long)sp * a_slab + i];
    if (gyB) {
      T h = gyB[i];
      for (int sp = 1; sp < b_splits; ++sp) h += gyB[(long long)sp * b_slab + i];
      g = g + h;
    }
    if (mask_src) g = mask_src[i] > (T)0 ? g : (T)0;
    if (g_out) g_out[i] = g;
    if (ga_out) {
      const unsigned c = i % C;
      ga_out[i] = g * ((w ? w[c] : (T)1) * rstd[c]);
    }
  }
}

// Forward of conv -> (eval-BatchNorm | bias) (+ residual) (+ ReLU) from the convolution's split-K slabs,
// NHWC [rows, C]; one element per thread (activation-sized, latency-bound).  The rounding sequence
// is chan_affine_body's forward: ((s - mean)*rstd)*w, + b, + res.
__global__ __launch_bounds__(BLOCK) void k_bn_forward(
    float* __restrict__ y, float* __restrict__ y2, unsigned y2_ld, float* a_out,
    const float* a, int splits, long long slab, const float* __restrict__ mean,  // (a_out may alias a: in place)
    const float* __restrict__ rstd, const float* __restrict__ w, const float* __restrict__ b,
    const float* __restrict__ res, unsigned res_ld, int relu, unsigned total, unsigned C) {
  for (unsigned i = blockIdx.x * BLOCK + threadIdx.x; i < total; i += gridDim.x * BLOCK) {
  const unsigned c = i % C, row = i / C;
  float av = a[i];
  for (int sp = 1; sp < splits; sp += 8) {  // eight slabs in flight, summed in split order
    float t8[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t8[u] = a[(long long)(sp + u < splits ? sp + u : 0) * slab + i];
#pragma unroll
    for (int u = 0; u < 8; ++u) av += sp + u < splits ? t8[u] : 0.f;
  }
  if (a_out) a_out[i] = av;
  float t = av;
  if (rstd) t = ((av - mean[c]) * rstd[c]) * w[c];
  if (b) t += b[c];
  if (res) t += res[res_ld ? (size_t)row * res_ld + c : i];
  if (relu) t = t > 0.f ? t : 0.f;
  if (y) y[i] = t;
  if (y2) y2[(size_t)row * y2_ld + c] = t;
  }
}

// Forward of a TRAIN-mode BatchNorm (+ residual, + ReLU) whose workgroups first add the one-pass statistics' partial
// rows up themselves (k_bn_stats_rows without its tail: part [nparts][2][C] doubles = sum a, sum a^2 per row block) --
// mean, biased variance = E[a^2] - mean^2, rstd exactly as that tail computes them; workgroup 0 also writes mean /
// rstd (the sweeps read them) and moves the running statistics.  fp32 NHWC, C % 4 == 0, `a` already summed.
__global__ __launch_bounds__(BLOCK) void k_bn_forward_train(
    float* __restrict__ y, float* __restrict__ y2, unsigned y2_ld, const float* __restrict__ a,
    const double* __restrict__ part, unsigned nparts, float* __restrict__ mean_out, float* __restrict__ rstd_out,
    float* __restrict__ run_mean, float* __restrict__ run_var, double count, float eps, float momentum,
    const float* __restrict__ w, const float* __restrict__ b, const float* __restrict__ res, unsigned res_ld,
    int relu, unsigned total, unsigned C) {
  struct alignas(16) D2 { double e[2]; };
  __shared__ D2 scratch[2 * BLOCK];  // [row group][column pair], C <= 2 * BLOCK column pairs
  __shared__ double fin[8 * BLOCK];   // sum a | sum a^2, C <= 4 * BLOCK each
  __shared__ float mus[4 * BLOCK], rss[4 * BLOCK];
  // this thread's element quad first (independent of the statistics)
  const unsigned quads_total = total >> 2;
  const unsigned v = blockIdx.x * BLOCK + threadIdx.x;
  const bool have = v < quads_total;
  const unsigned i = have ? v << 2 : 0u;
  const unsigned row = i / C, c = i - row * C;
  const F4 av = ld4(a + i);
  const F4 w4 = ld4(w + c);
  F4 b4, r4;
  if (b) b4 = ld4(b + c);
  if (res) r4 = ld4(res + (res_ld ? row * res_ld + c : i));
  // column sums of the [nparts][2C] matrix of doubles, as pairs: thread (tx, ty) adds rows ty, ty + G, ... of column
  // pair tx (+ lanes, ...), four rows in flight, fixed order; the row groups are combined through LDS
  const unsigned CP = C;  // pairs of doubles per row
  const unsigned lanes = CP < BLOCK ? CP : BLOCK, G = BLOCK / lanes;
  const unsigned tx = threadIdx.x % lanes, ty = threadIdx.x / lanes;
  const D2* rows2 = reinterpret_cast<const D2*>(part);
  for (unsigned col = tx; col < CP; col += lanes) {
    D2 acc = {{0.0, 0.0}};
    if (ty < G) {
      for (unsigned p0 = ty; p0 < nparts; p0 += 4 * G) {
        D2 t4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) t4[u] = rows2[(size_t)(p0 + u * G < nparts ? p0 + u * G : p0) * CP + col];
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (p0 + u * G < nparts) { acc.e[0] += t4[u].e[0]; acc.e[1] += t4[u].e[1]; }
      }
      if (G > 1) scratch[ty * lanes + col] = acc;
      else { fin[2 * col] = acc.e[0]; fin[2 * col + 1] = acc.e[1]; }
    }
  }
  __syncthreads();
  if (G > 1) {
    for (unsigned j = threadIdx.x; j < 2 * C; j += BLOCK) {
      double sum = 0.0;
      for (unsigned t = 0; t < G; ++t) sum += scratch[t * lanes + (j >> 1)].e[j & 1];
      fin[j] = sum;
    }
    __syncthreads();
  }
  for (unsigned ch = threadIdx.x; ch < C; ch += BLOCK) {
    const double m = fin[ch] / count;
    double var = fin[C + ch] / count - m * m;
    if (var < 0.0) var = 0.0;
    const float mf = (float)m, rf = (float)(1.0 / sqrt(var + (double)eps));
    mus[ch] = mf;
    rss[ch] = rf;
    if (blockIdx.x == 0) {
      mean_out[ch] = mf;
      rstd_out[ch] = rf;
      if (momentum >= 0.f && run_mean && run_var) {
        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        run_mean[ch] = (float)((1.0 - (double)momentum) * (double)run_mean[ch] + (double)momentum * (double)mf);
        run_var[ch] = (float)((1.0 - (double)momentum) * (double)run_var[ch] + (double)momentum * unbiased);
      }
    }
  }
  __syncthreads();
  if (!have) return;
  F4 o;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float t = ((av.e[k] - mus[c + k]) * rss[c + k]) * w4.e[k];
    if (b) t += b4.e[k];
    if (res) t += r4.e[k];
    if (relu) t = t > 0.f ? t : 0.f;
    o.e[k] = t;
  }
  if (y) *reinterpret_cast<F4*>(y + i) = o;
  if (y2) *reinterpret_cast<F4*>(y2 + (size_t)row * y2_ld + c) = o;
}

// Train-mode BatchNorm inside the curvature product: the batch statistics depend on the layer input, so
// tangent and adjoint carry two per-channel corrections,
//     xhat' = rstd * [a' - mean(a') - xhat * mean(xhat * a')]          (the same operator for the adjoint),
// which fold into the per-channel vectors of the elementwise kernel k_chan_affine (t = a*(w*rstd) + xhat*q + r):
//     q[c] = vq[c] - w[c]*rstd[c] * S_x[c]/m ,   r[c] = vr[c] - w[c]*rstd[c] * S_1[c]/m
// with S_x = sum(xhat * a'), S_1 = sum(a') given as `nparts` partial sums (the row shares of k_bn_adjoint_rows),
// added up here in order.  One tiny launch per layer and sweep.
// Batch statistics of a train-mode BatchNorm from per-row-block partial sums (what hf_chan_affine_bwd_ex leaves in
// gb / gw), added in order in fp64.  stage 0: mean = sum(part) / count.  stage 1: part holds sum a*(a - mean):
// var = sum / count (biased, what the layer normalises with), rstd = 1 / sqrt(var + eps), and -- momentum >= 0 -- the
// running statistics move as torch.nn.BatchNorm2d's forward moves them (unbiased variance).
__global__ __launch_bounds__(BLOCK) void k_bn_batch_stats(float* __restrict__ mean, float* __restrict__ rstd,
                                                         float* __restrict__ run_mean, float* __restrict__ run_var,
                                                         const float* __restrict__ part, int nparts, double count,
                                                         float eps, float momentum, int stage, int C) {
  const int c = blockIdx.x * BLOCK + threadIdx.x;
  if (c >= C) return;
  double s = 0.0;
  for (int k = 0; k < nparts; ++k) s += (double)part[(size_t)k * C + c];
  if (stage == 0) {
    mean[c] = (float)(s / count);
    return;
  }
  double var = s / count;
  if (var < 0.0) var = 0.0;
  rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (momentum >= 0.f && run_mean && run_var) {
    const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
    run_mean[c] = (float)((1.0 - (double)momentum) * (double)run_mean[c] + (double)momentum * (double)mean[c]);
    run_var[c] = (float)((1.0 - (double)momentum) * (double)run_var[c] + (double)momentum * unbiased);
  }
}

__global__ __launch_bounds__(BLOCK) void k_bn_train_coeffs(float* __restrict__ q_out, float* __restrict__ r_out,
                                                           const float* __restrict__ part_x,
                                                           const float* __restrict__ part_1, int nparts,
                                                           const float* __restrict__ w, const float* __restrict__ rstd,
                                                           const float* __restrict__ vq, const float* __restrict__ vr,
                                                           float inv_m, int C) {
  const int c = blockIdx.x * BLOCK + threadIdx.x;
  if (c >= C) return;
  double sx = 0.0, s1 = 0.0;
  for (int p = 0; p < nparts; ++p) {
    sx += (double)part_x[(size_t)p * C + c];
    s1 += (double)part_1[(size_t)p * C + c];
  }
  const float k = (w ? w[c] : 1.f) * rstd[c] * inv_m;
  q_out[c] = (vq ? vq[c] : 0.f) - k * (float)sx;
  r_out[c] = (vr ? vr[c] : 0.f) - k * (float)s1;
}

// Hessian of a softmax cross-entropy w.r.t. the logits, applied to v, row by row:
//   out[r, :] = scale * p[r, :] * (v[r, :] - <p[r, :], v[r, :]>)      (p = softmax(logits))
// One block per row; the dot product is accumulated in fp64.
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_softmax_ce_hvp(T* __restrict__ out,
                                                          const T* __restrict__ p,
                                                          const T* __restrict__ v, T scale,
                                                          int cols) {
  __shared__ double lds[WAVES];
  const long long base = (long long)blockIdx.x * cols;
  double acc[1] = {0.0};
  for (int j = threadIdx.x; j < cols; j += BLOCK) acc[0] += (double)p[base + j] * (double)v[base + j];
  block_allreduce<1>(acc, lds);
  const T d = (T)acc[0];
  for (int j = threadIdx.x; j < cols; j += BLOCK)
    out[base + j] = scale * (p[base + j] * (v[base + j] - d));
}

inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

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

#define HF_HIP(expr)                         \
  do {                                       \
    hipError_t e_ = (expr);                  \
    if (e_ != hipSuccess) return (int)e_;    \
  } while (0)

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
  static long long min_n = -1;
  if (min_n < 0) { const char* e = getenv("HF_PCG_NT_MIN"); min_n = e ? atoll(e) : 16000000LL; }
  return h->n >= min_n;
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
  const bool nt = false;  // (see HF_LD: K3's operands are on-die)
  (void)nt_streams;
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

// ---- two captured graphs as ONE launch with a hand-over event in between -----------------------------------
struct hf_graph_chain {
  hipGraph_t graph;
  hipGraphExec_t exec;
  hipEvent_t mid;
};

int hf_graph_chain_create(hf_graph_chain_t** out, void* graph_a, void* graph_b) {
  if (!out || !graph_a || !graph_b) return HF_ERR_ARG;
  hf_graph_chain* c = new (std::nothrow) hf_graph_chain();
  if (!c) return HF_ERR_ARG;
  memset(c, 0, sizeof(*c));
  int rc = (int)hipEventCreateWithFlags(&c->mid, hipEventDisableTiming);
  if (!rc) rc = (int)hipGraphClone(&c->graph, (hipGraph_t)graph_a);
  std::vector<hipGraphNode_t> deps;
  if (!rc) rc = graph_leaves(c->graph, deps);
  hipGraphNode_t ev = nullptr, child = nullptr;
  if (!rc) rc = (int)hipGraphAddEventRecordNode(&ev, c->graph, deps.data(), deps.size(), c->mid);
  if (!rc) rc = (int)hipGraphAddChildGraphNode(&child, c->graph, &ev, 1, (hipGraph_t)graph_b);
  if (!rc) rc = (int)hipGraphInstantiate(&c->exec, c->graph, nullptr, nullptr, 0);
  if (rc) { hf_graph_chain_destroy(c); return rc; }
  *out = c;
  return HF_OK;
}

int hf_graph_chain_launch(hf_graph_chain_t* c, void* stream) {
  if (!c || !c->exec) return HF_ERR_ARG;
  HF_HIP(hipGraphLaunch(c->exec, (hipStream_t)stream));
  return HF_OK;
}

int hf_graph_chain_wait_mid(hf_graph_chain_t* c, void* stream) {
  if (!c || !c->mid) return HF_ERR_ARG;
  HF_HIP(hipStreamWaitEvent((hipStream_t)stream, c->mid, 0));
  return HF_OK;
}

int hf_graph_chain_destroy(hf_graph_chain_t* c) {
  if (!c) return HF_OK;
  if (c->exec) (void)hipGraphExecDestroy(c->exec);
  if (c->graph) (void)hipGraphDestroy(c->graph);
  if (c->mid) (void)hipEventDestroy(c->mid);
  delete c;
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

// ---- vector helpers -------------------------------------------------------
template <typename T>
static int pack_impl(void* dst, const void* const* srcs, const int64_t* numels,
                     const int64_t* perm, const int64_t* splits, const int64_t* live, int nt,
                     double scale, int mode, hipStream_t s) {
  int t = 0;
  long long off = 0;
  while (t < nt) {
    PackArgs a;
    memset(&a, 0, sizeof(a));
    int k = 0, blocks = 0;
    while (t < nt && k < PACK_MAXT) {
      if (numels[t] < 0) return HF_ERR_ARG;
      if (numels[t] > 0) {
        if (!srcs[t]) return HF_ERR_ARG;
        a.src[k] = srcs[t];
        a.dst_off[k] = off;
        a.numel[k] = numels[t];
        a.chunk[k] = PACK_CHUNK;
        a.nsplit[k] = 1;
        if (splits) {
          if (splits[2 * t] < 1 || (splits[2 * t] > 1 && splits[2 * t + 1] < numels[t])) return HF_ERR_ARG;
          a.nsplit[k] = (int)splits[2 * t];
          a.split_stride[k] = splits[2 * t + 1];
          if (a.nsplit[k] > 1) a.chunk[k] = BLOCK * 4;  // latency-bound blocks: more of them
        }
        if (((perm && perm[2 * t] > 0) || a.nsplit[k] > 1) && numels[t] >= 0xffffffffLL) return HF_ERR_ARG;
        if (perm && perm[2 * t] > 0) {
          const int64_t I = perm[2 * t], HW = perm[2 * t + 1];
          if (HW <= 0 || numels[t] % (I * HW) != 0 || I * HW > 0x7fffffffLL) return HF_ERR_ARG;
          a.perm_I[k] = (int)I;
          a.perm_HW[k] = (int)HW;
          if (live && live[t] > 0 && HW <= 16) a.live[k] = (unsigned short)(live[t] & ((1 << HW) - 1));
          const int64_t slabs = (int64_t)(TILE_BYTES / sizeof(T)) / (I * HW + HW);
          if (slabs >= 1 && !(splits && splits[2 * t] > 1) && a.live[k] == 0)
            a.chunk[k] = (int)(slabs * I * HW);  // LDS-tiled path
          else if (a.nsplit[k] > 1 && sizeof(T) == 4 && I % 4 == 0 && I * HW <= (int64_t)(TILE_BYTES / sizeof(T)))
            a.chunk[k] = (int)(((2048 + I * HW - 1) / (I * HW)) * I * HW);  // LDS-staged stores, >= 2048 elements
          else if (a.live[k] != 0 && a.nsplit[k] == 1 && I * HW <= 2 * PACK_CHUNK)
            a.chunk[k] = (int)(((PACK_CHUNK + I * HW - 1) / (I * HW)) * I * HW);  // zero stream + live stores
        }
        a.blk_start[k] = blocks;
        blocks += (int)((numels[t] + a.chunk[k] - 1) / a.chunk[k]);
        ++k;
      }
      off += numels[t];
      ++t;
    }
    a.blk_start[k] = blocks;
    a.nt = k;
    if (blocks == 0) continue;
    if (mode == 0)
      hipLaunchKernelGGL((k_pack<T, 0>), dim3(blocks), dim3(BLOCK), 0, s, (T*)dst, a, (T)scale);
    else
      hipLaunchKernelGGL((k_pack<T, 1>), dim3(blocks), dim3(BLOCK), 0, s, (T*)dst, a, (T)scale);
    HF_HIP(hipGetLastError());
  }
  return HF_OK;
}

int hf_pack(void* dst, const void* const* srcs, const int64_t* numels, const int64_t* perm,
            int n_tensors, double scale, int mode, int dtype, void* stream) {
  return hf_pack_ex(dst, srcs, numels, perm, nullptr, nullptr, n_tensors, scale, mode, dtype, stream);
}

int hf_pack_ex(void* dst, const void* const* srcs, const int64_t* numels, const int64_t* perm,
               const int64_t* splits, const int64_t* live, int n_tensors, double scale, int mode,
               int dtype, void* stream) {
  if (!dst || !srcs || !numels || n_tensors < 0 || (mode != 0 && mode != 1)) return HF_ERR_ARG;
  if (dtype == HF_F32)
    return pack_impl<float>(dst, srcs, numels, perm, splits, live, n_tensors, scale, mode,
                            (hipStream_t)stream);
  if (dtype == HF_F64)
    return pack_impl<double>(dst, srcs, numels, perm, splits, live, n_tensors, scale, mode,
                             (hipStream_t)stream);
  return HF_ERR_ARG;
}

// one element per thread: these activation-sized kernels (<= a few hundred thousand
// elements) are latency-bound, every extra grid-stride iteration adds a full round trip
static int wide_grid(int64_t n) {
  int64_t g = (n + BLOCK - 1) / BLOCK;
  if (g < 1) g = 1;
  if (g > 16384) g = 16384;
  return (int)g;
}

static int small_grid(int64_t n) {
  int64_t g = (n + BLOCK * 4 - 1) / (BLOCK * 4);
  if (g < 1) g = 1;
  if (g > 2048) g = 2048;
  return (int)g;
}

template <typename T>
static int unpack_impl(const void* src, void* const* dsts, const int64_t* src_offs,
                       const int64_t* numels, const int64_t* slabs, const int64_t* inners,
                       const int64_t* live, const int64_t* halves, int nt, hipStream_t s) {
  int t = 0;
  while (t < nt) {
    UnpackArgs a;
    int blocks = 0;
    t = hf_shared::fill_unpack_args<T>(a, &blocks, t, dsts, src_offs, numels, slabs, inners, live, halves, nt);
    if (t < 0) return t;
    if (blocks == 0) continue;
    hipLaunchKernelGGL((k_unpack_tangent<T>), dim3(blocks), dim3(BLOCK), 0, s, (const T*)src, a);
    HF_HIP(hipGetLastError());
  }
  return HF_OK;
}

int hf_unpack_tangent(const void* src, void* const* dsts, const int64_t* src_offs,
                      const int64_t* numels, const int64_t* slabs, const int64_t* inners,
                      int n_tensors, int dtype, void* stream) {
  return hf_unpack_tangent_ex(src, dsts, src_offs, numels, slabs, inners, nullptr, n_tensors, dtype, stream);
}

int hf_unpack_tangent_ex(const void* src, void* const* dsts, const int64_t* src_offs,
                         const int64_t* numels, const int64_t* slabs, const int64_t* inners,
                         const int64_t* live, int n_tensors, int dtype, void* stream) {
  return hf_unpack_weights(src, dsts, src_offs, numels, slabs, inners, live, nullptr, n_tensors, dtype, stream);
}

int hf_unpack_weights(const void* src, void* const* dsts, const int64_t* src_offs,
                      const int64_t* numels, const int64_t* slabs, const int64_t* inners,
                      const int64_t* live, const int64_t* halves, int n_tensors, int dtype, void* stream) {
  if (!src || !dsts || !src_offs || !numels || !slabs || !inners || n_tensors < 0) return HF_ERR_ARG;
  if (dtype == HF_F32)
    return unpack_impl<float>(src, dsts, src_offs, numels, slabs, inners, live, halves, n_tensors,
                              (hipStream_t)stream);
  if (dtype == HF_F64)
    return unpack_impl<double>(src, dsts, src_offs, numels, slabs, inners, live, halves, n_tensors,
                               (hipStream_t)stream);
  return HF_ERR_ARG;
}

int hf_live_copy(void* full, void* compact, int scatter, const int64_t* full_offs, const int64_t* counts,
                 const int64_t* periods, const int64_t* masks, int n_segments, int dtype, void* stream) {
  if (!full || !compact || !full_offs || !counts || !periods || !masks || n_segments < 1 ||
      n_segments > LIVE_MAXS)
    return HF_ERR_ARG;
  LiveSegs a;
  memset(&a, 0, sizeof(a));
  long long total = 0;
  for (int i = 0; i < n_segments; ++i) {
    if (full_offs[i] < 0 || counts[i] < 1 || periods[i] < 0 || periods[i] > 16) return HF_ERR_ARG;
    a.full_off[i] = full_offs[i];
    a.comp_off[i] = total;
    a.hw[i] = (int)periods[i];
    if (periods[i] == 0) {
      a.nl[i] = 1;
      total += counts[i];
    } else {
      // counts[i] = elements of the weight tensor in the FULL vector (a whole number of periods)
      if (counts[i] % periods[i] != 0) return HF_ERR_ARG;
      int nl = 0;
      for (int t = 0; t < (int)periods[i]; ++t)
        if ((masks[i] >> t) & 1) a.pos[i][nl++] = t;
      if (nl < 1) return HF_ERR_ARG;
      a.nl[i] = nl;
      total += counts[i] / periods[i] * nl;
    }
  }
  a.comp_off[n_segments] = total;
  a.ns = n_segments;
  // blocks never straddle segments
  long long blocks = 0;
  for (int i = 0; i < n_segments; ++i) {
    if (a.comp_off[i + 1] - a.comp_off[i] >= 0xffffffffLL) return HF_ERR_ARG;
    a.blk_start[i] = (int)blocks;
    blocks += (a.comp_off[i + 1] - a.comp_off[i] + LIVE_CHUNK - 1) / LIVE_CHUNK;
  }
  a.blk_start[n_segments] = (int)blocks;
  if (blocks < 1 || blocks > 0x7fffffffLL) return HF_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == HF_F32) {
    if (scatter) hipLaunchKernelGGL((k_live_copy<float, true>), dim3((unsigned)blocks), dim3(BLOCK), 0, s, (float*)full, (float*)compact, a);
    else hipLaunchKernelGGL((k_live_copy<float, false>), dim3((unsigned)blocks), dim3(BLOCK), 0, s, (float*)full, (float*)compact, a);
  } else if (dtype == HF_F64) {
    if (scatter) hipLaunchKernelGGL((k_live_copy<double, true>), dim3((unsigned)blocks), dim3(BLOCK), 0, s, (double*)full, (double*)compact, a);
    else hipLaunchKernelGGL((k_live_copy<double, false>), dim3((unsigned)blocks), dim3(BLOCK), 0, s, (double*)full, (double*)compact, a);
  } else {
    return HF_ERR_ARG;
  }
  HF_HIP(hipGetLastError());
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

int hf_bn_adjoint_pre(void* g_out, void* ga_out, const void* gy_a, int a_splits, int64_t a_slab,
                      const void* gy_b, int b_splits, int64_t b_slab, const void* mask_src, const void* w,
                      const void* rstd, int64_t rows, int64_t c, int dtype, void* stream) {
  if (!gy_a || (!g_out && !ga_out) || rows <= 0 || c <= 0 || a_splits < 1 || b_splits < 1) return HF_ERR_ARG;
  if (ga_out && !rstd) return HF_ERR_ARG;
  if ((a_splits > 1 && a_slab <= 0) || (gy_b && b_splits > 1 && b_slab <= 0)) return HF_ERR_ARG;
  const long long total = (long long)rows * c;
  if (total > 0x7fffffffLL) return HF_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == HF_F32)
    hipLaunchKernelGGL((k_bn_adjoint_pre<float>), dim3(wide_grid(total)), dim3(BLOCK), 0, s, (float*)g_out,
                       (float*)ga_out, (const float*)gy_a, a_splits, (long long)a_slab, (const float*)gy_b,
                       b_splits, (long long)b_slab, (const float*)mask_src, (const float*)w, (const float*)rstd,
                       (unsigned)total, (unsigned)c);
  else if (dtype == HF_F64)
    hipLaunchKernelGGL((k_bn_adjoint_pre<double>), dim3(wide_grid(total)), dim3(BLOCK), 0, s, (double*)g_out,
                       (double*)ga_out, (const double*)gy_a, a_splits, (long long)a_slab, (const double*)gy_b,
                       b_splits, (long long)b_slab, (const double*)mask_src, (const double*)w,
                       (const double*)rstd, (unsigned)total, (unsigned)c);
  else
    return HF_ERR_ARG;
  HF_HIP(hipGetLastError());
  return HF_OK;
}

int hf_bn_train_coeffs(void* q_out, void* r_out, const void* part_x, const void* part_1, int nparts, const void* w,
                       const void* rstd, const void* vq, const void* vr, double count, int64_t c, int dtype,
                       void* stream) {
  if (dtype != HF_F32 || !q_out || !r_out || !part_x || !part_1 || !rstd || nparts < 1 || c < 1 || !(count > 0))
    return HF_ERR_ARG;
  hipLaunchKernelGGL(k_bn_train_coeffs, dim3((unsigned)((c + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, (hipStream_t)stream,
                     (float*)q_out, (float*)r_out, (const float*)part_x, (const float*)part_1, nparts, (const float*)w,
                     (const float*)rstd, (const float*)vq, (const float*)vr, (float)(1.0 / count), (int)c);
  HF_HIP(hipGetLastError());
  return HF_OK;
}

int hf_bn_batch_stats(void* mean, void* rstd, void* running_mean, void* running_var, const void* part, int nparts,
                      double count, double eps, double momentum, int stage, int64_t c, int dtype, void* stream) {
  if (dtype != HF_F32 || !mean || !part || nparts < 1 || c < 1 || !(count > 0) || stage < 0 || stage > 1)
    return HF_ERR_ARG;
  if (stage == 1 && (!rstd || !(eps >= 0))) return HF_ERR_ARG;
  hipLaunchKernelGGL(k_bn_batch_stats, dim3((unsigned)((c + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, (hipStream_t)stream,
                     (float*)mean, (float*)rstd, (float*)running_mean, (float*)running_var, (const float*)part, nparts,
                     count, (float)eps, (float)momentum, stage, (int)c);
  HF_HIP(hipGetLastError());
  return HF_OK;
}

int hf_bn_forward(void* y, void* y2, int64_t y2_ld, void* a_out, const void* a, int splits, int64_t slab_stride,
                  const void* mean, const void* rstd, const void* w, const void* b, const void* res,
                  int64_t res_ld, int relu, int64_t rows, int64_t c, int dtype, void* stream) {
  if (dtype != HF_F32 || !a || (!y && !y2) || rows <= 0 || c <= 0 || splits < 1) return HF_ERR_ARG;
  if (splits > 1 && slab_stride < rows * c) return HF_ERR_ARG;
  if (rstd && (!mean || !w)) return HF_ERR_ARG;
  if ((y2 && y2_ld < c) || (res && res_ld && res_ld < c)) return HF_ERR_ARG;
  const long long total = (long long)rows * c;
  const long long widest = (long long)rows * (y2_ld > res_ld ? (y2_ld > c ? y2_ld : c) : (res_ld > c ? res_ld : c));
  if (total > 0x7fffffffLL || widest > 0x7fffffffLL || y2_ld > 0x3fffffffLL || res_ld > 0x3fffffffLL) return HF_ERR_ARG;
  hipLaunchKernelGGL(k_bn_forward, dim3(wide_grid(total)), dim3(BLOCK), 0, (hipStream_t)stream, (float*)y,
                     (float*)y2, (unsigned)y2_ld, (float*)a_out, (const float*)a, splits, (long long)slab_stride,
                     (const float*)mean, (const float*)rstd, (const float*)w, (const float*)b, (const float*)res,
                     (unsigned)res_ld, relu, (unsigned)total, (unsigned)c);
  HF_HIP(hipGetLastError());
  return HF_OK;
}

int hf_bn_forward_train(void* y, void* y2, int64_t y2_ld, const void* a, const void* part, int nparts, void* mean,
                        void* rstd, void* running_mean, void* running_var, double count, double eps, double momentum,
                        const void* w, const void* b, const void* res, int64_t res_ld, int relu, int64_t rows,
                        int64_t c, int dtype, void* stream) {
  if (dtype != HF_F32 || !a || (!y && !y2) || !part || nparts < 1 || !mean || !rstd || !w || count <= 0.0 ||
      rows <= 0 || c <= 0)
    return HF_ERR_ARG;
  if (!(c % 4 == 0 && c / 4 <= BLOCK)) return HF_ERR_ARG;
  if ((y2 && (y2_ld < c || (y2_ld & 3))) || (res && res_ld && (res_ld < c || (res_ld & 3)))) return HF_ERR_ARG;
  const long long total = (long long)rows * c;
  const long long widest = (long long)rows * (y2_ld > res_ld ? (y2_ld > c ? y2_ld : c) : (res_ld > c ? res_ld : c));
  if (total > 0x7fffffffLL || widest > 0x7fffffffLL) return HF_ERR_ARG;
  const void* al[] = {y, y2, a, part, w, b, res};
  for (const void* p : al)
    if (p && !aligned16(p)) return HF_ERR_ALIGN;
  // (the grid covers every quad: one per thread, as the prologue's sums are per workgroup anyway)
  const long long wgs = (total / 4 + BLOCK - 1) / BLOCK;
  if (wgs > 0x7fffffLL) return HF_ERR_ARG;
  hipLaunchKernelGGL(k_bn_forward_train, dim3((unsigned)wgs), dim3(BLOCK), 0, (hipStream_t)stream, (float*)y,
                     (float*)y2, (unsigned)y2_ld, (const float*)a, (const double*)part, (unsigned)nparts, (float*)mean,
                     (float*)rstd, (float*)running_mean, (float*)running_var, count, (float)eps, (float)momentum,
                     (const float*)w, (const float*)b, (const float*)res, (unsigned)res_ld, relu, (unsigned)total,
                     (unsigned)c);
  HF_HIP(hipGetLastError());
  return HF_OK;
}

int hf_softmax_ce_hvp(void* out, const void* p, const void* v, double scale, int64_t rows,
                      int64_t cols, int dtype, void* stream) {
  if (!out || !p || !v || rows <= 0 || cols <= 0 || cols > 0x7fffffffLL || rows > 0x7fffffffLL)
    return HF_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == HF_F32)
    hipLaunchKernelGGL((k_softmax_ce_hvp<float>), dim3((unsigned)rows), dim3(BLOCK), 0, s, (float*)out,
                       (const float*)p, (const float*)v, (float)scale, (int)cols);
  else if (dtype == HF_F64)
    hipLaunchKernelGGL((k_softmax_ce_hvp<double>), dim3((unsigned)rows), dim3(BLOCK), 0, s,
                       (double*)out, (const double*)p, (const double*)v, scale, (int)cols);
  else
    return HF_ERR_ARG;
  HF_HIP(hipGetLastError());
  return HF_OK;
}

static bool affine_vec4_ok(const void* out, const void* a, const void* x, const void* mean, const void* rstd,
                          const void* w, const void* q, const void* r, const void* add, const void* mask_src,
                          long long total, long long c, int nhwc, long long out_ld, long long add_ld,
                          long long a_slab) {
  const void* ptrs[] = {out, a, x, mean, rstd, w, q, r, add, mask_src};
  for (const void* p : ptrs)
    if (p && !aligned16(p)) return false;
  static const bool scalar_only = getenv("HF_AFFINE_SCALAR") != nullptr;  // (A/B switch for measurements)
  return nhwc && c % 4 == 0 && out_ld % 4 == 0 && add_ld % 4 == 0 && a_slab % 4 == 0 && 2 * total < 0x7fffffffLL &&
         !scalar_only;
}

template <typename T>
static void launch_chan_affine(hipStream_t s, void* out, const void* a, const void* x,
                               const void* mean, const void* rstd, const void* w, const void* q,
                               const void* r, const void* add, const void* mask_src,
                               int relu_self, long long total, long long c, long long hw,
                               int nhwc, long long out_ld, long long add_ld, int a_splits = 1,
                               long long a_slab = 0) {
  if (sizeof(T) == 4 && affine_vec4_ok(out, a, x, mean, rstd, w, q, r, add, mask_src, total, c, nhwc || hw == 1,
                                       out_ld, add_ld, a_slab))
    hipLaunchKernelGGL(k_chan_affine_v4, dim3(wide_grid(total / 4)), dim3(BLOCK), 0, s, (float*)out,
                       (const float*)a, (const float*)x, (const float*)mean, (const float*)rstd, (const float*)w,
                       (const float*)q, (const float*)r, (const float*)add, (const float*)mask_src, relu_self,
                       (unsigned)total, (unsigned)c, (unsigned)out_ld, (unsigned)add_ld, a_splits, a_slab);
  else if (2 * total < 0x7fffffffLL)  // strided operands reach at most 2*total
    hipLaunchKernelGGL((k_chan_affine<T, unsigned>), dim3(wide_grid(total)), dim3(BLOCK), 0, s,
                       (T*)out, (const T*)a, (const T*)x, (const T*)mean, (const T*)rstd,
                       (const T*)w, (const T*)q, (const T*)r, (const T*)add, (const T*)mask_src,
                       relu_self, (unsigned)total, (unsigned)c, (unsigned)hw, nhwc,
                       (unsigned)out_ld, (unsigned)add_ld, a_splits, a_slab);
  else
    hipLaunchKernelGGL((k_chan_affine<T, long long>), dim3(wide_grid(total)), dim3(BLOCK), 0, s,
                       (T*)out, (const T*)a, (const T*)x, (const T*)mean, (const T*)rstd,
                       (const T*)w, (const T*)q, (const T*)r, (const T*)add, (const T*)mask_src,
                       relu_self, total, c, hw, nhwc, out_ld, add_ld, a_splits, a_slab);
}

int hf_chan_affine(void* out, const void* a, const void* x, const void* mean, const void* rstd,
                   const void* w, const void* q, const void* r, const void* add,
                   const void* mask_src, int relu_self, int64_t n, int64_t c, int64_t hw,
                   int channels_last, int64_t out_ld, int64_t add_ld, int dtype, void* stream) {
  return hf_chan_affine_ex(out, a, x, mean, rstd, w, q, r, add, mask_src, relu_self, n, c, hw, channels_last,
                           out_ld, add_ld, 1, 0, dtype, stream);
}

int hf_chan_affine_ex(void* out, const void* a, const void* x, const void* mean, const void* rstd,
                      const void* w, const void* q, const void* r, const void* add,
                      const void* mask_src, int relu_self, int64_t n, int64_t c, int64_t hw,
                      int channels_last, int64_t out_ld, int64_t add_ld, int a_splits, int64_t a_slab,
                      int dtype, void* stream) {
  if (a_splits < 1 || (a_splits > 1 && (!a || a_slab <= 0))) return HF_ERR_ARG;
  if (!out || n <= 0 || c <= 0 || hw <= 0) return HF_ERR_ARG;
  if (q && (!x || !mean || !rstd)) return HF_ERR_ARG;
  // a leading dimension is that of a buffer with MORE channels: >= 2x would be the
  // tangent buffers' case, anything above the dense one is accepted
  const int64_t dense = channels_last ? c : c * hw;
  if ((out_ld && out_ld < dense) || (add_ld && (add_ld < dense || !add)) ||
      out_ld > 0x3fffffffLL || add_ld > 0x3fffffffLL)
    return HF_ERR_ARG;
  const long long total = (long long)n * c * hw;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == HF_F32)
    launch_chan_affine<float>(s, out, a, x, mean, rstd, w, q, r, add, mask_src, relu_self, total, c, hw,
                              channels_last, out_ld, add_ld, a_splits, a_slab);
  else if (dtype == HF_F64)
    launch_chan_affine<double>(s, out, a, x, mean, rstd, w, q, r, add, mask_src, relu_self, total, c, hw,
                               channels_last, out_ld, add_ld, a_splits, a_slab);
  else
    return HF_ERR_ARG;
  HF_HIP(hipGetLastError());
  return HF_OK;
}

static int fill_aff_train(AffTrainArgs& q, void* out, const void* a, const void* x, const void* mean, const void* rstd,
                          const void* w, const void* part_x, const void* part_1, int nparts, const void* vq,
                          const void* vr, double count, const void* add, const void* mask_src, int64_t n, int64_t c,
                          int64_t hw, int64_t out_ld, int64_t add_ld, int a_splits, int64_t a_slab, int dtype) {
  if (!out || !a || !x || !mean || !rstd || !w || !part_x || !part_1 || nparts < 1 || count <= 0.0 || n <= 0 || c <= 0 ||
      hw <= 0 || a_splits < 1 || (a_splits > 1 && a_slab <= 0) || dtype != HF_F32)
    return HF_ERR_ARG;
  if (!(c % 4 == 0 && c / 4 <= BLOCK)) return HF_ERR_ARG;
  if ((out_ld && out_ld < c) || (add_ld && (add_ld < c || !add)) || out_ld > 0x3fffffffLL || add_ld > 0x3fffffffLL)
    return HF_ERR_ARG;
  const long long total = (long long)n * c * hw;
  if (!affine_vec4_ok(out, a, x, mean, rstd, w, nullptr, nullptr, add, mask_src, total, c, 1, out_ld, add_ld, a_slab) ||
      !aligned16(part_x) || !aligned16(part_1))
    return HF_ERR_ALIGN;
  q = AffTrainArgs{(float*)out, (const float*)a, (const float*)x, (const float*)mean, (const float*)rstd,
                   (const float*)w, (const float*)part_x, (const float*)part_1, (unsigned)nparts, (const float*)vq,
                   (const float*)vr, (float)(1.0 / count), (const float*)add, (const float*)mask_src, (unsigned)total,
                   (unsigned)c, (unsigned)out_ld, (unsigned)add_ld, a_splits, (long long)a_slab};
  return HF_OK;
}

int hf_chan_affine_train(void* out, const void* a, const void* x, const void* mean, const void* rstd, const void* w,
                         const void* part_x, const void* part_1, int nparts, const void* vq, const void* vr,
                         double count, const void* add, const void* mask_src, int64_t n, int64_t c, int64_t hw,
                         int64_t out_ld, int64_t add_ld, int a_splits, int64_t a_slab, int dtype, void* stream) {
  AffTrainArgs q;
  const int rc = fill_aff_train(q, out, a, x, mean, rstd, w, part_x, part_1, nparts, vq, vr, count, add, mask_src, n, c,
                                hw, out_ld, add_ld, a_splits, a_slab, dtype);
  if (rc) return rc;
  typedef void (*Kern)(AffTrainArgs);
  static const Kern kerns[8] = {
      k_chan_affine_v4_train<false, false, false>, k_chan_affine_v4_train<true, false, false>,
      k_chan_affine_v4_train<false, true, false>,  k_chan_affine_v4_train<true, true, false>,
      k_chan_affine_v4_train<false, false, true>,  k_chan_affine_v4_train<true, false, true>,
      k_chan_affine_v4_train<false, true, true>,   k_chan_affine_v4_train<true, true, true>};
  hipLaunchKernelGGL(kerns[(a_splits > 1 ? 1 : 0) | (add ? 2 : 0) | (mask_src ? 4 : 0)],
                     dim3(wide_grid(q.total / 4)), dim3(BLOCK), 0, (hipStream_t)stream, q);
  HF_HIP(hipGetLastError());
  return HF_OK;
}

int hf_chan_affine_train_pair(const hf_affine_train_problem* problems, int dtype, void* stream) {
  if (!problems) return HF_ERR_ARG;
  AffTrainArgs q[2];
  for (int i = 0; i < 2; ++i) {
    const hf_affine_train_problem& p = problems[i];
    if (p.add) return HF_ERR_ARG;  // (no residual operand in the paired form)
    const int rc = fill_aff_train(q[i], p.out, p.a, p.x, p.mean, p.rstd, p.w, p.part_x, p.part_1, p.nparts, p.vq, p.vr,
                                  p.count, nullptr, p.mask_src, p.n, p.c, p.hw, p.out_ld, 0, p.a_splits, p.a_slab,
                                  dtype);
    if (rc) return rc;
  }
  typedef void (*Kern)(AffTrainArgs, AffTrainArgs, unsigned);
#define HF_ATP(SA, MA, SB, MB) k_chan_affine_v4_train_pair<SA, MA, SB, MB>
  static const Kern kerns[16] = {
      HF_ATP(false, false, false, false), HF_ATP(true, false, false, false), HF_ATP(false, true, false, false),
      HF_ATP(true, true, false, false),   HF_ATP(false, false, true, false), HF_ATP(true, false, true, false),
      HF_ATP(false, true, true, false),   HF_ATP(true, true, true, false),   HF_ATP(false, false, false, true),
      HF_ATP(true, false, false, true),   HF_ATP(false, true, false, true),  HF_ATP(true, true, false, true),
      HF_ATP(false, false, true, true),   HF_ATP(true, false, true, true),   HF_ATP(false, true, true, true),
      HF_ATP(true, true, true, true)};
#undef HF_ATP
  const unsigned ba = (unsigned)wide_grid(q[0].total / 4), bb = (unsigned)wide_grid(q[1].total / 4);
  const int idx = (q[0].a_splits > 1 ? 1 : 0) | (q[0].mask_src ? 2 : 0) | (q[1].a_splits > 1 ? 4 : 0) |
                  (q[1].mask_src ? 8 : 0);
  hipLaunchKernelGGL(kerns[idx], dim3(ba + bb), dim3(BLOCK), 0, (hipStream_t)stream, q[0], q[1], ba);
  HF_HIP(hipGetLastError());
  return HF_OK;
}

int hf_chan_affine_pair(const hf_affine_problem* problems, int dtype, void* stream) {
  if (!problems || dtype != HF_F32) return HF_ERR_ARG;
  AffArgs q[2];
  unsigned blocks[2];
  for (int i = 0; i < 2; ++i) {
    const hf_affine_problem& p = problems[i];
    if (p.a_splits < 1 || (p.a_splits > 1 && (!p.a || p.a_slab <= 0))) return HF_ERR_ARG;
    if (!p.out || p.n <= 0 || p.c <= 0 || p.hw <= 0 || (p.q && (!p.x || !p.mean || !p.rstd))) return HF_ERR_ARG;
    const int64_t dense = p.c;  // NHWC
    if ((p.out_ld && p.out_ld < dense) || (p.add_ld && (p.add_ld < dense || !p.add)) ||
        p.out_ld > 0x3fffffffLL || p.add_ld > 0x3fffffffLL)
      return HF_ERR_ARG;
    const long long total = (long long)p.n * p.c * p.hw;
    if (2 * total >= 0x7fffffffLL) return HF_ERR_ARG;
    q[i] = AffArgs{(float*)p.out, (const float*)p.a, (const float*)p.x, (const float*)p.mean,
                   (const float*)p.rstd, (const float*)p.w, (const float*)p.q, (const float*)p.r,
                   (const float*)p.add, (const float*)p.mask_src, p.relu_self, (unsigned)total, (unsigned)p.c,
                   (unsigned)p.hw, 1, (unsigned)p.out_ld, (unsigned)p.add_ld, p.a_splits, (long long)p.a_slab, 0};
    q[i].vec4 = affine_vec4_ok(p.out, p.a, p.x, p.mean, p.rstd, p.w, p.q, p.r, p.add, p.mask_src, total, p.c, 1,
                               p.out_ld, p.add_ld, p.a_slab) ? 1 : 0;
    blocks[i] = (unsigned)wide_grid(q[i].vec4 ? total / 4 : total);
  }
  hipLaunchKernelGGL(k_chan_affine_pair, dim3(blocks[0] + blocks[1]), dim3(BLOCK), 0, (hipStream_t)stream, q[0],
                     q[1], blocks[0]);
  HF_HIP(hipGetLastError());
  return HF_OK;
}

int hf_chan_affine_bwd_pair(const hf_bn_adjoint_problem* problems, int dtype, void* stream) {
  if (!problems || dtype != HF_F32) return HF_ERR_ARG;
  BnAdjArgs q[2];
  unsigned blocks[2];
  for (int i = 0; i < 2; ++i) {
    const hf_bn_adjoint_problem& p = problems[i];
    if (!p.gy || p.n <= 0 || p.c <= 0 || p.hw <= 0 || p.gy_splits < 1 || p.gy2_splits < 1 || p.row_blocks < 2)
      return HF_ERR_ARG;
    if (!(p.c % 4 == 0 && p.c / 4 <= BLOCK)) return HF_ERR_ARG;
    const int64_t rows = p.n * p.hw;
    if (rows * p.c > 0x7fffffffLL || !aligned16(p.gy) || (p.gy2 && !aligned16(p.gy2)) || (p.x && !aligned16(p.x)) ||
        (p.mask_src && !aligned16(p.mask_src)) || (p.gx && !aligned16(p.gx)) || (p.gres && !aligned16(p.gres)))
      return HF_ERR_ALIGN;
    const unsigned rpb = (unsigned)((rows + p.row_blocks - 1) / p.row_blocks);
    q[i] = BnAdjArgs{(float*)p.gx, (float*)p.gw, (float*)p.gb, (float*)p.gres, (const float*)p.gy, p.gy_splits,
                     (long long)p.gy_slab, (const float*)p.gy2, p.gy2_splits, (long long)p.gy2_slab,
                     (const float*)p.x, (const float*)p.mean, (const float*)p.rstd, (const float*)p.w,
                     (const float*)p.mask_src, (unsigned)rows, (unsigned)p.c, rpb};
    blocks[i] = (unsigned)p.row_blocks;
  }
  hipLaunchKernelGGL(k_bn_adjoint_rows_pair, dim3(blocks[0] + blocks[1]), dim3(BLOCK), 0, (hipStream_t)stream,
                     q[0], q[1], blocks[0]);
  HF_HIP(hipGetLastError());
  return HF_OK;
}

template <typename T>
static void launch_chan_affine_bwd(hipStream_t s, void* gx, void* gw, void* gb, void* gres,
                                   const void* gy, const void* gy2, const void* x, const void* mean,
                                   const void* rstd, const void* w, const void* mask_src,
                                   long long n, long long c, long long hw, int nhwc, int s1 = 1,
                                   long long l1 = 0, int s2 = 1, long long l2 = 0, int row_blocks = 1) {
  const long long total = n * c * hw;
  if (nhwc && hw > 1) {
    const bool vec = c % 4 == 0 && aligned16(gy) && (!gy2 || aligned16(gy2)) && (!x || aligned16(x)) &&
                     (!mask_src || aligned16(mask_src)) &&
                     (!gx || aligned16(gx)) && (!gres || aligned16(gres)) && sizeof(T) == 4;
#define HF_BWD_CL(I, W, BS)                                                                        \
  hipLaunchKernelGGL((k_chan_affine_bwd_nhwc<T, I, W, BS>), dim3((unsigned)(c / W * row_blocks)), dim3(BS), 0, s, \
                     (T*)gx, (T*)gw, (T*)gb, (T*)gres, (const T*)gy, (const T*)gy2, (const T*)x,    \
                     (const T*)mean, (const T*)rstd, (const T*)w, (const T*)mask_src, (I)(n * hw), (I)c,  \
                     s1, l1, s2, l2, row_blocks)
    // (512- and 1024-thread blocks for the early layers' tall reductions were measured: no
    // gain; a row-major kernel with a two-level reduction (block partials + last-ticket block)
    // was correct but slower end to end (885 vs 915 matvecs/s): its extra dependent round
    // trips cost more than the coalescing wins on tensors this small)
    if (total < 0x7fffffffLL) {
      if (vec) HF_BWD_CL(unsigned, 4, BLOCK); else HF_BWD_CL(unsigned, 1, BLOCK);
    } else {
      if (vec) HF_BWD_CL(long long, 4, BLOCK); else HF_BWD_CL(long long, 1, BLOCK);
    }
#undef HF_BWD_CL
    return;
  }
  const bool small = n * hw <= 256;
#define HF_BWD(I, TPC, GRID)                                                                    \
  hipLaunchKernelGGL((k_chan_affine_bwd<T, I, TPC>), dim3((unsigned)(GRID)), dim3(BLOCK), 0, s,  \
                     (T*)gx, (T*)gw, (T*)gb, (T*)gres, (const T*)gy, (const T*)gy2, (const T*)x,  \
                     (const T*)mean, (const T*)rstd, (const T*)w, (const T*)mask_src, (I)n, (I)c, (I)hw,  \
                     s1, l1, s2, l2)
  if (total < 0x7fffffffLL) {
    if (small) HF_BWD(unsigned, 64, (c + 3) / 4); else HF_BWD(unsigned, 256, c);
  } else {
    HF_BWD(long long, 256, c);
  }
#undef HF_BWD
}

int hf_chan_affine_bwd(void* gx, void* gw, void* gb, void* gres, const void* gy, const void* gy2,
                       const void* x, const void* mean, const void* rstd, const void* w,
                       const void* mask_src, int64_t n, int64_t c, int64_t hw, int channels_last,
                       int dtype, void* stream) {
  return hf_chan_affine_bwd_ex(gx, gw, gb, gres, gy, 1, 0, gy2, 1, 0, x, mean, rstd, w, mask_src, n, c, hw,
                               channels_last, 1, dtype, stream);
}

int hf_chan_affine_bwd_ex(void* gx, void* gw, void* gb, void* gres, const void* gy, int gy_splits,
                          int64_t gy_slab, const void* gy2, int gy2_splits, int64_t gy2_slab,
                          const void* x, const void* mean, const void* rstd, const void* w,
                          const void* mask_src, int64_t n, int64_t c, int64_t hw, int channels_last,
                          int row_blocks, int dtype, void* stream) {
  if (!gy || n <= 0 || c <= 0 || hw <= 0 || gy_splits < 1 || gy2_splits < 1 || row_blocks < 1) return HF_ERR_ARG;
  // row shares: the row-major NHWC fp32 kernel
  if (row_blocks > 1) {
    if (!(channels_last && c % 4 == 0 && c / 4 <= BLOCK && dtype == HF_F32)) return HF_ERR_ARG;
    const int64_t rows = n * hw;
    if (rows * c > 0x7fffffffLL || !aligned16(gy) || (gy2 && !aligned16(gy2)) || (x && !aligned16(x)) ||
        (mask_src && !aligned16(mask_src)) || (gx && !aligned16(gx)) || (gres && !aligned16(gres)))
      return HF_ERR_ALIGN;
    const unsigned rpb = (unsigned)((rows + row_blocks - 1) / row_blocks);
    hipLaunchKernelGGL(k_bn_adjoint_rows, dim3((unsigned)row_blocks), dim3(BLOCK), 0, (hipStream_t)stream,
                       (float*)gx, (float*)gw, (float*)gb, (float*)gres, (const float*)gy, gy_splits,
                       (long long)gy_slab, (const float*)gy2, gy2_splits, (long long)gy2_slab, (const float*)x,
                       (const float*)mean, (const float*)rstd, (const float*)w, (const float*)mask_src,
                       (unsigned)rows, (unsigned)c, rpb);
    HF_HIP(hipGetLastError());
    return HF_OK;
  }
  if ((gy_splits > 1 && gy_slab <= 0) || (gy2 && gy2_splits > 1 && gy2_slab <= 0)) return HF_ERR_ARG;
  if ((gy_splits > 1 || gy2_splits > 1) && !(channels_last || hw == 1)) return HF_ERR_ARG;
  if (gw && (!x || !mean || !rstd)) return HF_ERR_ARG;  // (rstd == NULL: no BatchNorm, gx = g * w or g)
  if (!gw) x = nullptr;  // plain per-channel sums (a conv layer's bias gradient)
  hipStream_t s = (hipStream_t)stream;
  if (dtype == HF_F32)
    launch_chan_affine_bwd<float>(s, gx, gw, gb, gres, gy, gy2, x, mean, rstd, w, mask_src, n, c, hw,
                                  channels_last, gy_splits, gy_slab, gy2_splits, gy2_slab, row_blocks);
  else if (dtype == HF_F64)
    launch_chan_affine_bwd<double>(s, gx, gw, gb, gres, gy, gy2, x, mean, rstd, w, mask_src, n, c, hw,
                                   channels_last, gy_splits, gy_slab, gy2_splits, gy2_slab, row_blocks);
  else
    return HF_ERR_ARG;
  HF_HIP(hipGetLastError());
  return HF_OK;
}

int hf_bn_adjoint_rows_train(void* gx, void* gw, void* gb, void* gres, const void* gy, int gy_splits,
                             int64_t gy_slab, const void* gy2, int gy2_splits, int64_t gy2_slab, const void* x,
                             const void* mean, const void* rstd, const void* w, const void* mask_src, int64_t n,
                             int64_t c, int64_t hw, int row_blocks, void* ticket, void* q_out, void* r_out,
                             const void* final_w, const void* vq, const void* vr, double count, int dtype,
                             void* stream) {
  if (!gy || !gw || !gb || !x || !mean || !rstd || !ticket || !q_out || !r_out || n <= 0 || c <= 0 || hw <= 0 ||
      gy_splits < 1 || gy2_splits < 1 || row_blocks < 2 || count <= 0.0 || dtype != HF_F32)
    return HF_ERR_ARG;
  if (!(c % 4 == 0 && c / 4 <= BLOCK)) return HF_ERR_ARG;
  if ((gy_splits > 1 && gy_slab <= 0) || (gy2 && gy2_splits > 1 && gy2_slab <= 0)) return HF_ERR_ARG;
  const int64_t rows = n * hw;
  if (rows * c > 0x7fffffffLL) return HF_ERR_ARG;
  if (!aligned16(gy) || (gy2 && !aligned16(gy2)) || !aligned16(x) || (mask_src && !aligned16(mask_src)) ||
      (gx && !aligned16(gx)) || (gres && !aligned16(gres)))
    return HF_ERR_ALIGN;
  const unsigned rpb = (unsigned)((rows + row_blocks - 1) / row_blocks);
  TrainFinal f{(unsigned*)ticket, (float*)q_out, (float*)r_out, (const float*)final_w, (const float*)vq,
               (const float*)vr, (float)(1.0 / count)};
  hipLaunchKernelGGL(k_bn_adjoint_rows_train, dim3((unsigned)row_blocks), dim3(BLOCK), 0, (hipStream_t)stream,
                     (float*)gx, (float*)gw, (float*)gb, (float*)gres, (const float*)gy, gy_splits,
                     (long long)gy_slab, (const float*)gy2, gy2_splits, (long long)gy2_slab, (const float*)x,
                     (const float*)mean, (const float*)rstd, (const float*)w, (const float*)mask_src,
                     (unsigned)rows, (unsigned)c, rpb, f);
  HF_HIP(hipGetLastError());
  return HF_OK;
}

int hf_bn_rows_train_apply(void* out, int64_t out_ld, void* gw, void* gb, void* gres, const void* gy, int gy_splits,
                           int64_t gy_slab, const void* gy2, int gy2_splits, int64_t gy2_slab, const void* x,
                           const void* mean, const void* rstd, const void* mask_src, int64_t n, int64_t c, int64_t hw,
                           int row_blocks, void* barrier, void* q_out, void* r_out, const void* final_w,
                           const void* vq, const void* vr, double count, const void* add, int64_t add_ld,
                           const void* out_mask, int dtype, void* stream) {
  if (!out || !gy || !gw || !gb || !gres || !x || !mean || !rstd || !barrier || n <= 0 || c <= 0 || hw <= 0 ||
      gy_splits < 1 || gy2_splits < 1 || row_blocks < 1 || count <= 0.0 || dtype != HF_F32 || (!q_out != !r_out) ||
      out_ld < 0 || add_ld < 0)
    return HF_ERR_ARG;
  if (!(c % 4 == 0 && c / 4 <= BLOCK)) return HF_ERR_ARG;
  if ((gy_splits > 1 && gy_slab <= 0) || (gy2 && gy2_splits > 1 && gy2_slab <= 0)) return HF_ERR_ARG;
  const int64_t rows = n * hw;
  if (rows * (out_ld > c ? out_ld : c) > 0x7fffffffLL || rows * add_ld > 0x7fffffffLL) return HF_ERR_ARG;
  if ((out_ld && out_ld < c) || (add_ld && add_ld < c) || (out_ld & 3) || (add_ld & 3) || (gy_slab & 3) ||
      (gy2_slab & 3))
    return HF_ERR_ARG;
  // the launch waits inside itself for ALL its workgroups: they must all be resident at once
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, v = 0;
    HF_HIP(hipGetDevice(&dev));
    HF_HIP(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev));
    if (v < 1) return HF_ERR_STATE;
    cus = v;
  }
  if (row_blocks > cus) return HF_ERR_ARG;
  const void* al[] = {out, gres, gy, gy2, x, mask_src, add, out_mask};
  for (const void* p : al)
    if (p && !aligned16(p)) return HF_ERR_ALIGN;
  const unsigned rpb = (unsigned)((rows + row_blocks - 1) / row_blocks);
  TrainApply f{(unsigned long long*)barrier, (float*)out, (const float*)add, (const float*)out_mask,
               (unsigned)out_ld, (unsigned)add_ld, (float*)q_out, (float*)r_out, (const float*)final_w,
               (const float*)vq, (const float*)vr, (float)(1.0 / count)};
  hipLaunchKernelGGL(k_bn_rows_train_apply, dim3((unsigned)row_blocks), dim3(BLOCK), 0, (hipStream_t)stream,
                     (float*)gw, (float*)gb, (float*)gres, (const float*)gy, gy_splits, (long long)gy_slab,
                     (const float*)gy2, gy2_splits, (long long)gy2_slab, (const float*)x, (const float*)mean,
                     (const float*)rstd, (const float*)mask_src, (unsigned)rows, (unsigned)c, rpb, f);
  HF_HIP(hipGetLastError());
  return HF_OK;
}

int hf_bn_stats_rows(void* a_out, const void* a, int splits, int64_t slab_stride, void* part, void* ticket,
                     void* mean, void* rstd, void* running_mean, void* running_var, double count, double eps,
                     double momentum, int64_t rows, int64_t c, int row_blocks, int dtype, void* stream) {
  if (!a || !part || (ticket && (!mean || !rstd)) || splits < 1 || (splits > 1 && slab_stride <= 0) || rows <= 0 ||
      c <= 0 || row_blocks < 1 || count <= 0.0 || dtype != HF_F32)
    return HF_ERR_ARG;
  if (!(c % 4 == 0 && c / 4 <= BLOCK) || rows * c > 0x7fffffffLL) return HF_ERR_ARG;
  if (!aligned16(a) || (a_out && !aligned16(a_out)) || (slab_stride & 3)) return HF_ERR_ALIGN;
  const unsigned rpb = (unsigned)((rows + row_blocks - 1) / row_blocks);
  hipLaunchKernelGGL(k_bn_stats_rows, dim3((unsigned)row_blocks), dim3(BLOCK), 0, (hipStream_t)stream,
                     (float*)a_out, (const float*)a, splits, (long long)slab_stride, (double*)part,
                     (unsigned*)ticket, (float*)mean, (float*)rstd, (float*)running_mean, (float*)running_var,
                     count, (float)eps, (float)momentum, (unsigned)rows, (unsigned)c, rpb);
  HF_HIP(hipGetLastError());
  return HF_OK;
}

// ---- RCCL, resolved at run time -------------------------------------------
struct hf_comm {
  void* comm;  // ncclComm_t
};

namespace {
struct NcclUid { char internal[128]; };
typedef int (*fn_get_uid)(NcclUid*);
typedef int (*fn_init_rank)(void**, int, NcclUid, int);
typedef int (*fn_destroy)(void*);
typedef int (*fn_allreduce)(const void*, void*, size_t, int, int, void*, hipStream_t);

void* rccl_sym(const char* name) {
  void* f = dlsym(RTLD_DEFAULT, name);
  if (f) return f;
  // torch's extension modules are loaded RTLD_LOCAL: look the library up by its
  // SONAME among the objects already mapped into this process (never load a
  // second copy).
  static void* lib = nullptr;
  if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
  if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
  if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW);
  return lib ? dlsym(lib, name) : nullptr;
}
}  // namespace

int hf_comm_unique_id(char* out128) {
  if (!out128) return HF_ERR_ARG;
  fn_get_uid f = (fn_get_uid)rccl_sym("ncclGetUniqueId");
  if (!f) return HF_ERR_NOSYMBOL;
  NcclUid id;
  const int rc = f(&id);
  if (rc) return 1000 + rc;
  memcpy(out128, id.internal, 128);
  return HF_OK;
}

int hf_comm_create(hf_comm_t** out, const char* id128, int nranks, int rank) {
  if (!out || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return HF_ERR_ARG;
  fn_init_rank f = (fn_init_rank)rccl_sym("ncclCommInitRank");
  if (!f) return HF_ERR_NOSYMBOL;
  NcclUid id;
  memcpy(id.internal, id128, 128);
  hf_comm* c = new (std::nothrow) hf_comm();
  if (!c) return HF_ERR_ARG;
  const int rc = f(&c->comm, nranks, id, rank);
  if (rc) { delete c; return 1000 + rc; }
  *out = c;
  return HF_OK;
}

int hf_comm_destroy(hf_comm_t* c) {
  if (!c) return HF_OK;
  fn_destroy f = (fn_destroy)rccl_sym("ncclCommDestroy");
  if (f && c->comm) (void)f(c->comm);
  delete c;
  return HF_OK;
}

int hf_allreduce_sum(hf_comm_t* c, void* buf, int64_t n, int dtype, void* stream) {
  if (!c || !c->comm || !buf || n <= 0) return HF_ERR_ARG;
  fn_allreduce f = (fn_allreduce)rccl_sym("ncclAllReduce");
  if (!f) return HF_ERR_NOSYMBOL;
  // ncclFloat32 = 7, ncclFloat64 = 8, ncclSum = 0 (nccl.h / rccl.h enum values)
  const int nccl_dtype = dtype == HF_F32 ? 7 : 8;
  const int rc = f(buf, buf, (size_t)n, nccl_dtype, 0, c->comm, (hipStream_t)stream);
  return rc ? 1000 + rc : HF_OK;
}

int hf_allreduce_sum_multi(hf_comm_t* c, void* const* bufs, const int64_t* ns, int count, int dtype,
                           void* stream) {
  if (!c || !c->comm || !bufs || !ns || count < 1 || count > 16) return HF_ERR_ARG;
  if (dtype != HF_F32 && dtype != HF_F64) return HF_ERR_ARG;
  for (int i = 0; i < count; ++i)
    if (!bufs[i] || ns[i] <= 0) return HF_ERR_ARG;
  if (count == 1) return hf_allreduce_sum(c, bufs[0], ns[0], dtype, stream);
  typedef int (*fn_group)(void);
  fn_allreduce f = (fn_allreduce)rccl_sym("ncclAllReduce");
  fn_group gs = (fn_group)rccl_sym("ncclGroupStart");
  fn_group ge = (fn_group)rccl_sym("ncclGroupEnd");
  if (!f || !gs || !ge) return HF_ERR_NOSYMBOL;
  const int nccl_dtype = dtype == HF_F32 ? 7 : 8;
  int rc = gs();
  if (rc) return 1000 + rc;
  int first = 0;
  for (int i = 0; i < count; ++i) {
    rc = f(bufs[i], bufs[i], (size_t)ns[i], nccl_dtype, 0, c->comm, (hipStream_t)stream);
    if (rc && !first) first = rc;
  }
  rc = ge();  // (always closed: an open group would swallow every later collective of the communicator)
  if (first) return 1000 + first;
  return rc ? 1000 + rc : HF_OK;
}

