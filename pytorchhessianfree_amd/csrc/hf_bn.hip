// hf_bn.hip -- BatchNorm / bias (+ residual, + ReLU) as per-channel affine maps inside the curvature sweeps and the
// engine's own forward pass (gfx950): forward, tangent, adjoint; eval mode and train mode (batch statistics whose
// per-channel finalisation runs in the prologue of the launch that consumes it); the softmax-CE Hessian rows.
// Reference: what BackPACK's R-op / L-op differentiate through, hessianfree/optimizer.py:450-462.
#include "hf_common.h"

namespace {

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

// Both column sums in one pass (all loads of both partial-row sets in flight together).  scratch: 8 * BLOCK doubles.
// `between()` runs right after the first batch of partial-row loads is issued: the caller's own independent loads go
// there, so that one round trip covers both.
// HF_FCS_BATCH: how many partial rows of each set a thread keeps in flight per round trip.  The rows come out of the
// memory-side cache (~0.6-1 us per dependent round trip); the consumers of a train-mode BatchNorm add up 64 ... 256 rows
// per channel, i.e. 16 ... 32 per thread: 4 per batch = 4 ... 8 dependent round trips in the prologue of a launch that
// otherwise takes ~4.6 us.  The order of the additions does not depend on it (bitwise the same sums).  Measured, same box
// (profiles/r06_fcs_batch_ab.jsonl, bench.py --bn train): 4 -> 1 174 / 1 190, 8 -> 1 211 / 1 208, 16 -> 1 201 / 1 202
// matvecs/s.  (A column-blocked workgroup mapping -- 64 rows x 16 channels, a quarter ... a 32nd of the redundant
// partial-row traffic and one round trip -- was built and measured at NO gain: 1 196-1 199 against 1 200,
// profiles/r06_colblock_rejected.jsonl; removed.)
#ifndef HF_FCS_BATCH
#define HF_FCS_BATCH 8
#endif
template <typename Between>
__device__ __forceinline__ void final_column_sums2(const float* __restrict__ rows_a, const float* __restrict__ rows_b,
                                                   unsigned nrows, unsigned C, double* scratch, double* out_a,
                                                   double* out_b, Between&& between) {
  const unsigned quads = C / 4, G = BLOCK / quads;
  const unsigned tx = threadIdx.x % quads, ty = threadIdx.x / quads;
  const bool live = ty < G;
  double a[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  constexpr int NB = HF_FCS_BATCH;  // partial rows of each set in flight per thread and round trip
  F4 va[NB], vb[NB];
  auto issue = [&](unsigned p0) {
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const unsigned p = p0 + u * G < nrows ? p0 + u * G : 0u;
      va[u] = ld4(rows_a + (size_t)p * C + 4 * tx);
      vb[u] = ld4(rows_b + (size_t)p * C + 4 * tx);
    }
  };
  auto add = [&](unsigned p0) {  // NB partial rows of each set, added in row order
#pragma unroll
    for (int u = 0; u < NB; ++u)
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
    for (int u = 0; u < NB; ++u) {
      asm volatile("" : "+v"(va[u].e[0]), "+v"(va[u].e[1]), "+v"(va[u].e[2]), "+v"(va[u].e[3]) : : "memory");
      asm volatile("" : "+v"(vb[u].e[0]), "+v"(vb[u].e[1]), "+v"(vb[u].e[2]), "+v"(vb[u].e[3]) : : "memory");
    }
    add(ty);
  }
  if (live) {
    for (unsigned p0 = ty + NB * G; p0 < nrows; p0 += NB * G) { issue(p0); add(p0); }
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
// a_out (row-major walk as k_bn_adjoint_rows), per-channel sum a and sum a^2 in fp64 per thread / block.
// part: [gridDim.x, 2, C] doubles; the normalising launch (k_bn_forward_train) adds the rows up in its prologue:
// mean, biased variance = E[a^2] - mean^2 (fp64: 1e-16 * mean^2/var relative, far below fp32 for any layer a network
// can train), rstd, running statistics.
__global__ __launch_bounds__(BLOCK) void k_bn_stats_rows(
    float* __restrict__ a_out, const float* __restrict__ a, int splits, long long slab, double* part,
    unsigned rows, unsigned C, unsigned rows_per_block) {
  struct alignas(16) Col { float e[4]; };
  __shared__ double red[BLOCK * 8];
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
    part[((size_t)blockIdx.x * 2 + (k & 1)) * C + col * 4 + (k >> 1)] = sum;
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

// ---- Hessian product through a TRAIN-mode BatchNorm (forward-over-reverse, optimizer.py:450-455) --------------
// z = gamma * xhat + beta, xhat = (a - mean(a)) * rstd(a).  With the step's first-order cotangents g_z (masked), g_a
// and the tangent sweep's a' (dot quantities: derivatives along v; m = rows):
//   S1 = mean(a'),  Sx = mean(a' xhat),  xhat' = rstd (a' - S1 - xhat Sx),  rstd'/rstd = -rstd Sx
//   g_gamma' = sum(g_z' xhat + g_z xhat') = sum(g_z' xhat) + rstd sum(g_z a') - rstd (S1 g_beta + Sx g_gamma)
//   g_a'     = (rstd'/rstd) g_a + rstd [ G' - mean(G') - xhat' mean(G xhat) - xhat mean(G' xhat + G xhat') ],
//              G = gamma g_z,  G' = v_gamma g_z + gamma g_z'
//            = c0 g_a + c1 g_z + c2 g_z' + c3 a' + c4 xhat + c5          (per-channel c0 ... c5, below)
// k_bn_train_hessian_coeffs adds the partial rows of the five row reductions up (fp64, row order) and writes the six
// coefficient vectors and the closed-form share of g_gamma' as one more partial row for the gather;
// k_bn_train_hessian_apply is the elementwise pass (a' arrives as the tangent convolution's split-K slabs).
__global__ __launch_bounds__(BLOCK) void k_bn_train_hessian_coeffs(
    float* __restrict__ coef, float* __restrict__ gw_corr, const float* __restrict__ sum_gx2,
    const float* __restrict__ sum_g2, const float* __restrict__ sum_ga, const float* __restrict__ sum_tx,
    const float* __restrict__ sum_t1, int nparts, int nparts_t, const float* __restrict__ g_gamma1,
    const float* __restrict__ g_beta1, const float* __restrict__ gamma, const float* __restrict__ v_gamma,
    const float* __restrict__ rstd, double count, int C) {
  // 8 channels x 32 row lanes per workgroup: lane rl adds rows rl, rl + 32, ... (fp64, increasing), the 32 shares are
  // combined in lane order -- a fixed summation order, and at most nparts / 32 dependent round trips instead of nparts
  // (one thread per channel walking up to 256 rows: 530 matvecs/s on the train-mode ResNet-18; 8 row lanes: 628)
  constexpr int CH = 8, RL = BLOCK / CH;
  __shared__ double red[5][RL][CH];
  const int cl = threadIdx.x % CH, rl = threadIdx.x / CH, c = blockIdx.x * CH + cl;
  double s[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  if (c < C) {
    for (int p = rl; p < nparts; p += RL) {
      s[0] += (double)sum_gx2[(size_t)p * C + c];
      s[1] += (double)sum_g2[(size_t)p * C + c];
      s[2] += (double)sum_ga[(size_t)p * C + c];
    }
    for (int p = rl; p < nparts_t; p += RL) {  // (the tangent sweep's sums: a reduction launch's or the convolution epilogue's)
      s[3] += (double)sum_tx[(size_t)p * C + c];
      s[4] += (double)sum_t1[(size_t)p * C + c];
    }
  }
#pragma unroll
  for (int k = 0; k < 5; ++k) red[k][rl][cl] = s[k];
  __syncthreads();
  if (rl != 0 || c >= C) return;
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    double t = red[k][0][cl];
    for (int j = 1; j < RL; ++j) t += red[k][j][cl];
    s[k] = t;
  }
  const double s_gx = s[0], s_g = s[1], s_ga = s[2], s_tx = s[3], s_t1 = s[4];
  const double r = rstd[c], gam = gamma[c], dgam = v_gamma[c], gg = g_gamma1[c], gb = g_beta1[c];
  const double S1 = s_t1 / count, Sx = s_tx / count;
  const double corr = -r * (S1 * gb + Sx * gg);
  const double dgg = s_gx + s_ga + corr;                  // g_gamma'
  const double mG = (dgam * gb + gam * s_g) / count;      // mean(G')
  const double m2 = (dgam * gg + gam * dgg) / count;      // mean(G' xhat + G xhat')
  const double mGx = gam * gg / count;                    // mean(G xhat)
  coef[0 * C + c] = (float)(-r * Sx);
  coef[1 * C + c] = (float)(r * dgam);
  coef[2 * C + c] = (float)(r * gam);
  coef[3 * C + c] = (float)(-r * r * mGx);
  coef[4 * C + c] = (float)(r * r * mGx * Sx - r * m2);
  coef[5 * C + c] = (float)(-r * mG + r * r * mGx * S1);
  gw_corr[c] = (float)corr;
}

__global__ __launch_bounds__(BLOCK) void k_bn_train_hessian_apply(
    float* __restrict__ out, const float* __restrict__ ga1, const float* __restrict__ gz1,
    const float* __restrict__ gz2, const float* __restrict__ t, int t_splits, long long t_slab,
    const float* __restrict__ a, const float* __restrict__ mean, const float* __restrict__ rstd,
    const float* __restrict__ coef, unsigned total4, unsigned C) {
  for (unsigned q = blockIdx.x * BLOCK + threadIdx.x; q < total4; q += gridDim.x * BLOCK) {
    const unsigned i = 4 * q, c = i % C;
    F4 k0 = ld4(coef + c), k1 = ld4(coef + C + c), k2 = ld4(coef + 2 * C + c), k3 = ld4(coef + 3 * C + c),
       k4 = ld4(coef + 4 * C + c), k5 = ld4(coef + 5 * C + c);
    const F4 mu = ld4(mean + c), rs = ld4(rstd + c);
    const F4 va = ld4(ga1 + i), v1 = ld4(gz1 + i), v2 = ld4(gz2 + i), xa = ld4(a + i);
    F4 ta = ld4(t + i);
    for (int sp = 1; sp < t_splits; sp += 8) {  // the tangent convolution's slabs: eight in flight, added in split order
      F4 sl[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) sl[u] = ld4(t + (long long)(sp + u < t_splits ? sp + u : 0) * t_slab + i);
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (sp + u < t_splits) {
#pragma unroll
          for (int e = 0; e < 4; ++e) ta.e[e] += sl[u].e[e];
        }
    }
    F4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float xh = (xa.e[e] - mu.e[e]) * rs.e[e];
      o.e[e] = ((k0.e[e] * va.e[e] + k1.e[e] * v1.e[e]) + (k2.e[e] * v2.e[e] + k3.e[e] * ta.e[e])) +
               (k4.e[e] * xh + k5.e[e]);
    }
    *reinterpret_cast<F4*>(out + i) = o;
  }
}

}  // namespace

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
  return nhwc && c % 4 == 0 && out_ld % 4 == 0 && add_ld % 4 == 0 && a_slab % 4 == 0 && 2 * total < 0x7fffffffLL;
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

int hf_bn_stats_rows(void* a_out, const void* a, int splits, int64_t slab_stride, void* part, int64_t rows, int64_t c,
                     int row_blocks, int dtype, void* stream) {
  if (!a || !part || splits < 1 || (splits > 1 && slab_stride <= 0) || rows <= 0 || c <= 0 || row_blocks < 1 ||
      dtype != HF_F32)
    return HF_ERR_ARG;
  if (!(c % 4 == 0 && c / 4 <= BLOCK) || rows * c > 0x7fffffffLL) return HF_ERR_ARG;
  if (!aligned16(a) || (a_out && !aligned16(a_out)) || (slab_stride & 3)) return HF_ERR_ALIGN;
  const unsigned rpb = (unsigned)((rows + row_blocks - 1) / row_blocks);
  hipLaunchKernelGGL(k_bn_stats_rows, dim3((unsigned)row_blocks), dim3(BLOCK), 0, (hipStream_t)stream,
                     (float*)a_out, (const float*)a, splits, (long long)slab_stride, (double*)part, (unsigned)rows,
                     (unsigned)c, rpb);
  HF_HIP(hipGetLastError());
  return HF_OK;
}

int hf_bn_train_hessian_coeffs(void* coef, void* gw_corr, const void* sum_gx2, const void* sum_g2, const void* sum_ga,
                               int nparts, const void* sum_tx, const void* sum_t1, int nparts_t, const void* g_gamma1,
                               const void* g_beta1, const void* gamma, const void* v_gamma, const void* rstd,
                               double count, int64_t c, int dtype, void* stream) {
  if (!coef || !gw_corr || !sum_gx2 || !sum_g2 || !sum_ga || !sum_tx || !sum_t1 || nparts < 1 || nparts_t < 1 || !g_gamma1 ||
      !g_beta1 || !gamma || !v_gamma || !rstd || !(count > 0.0) || c <= 0 || c > 0x7fffffffLL || dtype != HF_F32)
    return HF_ERR_ARG;
  hipLaunchKernelGGL(k_bn_train_hessian_coeffs, dim3((unsigned)((c + 7) / 8)), dim3(BLOCK), 0,
                     (hipStream_t)stream, (float*)coef, (float*)gw_corr, (const float*)sum_gx2, (const float*)sum_g2,
                     (const float*)sum_ga, (const float*)sum_tx, (const float*)sum_t1, nparts, nparts_t,
                     (const float*)g_gamma1,
                     (const float*)g_beta1, (const float*)gamma, (const float*)v_gamma, (const float*)rstd, count,
                     (int)c);
  HF_HIP(hipGetLastError());
  return HF_OK;
}

int hf_bn_train_hessian_apply(void* out, const void* ga1, const void* gz1, const void* gz2, const void* t, int t_splits,
                              int64_t t_slab, const void* a, const void* mean, const void* rstd, const void* coef,
                              int64_t rows, int64_t c, int dtype, void* stream) {
  if (!out || !ga1 || !gz1 || !gz2 || !t || t_splits < 1 || (t_splits > 1 && t_slab <= 0) || !a || !mean || !rstd ||
      !coef || rows <= 0 || c <= 0 || c % 4 || rows * c > 0x7fffffffLL || dtype != HF_F32)
    return HF_ERR_ARG;
  if (!aligned16(out) || !aligned16(ga1) || !aligned16(gz1) || !aligned16(gz2) || !aligned16(t) || !aligned16(a) ||
      !aligned16(mean) || !aligned16(rstd) || !aligned16(coef) || (t_slab & 3))
    return HF_ERR_ALIGN;
  const int64_t total4 = rows * c / 4;
  hipLaunchKernelGGL(k_bn_train_hessian_apply, dim3(wide_grid(total4)), dim3(BLOCK), 0, (hipStream_t)stream,
                     (float*)out, (const float*)ga1, (const float*)gz1, (const float*)gz2, (const float*)t, t_splits,
                     (long long)t_slab, (const float*)a, (const float*)mean, (const float*)rstd, (const float*)coef,
                     (unsigned)total4, (unsigned)c);
  HF_HIP(hipGetLastError());
  return HF_OK;
}
