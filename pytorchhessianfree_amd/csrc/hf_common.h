// hf_common.h -- what the translation units of libhfpcg.so share: workgroup size, 16-byte vector types, the
// fixed-order block reduction, error / alignment / grid helpers.  gfx950 (wave64) only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "hf_pcg.h"
#include "hf_unpack.h"

#define HF_HIP(expr)                         \
  do {                                       \
    hipError_t e_ = (expr);                  \
    if (e_ != hipSuccess) return (int)e_;    \
  } while (0)

namespace {

constexpr int BLOCK = 256;          // 4 waves of 64
constexpr int WAVES = BLOCK / 64;

using hf_shared::VecOf;
using hf_shared::VU;

inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// one element per thread: these activation-sized kernels (<= a few hundred thousand
// elements) are latency-bound, every extra grid-stride iteration adds a full round trip
inline int wide_grid(int64_t n) {
  int64_t g = (n + BLOCK - 1) / BLOCK;
  if (g < 1) g = 1;
  if (g > 16384) g = 16384;
  return (int)g;
}

inline int small_grid(int64_t n) {
  int64_t g = (n + BLOCK * 4 - 1) / (BLOCK * 4);
  if (g < 1) g = 1;
  if (g > 2048) g = 2048;
  return (int)g;
}

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

}  // namespace
