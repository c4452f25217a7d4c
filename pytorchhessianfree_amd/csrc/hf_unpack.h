// Shared between hf_pcg.hip (hf_unpack_weights) and hf_conv.hip (the stem's tangent convolution carries the
// scatter of all other layers' v_W as extra workgroups): the multi-tensor scatter's argument block, its device body
// and the host code that fills the block.  Test infrastructure has no part in this file.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "hf_pcg.h"

namespace hf_shared {

constexpr int BLOCK = 256;      // threads per workgroup (hf_pcg.hip BLOCK, hf_conv.hip CT)
constexpr int PACK_MAXT = 64;   // (the argument struct must stay below the 4 KB kernarg limit)
constexpr int PACK_CHUNK = BLOCK * 16;  // elements per block

template <typename T> struct VecOf;
template <> struct VecOf<float>  { typedef float4  type; static constexpr int W = 4; };
template <> struct VecOf<double> { typedef double2 type; static constexpr int W = 2; };

template <typename T> union VU {
  typename VecOf<T>::type v;
  T e[VecOf<T>::W];
};

// Multi-tensor scatter for the tangent sweep (inverse of the gather above): tensor t,
// the contiguous [O, slab] block at src + src_off[t], goes into the second half of the
// input-channel axis of a [O, 2I, H, W] buffer stored NCHW (inner = 0) or NHWC (inner = I).
struct UnpackArgs {
  void* dst[PACK_MAXT];
  long long src_off[PACK_MAXT];
  long long numel[PACK_MAXT];
  int blk_start[PACK_MAXT + 1];
  int slab[PACK_MAXT];   // I*H*W
  int inner[PACK_MAXT];  // 0 (NCHW) or I (NHWC)
  int chunk[PACK_MAXT];  // elements per block
  unsigned short live[PACK_MAXT];  // NHWC, HW <= 16: taps whose slices are copied (0 = all), see PackArgs
  unsigned char half[PACK_MAXT];   // 1: the v_W half of the [W | v_W] operand, 0: the W half
  int nt;
};

// One workgroup's share (workgroup `bid`, BLOCK threads); the kernel below and the convolution launch that carries
// the scatter as extra workgroups (hf_conv.hip) both run it.
template <typename T>
__device__ __forceinline__ void unpack_block(const T* __restrict__ src_base, const UnpackArgs& a, unsigned bid) {
  int lo = 0, hi = a.nt;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (a.blk_start[mid] <= (int)bid) lo = mid; else hi = mid;
  }
  const T* __restrict__ src = src_base + a.src_off[lo];
  T* __restrict__ dst = reinterpret_cast<T*>(a.dst[lo]);
  const long long numel = a.numel[lo];
  const long long j0 = (long long)(bid - a.blk_start[lo]) * a.chunk[lo];
  const long long j1 = (j0 + a.chunk[lo] < numel) ? j0 + a.chunk[lo] : numel;
  const unsigned slab = (unsigned)a.slab[lo], I = (unsigned)a.inner[lo];
  const unsigned half = a.half[lo];
  constexpr int W = VecOf<T>::W;
  typedef typename VecOf<T>::type V;
  const bool al = ((((uintptr_t)src) | ((uintptr_t)dst)) & 15) == 0;
  if (half == 2) return;  // (transposed copies: unpack_transposed_block, k_unpack_tangent only)
  if (I == 0) {  // dst[o*2*slab + half*slab + r] = src[o*slab + r]
    if (al && slab % W == 0) {
      for (long long j = j0 + (long long)threadIdx.x * W; j < j1; j += (long long)BLOCK * W) {
        const long long o = j / slab;
        *reinterpret_cast<V*>(dst + j + (o + half) * slab) = *reinterpret_cast<const V*>(src + j);
      }
    } else {
      for (long long j = j0 + threadIdx.x; j < j1; j += BLOCK) dst[j + (j / slab + half) * slab] = src[j];
    }
    return;
  }
  dst += half ? I : 0;  // column offset inside the 2I-wide rows
  // destination order d = (o*HW + hw)*I + i  ->  dst[(o*HW + hw)*2I + half*I + i] = src[(o*I + i)*HW + hw]
  const unsigned HW = slab / I;
  const unsigned live = a.live[lo] ? a.live[lo] : 0xffffffffu;
  if (al && I % W == 0 && numel < 0x7fffffffLL) {
    const unsigned j1u = (unsigned)j1;  // (32-bit index arithmetic: see k_pack)
    for (unsigned d = (unsigned)j0 + threadIdx.x * W; d < j1u; d += BLOCK * W) {
      const unsigned row = d / I;  // o*HW + hw
      const unsigned i = d - row * I;
      const unsigned o = row / HW;
      const unsigned hw = row - o * HW;
      if (!((live >> hw) & 1u)) continue;  // a tap that never meets data: its slice is never read
      const T* s = src + (size_t)o * slab + i * HW + hw;
      VU<T> v;
#pragma unroll
      for (int c = 0; c < W; ++c) v.e[c] = s[c * HW];
      *reinterpret_cast<V*>(dst + (size_t)row * 2 * I + i) = v.v;
    }
  } else {
    for (long long d = j0 + threadIdx.x; d < j1; d += BLOCK) {
      const long long row = d / I;
      const unsigned i = (unsigned)(d - row * I);
      const long long o = row / HW;
      const unsigned hw = (unsigned)(row - o * HW);
      if (!((live >> hw) & 1u)) continue;
      dst[row * 2 * I + i] = src[o * slab + (long long)i * HW + hw];
    }
  }
}


// Transposed copy (half == 2): dst stored (I, H, W, O) dense -- the operand of a data-gradient convolution --,
// dst[r*O + o] = src[o*R + r] with r = i*HW + hw, R = I*HW: a [O, R] -> [R, O] matrix transpose, one TT x TT tile per
// workgroup through LDS (coalesced 256-byte rows on both sides; the element-wise walk this replaces scattered 4-byte
// stores O floats apart: 157 us for the 44.7 MB of a ResNet-18 vector, once per Hessian product).
// Returns false when workgroup `bid` belongs to a tensor of another kind.  lds: TT * (TT + 1) elements.
constexpr int TT = 64;
static_assert(BLOCK % TT == 0 && BLOCK >= TT, "unpack_transposed_block walks BLOCK / TT rows of a TT x TT tile per pass");
template <typename T>
__device__ __forceinline__ bool unpack_transposed_block(const T* __restrict__ src_base, const UnpackArgs& a,
                                                        unsigned bid, T* lds) {
  int lo = 0, hi = a.nt;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (a.blk_start[mid] <= (int)bid) lo = mid; else hi = mid;
  }
  if (a.half[lo] != 2) return false;
  const T* __restrict__ src = src_base + a.src_off[lo];
  T* __restrict__ dst = reinterpret_cast<T*>(a.dst[lo]);
  const unsigned R = (unsigned)a.slab[lo], O = (unsigned)(a.numel[lo] / a.slab[lo]);
  const unsigned tiles_o = (O + TT - 1) / TT;
  const unsigned tile = bid - (unsigned)a.blk_start[lo];
  const unsigned o0 = (tile % tiles_o) * TT, r0 = (tile / tiles_o) * TT;
  const unsigned tx = threadIdx.x % TT, ty = threadIdx.x / TT;  // BLOCK / TT = 4 rows per pass
#pragma unroll 4
  for (unsigned k = 0; k < TT; k += BLOCK / TT) {
    const unsigned o = o0 + ty + k, r = r0 + tx;
    if (o < O && r < R) lds[(ty + k) * (TT + 1) + tx] = src[(size_t)o * R + r];
  }
  __syncthreads();
#pragma unroll 4
  for (unsigned k = 0; k < TT; k += BLOCK / TT) {
    const unsigned r = r0 + ty + k, o = o0 + tx;
    if (o < O && r < R) dst[(size_t)r * O + o] = lds[tx * (TT + 1) + ty + k];
  }
  return true;
}


// Fills ONE argument block for tensors t0 ... (at most PACK_MAXT non-empty ones); returns the next tensor index,
// the number of workgroups in *blocks, or a negative error code.
template <typename T>
inline int fill_unpack_args(UnpackArgs& a, int* blocks_out, int t, void* const* dsts, const int64_t* src_offs,
                            const int64_t* numels, const int64_t* slabs, const int64_t* inners, const int64_t* live,
                            const int64_t* halves, int nt, bool allow_transposed) {
  // allow_transposed: the caller's kernel runs unpack_transposed_block for half == 2 tensors (k_unpack_tangent does;
  // a launch that only runs unpack_block -- the convolution carrying the scatter -- must refuse them: unpack_block
  // returns without writing anything for such a tensor)
  memset(&a, 0, sizeof(a));
  int k = 0, blocks = 0;
  while (t < nt && k < PACK_MAXT) {
    if (numels[t] < 0) return HF_ERR_ARG;
    if (numels[t] > 0) {
      const int64_t slab = slabs[t], I = inners[t];
      if (!dsts[t] || src_offs[t] < 0 || slab <= 0 || slab > 0x3fffffffLL || numels[t] % slab != 0 ||
          I < 0 || (I > 0 && slab % I != 0))
        return HF_ERR_ARG;
      a.dst[k] = dsts[t];
      a.src_off[k] = src_offs[t];
      a.numel[k] = numels[t];
      a.slab[k] = (int)slab;
      a.inner[k] = (int)I;
      if (live && live[t] > 0 && I > 0 && slab / I <= 16)
        a.live[k] = (unsigned short)(live[t] & ((1 << (slab / I)) - 1));
      a.half[k] = (unsigned char)(!halves ? 1 : halves[t] == 0 ? 0 : halves[t] == 2 ? 2 : 1);
      if (a.half[k] == 2 && (I <= 0 || !allow_transposed)) return HF_ERR_ARG;
      // (LDS-staged NHWC variants measured slower twice: round 1 30.9 vs 24.5 us; round 3 -- contiguous 16-byte
      // reads into LDS, lane = channel on the way out -- 21.2 vs 14.3 us, scripts/experiments/unpack_time.py)
      a.chunk[k] = PACK_CHUNK;
      a.blk_start[k] = blocks;
      if (a.half[k] == 2)  // one TT x TT tile of the [O, slab] -> [slab, O] transpose per workgroup
        blocks += (int)(((numels[t] / slab + TT - 1) / TT) * ((slab + TT - 1) / TT));
      else
        blocks += (int)((numels[t] + a.chunk[k] - 1) / a.chunk[k]);
      ++k;
    }
    ++t;
  }
  a.blk_start[k] = blocks;
  a.nt = k;
  *blocks_out = blocks;
  return t;
}

}  // namespace hf_shared
