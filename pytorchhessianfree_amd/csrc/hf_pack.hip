// hf_pack.hip -- gather / scatter between the flat parameter-space vector and per-layer buffers (gfx950):
//   k_pack            hf_pack / hf_pack_ex      all parameter gradients -> flat vector (sums split-K slabs and
//                                               partial rows while it gathers; un-permutes (O,H,W,I) -> (O,I,H,W))
//   k_unpack_tangent  hf_unpack_tangent* / hf_unpack_weights   flat vector -> the [W | v_W] operands, (I,H,W,O) copies
//   k_live_copy       hf_live_copy              the entries that can be non-zero <-> compact staging vector
// Reference: parameters_to_vector / vector_to_parameter_list, hessianfree/utils.py:8-76, optimizer.py:462.
#include "hf_common.h"

#ifndef HF_PACK_DEEP
#define HF_PACK_DEEP 1  // (A/B knob of round 5: 0 builds the gather with 8-deep slab batches and >= 2048-element chunks)
#endif

namespace {

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
        int sp = 1;
        // (many slabs -- the weight gradients of large-map layers arrive as up to 128 --: sixteen in flight per lane;
        // same split order, same bits)
        for (; HF_PACK_DEEP && sp + 16 <= nsp; sp += 16) {
          VU<T> tt[16];
#pragma unroll
          for (int u = 0; u < 16; ++u) {
#pragma unroll
            for (int c = 0; c < W; ++c) tt[u].e[c] = (T)0;
            if (rd) tt[u].v = *reinterpret_cast<const V*>(src + e + (long long)(sp + u) * sps);
          }
#pragma unroll
          for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int c = 0; c < W; ++c) acc.e[c] += tt[u].e[c];
        }
        for (; sp < nsp; sp += 8) {
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

}  // namespace

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
          else if (a.nsplit[k] > 1 && sizeof(T) == 4 && I % 4 == 0 && I * HW <= (int64_t)(TILE_BYTES / sizeof(T))) {
            // LDS-staged stores, whole [I, HW] slabs per workgroup: >= 2048 elements -- >= 512 where the source is
            // MANY split-K slabs (a workgroup's time is then the chain of its slab batches: more, shorter workgroups;
            // All-CNN-C's 96 x 96 x 9 weight gradient arrives as 85 slabs and used to be gathered by 32 workgroups)
            const int64_t least = (HF_PACK_DEEP && a.nsplit[k] >= 24) ? 512 : 2048;
            a.chunk[k] = (int)(((least + I * HW - 1) / (I * HW)) * I * HW);
          }
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

template <typename T>
static int unpack_impl(const void* src, void* const* dsts, const int64_t* src_offs,
                       const int64_t* numels, const int64_t* slabs, const int64_t* inners,
                       const int64_t* live, const int64_t* halves, int nt, hipStream_t s) {
  int t = 0;
  while (t < nt) {
    UnpackArgs a;
    int blocks = 0;
    t = hf_shared::fill_unpack_args<T>(a, &blocks, t, dsts, src_offs, numels, slabs, inners, live, halves, nt, true);
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

