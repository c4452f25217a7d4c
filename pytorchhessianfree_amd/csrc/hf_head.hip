// hf_head.hip -- the small non-convolution stages of the curvature engine's sweeps (gfx950):
// the max-pool behind a ResNet stem and the classifier head (linear layer + softmax
// cross-entropy Hessian).  Each of these was 2-6 ATen / rocBLAS launches of ~5 us inside
// every GGN product (BackPACK's R-op / L-op through them, optimizer.py:461); here each is ONE
// launch, with fixed summation orders (no atomics: the product stays bitwise repeatable).
//
// All tensors fp32, activations NHWC.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hf_pcg.h"

namespace {

constexpr int HB = 256;

// ---------------------------------------------------------------------------------------
// max-pool, tangent:   out[n,oy,ox,c] = t[n, idx[n,oy,ox,c], c]
// idx = flat position (y*W + x) of the window's maximum in the forward pass (what
// max_pool2d_with_indices returns), stored NHWC int32.  `out` may be the first-channels slice
// of a wider NHWC buffer (out_ld floats per pixel).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(HB) void k_maxpool_tangent(float* __restrict__ out, const float* __restrict__ t,
                                                        const int* __restrict__ idx, unsigned total,
                                                        unsigned C, unsigned out_ld, unsigned opix, unsigned ipix) {
  const unsigned i = blockIdx.x * HB + threadIdx.x;
  if (i >= total) return;
  const unsigned c = i % C, pix = i / C, n = pix / opix;
  out[(size_t)pix * out_ld + c] = t[((size_t)n * ipix + (unsigned)idx[i]) * C + c];
}

// ---------------------------------------------------------------------------------------
// max-pool, adjoint, in gather form (no zero-fill, no atomics):
//   g[n,y,x,c] = sum over the windows (oy,ox) that contain (y,x) and whose maximum sits at (y,x)
//                of  (sum_s A[s][n,oy,ox,c] + sum_s B[s][n,oy,ox,c])
// A, B: the cotangents of the pooled map from its two consumers (the first residual block's
// convolution and identity branch), each possibly split-K slabs that are summed here in split
// order.  Windows are visited in (oy, ox) order.
// ---------------------------------------------------------------------------------------
struct PoolGeo {
  int H, W, OH, OW, kh, kw, sh, sw, ph, pw;
};

__global__ __launch_bounds__(HB) void k_maxpool_adjoint(float* __restrict__ g, const float* __restrict__ A,
                                                        int a_splits, long long a_slab,
                                                        const float* __restrict__ B, int b_splits,
                                                        long long b_slab, const int* __restrict__ idx,
                                                        unsigned total, unsigned C, PoolGeo q) {
  const unsigned i = blockIdx.x * HB + threadIdx.x;
  if (i >= total) return;
  const unsigned c = i % C;
  unsigned pix = i / C;
  const int x = pix % q.W; pix /= q.W;
  const int y = pix % q.H;
  const int n = pix / q.H;
  const int self = y * q.W + x;
  // windows containing (y, x): oy*sh - ph <= y <= oy*sh - ph + kh - 1
  int oy0 = y + q.ph - q.kh + 1; oy0 = oy0 > 0 ? (oy0 + q.sh - 1) / q.sh : 0;
  int oy1 = (y + q.ph) / q.sh;   oy1 = oy1 < q.OH - 1 ? oy1 : q.OH - 1;
  int ox0 = x + q.pw - q.kw + 1; ox0 = ox0 > 0 ? (ox0 + q.sw - 1) / q.sw : 0;
  int ox1 = (x + q.pw) / q.sw;   ox1 = ox1 < q.OW - 1 ? ox1 : q.OW - 1;
  float acc = 0.f;
  for (int oy = oy0; oy <= oy1; ++oy)
    for (int ox = ox0; ox <= ox1; ++ox) {
      const size_t o = ((size_t)(n * q.OH + oy) * q.OW + ox) * C + c;
      if (idx[o] != self) continue;
      float v = A[o];
      for (int s = 1; s < a_splits; ++s) v += A[(size_t)s * a_slab + o];
      if (B) {
        float w = B[o];
        for (int s = 1; s < b_splits; ++s) w += B[(size_t)s * b_slab + o];
        v = v + w;
      }
      acc += v;
    }
  g[i] = acc;
}

// ---------------------------------------------------------------------------------------
// Classifier head of the GGN product in one launch (one workgroup, 16 waves):
//   Jv   = t_feat W^T + feat V_W^T + v_b          tangent of the logits        [B, K]
//   HJv  = scale * p * (Jv - <p, Jv>)             softmax-CE Hessian, row-wise [B, K]
//   g_feat = HJv W    [B, F],   g_W = HJv^T feat  [K, F],   g_b = sum_b HJv    [K]
// A wave owns a row b: the row's two feature vectors stay in registers while it walks the K
// classes; the cross-row sums (g_W, g_b) run after a barrier on HJv / feat in LDS, rows in
// order.  Sized for small heads (K <= 64, F <= 512: 120 VGPRs at 16 waves; feat + HJv fit LDS).
// ---------------------------------------------------------------------------------------
constexpr int HEAD_T = 1024, HEAD_W = HEAD_T / 64, KB = 5;

template <int CH>  // float4 chunks per lane: F <= 256*CH
__global__ __launch_bounds__(HEAD_T) void k_linear_ce_head(
    float* __restrict__ g_feat, float* __restrict__ g_w, float* __restrict__ g_b,
    const float* __restrict__ t_feat, const float* __restrict__ feat, const float* __restrict__ W,
    const float* __restrict__ VW, const float* __restrict__ vb, const float* __restrict__ p, float scale,
    int B, int F, int K) {
  extern __shared__ float lds[];
  float* s_h = lds;                 // [B][K]
  float* s_feat = lds + ((B * K + 3) & ~3);  // [B][F], 16-byte aligned
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int F4 = F >> 2;
  for (int b = wave; b < B; b += HEAD_W) {
    float4 tf[CH], ff[CH];
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      const int j = lane + 64 * u;
      tf[u] = ff[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (j < F4) {
        tf[u] = reinterpret_cast<const float4*>(t_feat + (size_t)b * F)[j];
        ff[u] = reinterpret_cast<const float4*>(feat + (size_t)b * F)[j];
        reinterpret_cast<float4*>(s_feat + (size_t)b * F)[j] = ff[u];
      }
    }
    // logits' tangent, KB classes per pass: all loads of a pass are issued before the first use
    // (clamped row index instead of a branch, so that nothing serialises them)
    float jv = 0.f;  // lane k keeps Jv[b][k] (K <= 64)
    for (int k0 = 0; k0 < K; k0 += KB) {
      float4 w[KB][CH], v[KB][CH];
#pragma unroll
      for (int q = 0; q < KB; ++q) {
        const int k = k0 + q < K ? k0 + q : K - 1;
#pragma unroll
        for (int u = 0; u < CH; ++u) {
          const int j = lane + 64 * u < F4 ? lane + 64 * u : 0;
          w[q][u] = reinterpret_cast<const float4*>(W + (size_t)k * F)[j];
          v[q][u] = reinterpret_cast<const float4*>(VW + (size_t)k * F)[j];
        }
      }
#pragma unroll
      for (int q = 0; q < KB; ++q) {
        float part = 0.f;
#pragma unroll
        for (int u = 0; u < CH; ++u) {
          if (lane + 64 * u < F4) {
            part += tf[u].x * w[q][u].x + tf[u].y * w[q][u].y + tf[u].z * w[q][u].z + tf[u].w * w[q][u].w;
            part += ff[u].x * v[q][u].x + ff[u].y * v[q][u].y + ff[u].z * v[q][u].z + ff[u].w * v[q][u].w;
          }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
        if (lane == k0 + q && k0 + q < K) jv = part + (vb ? vb[k0 + q] : 0.f);
      }
    }
    // softmax-CE Hessian on the row: lanes 0..K-1 hold Jv, the dot product in fp64
    const float pk = lane < K ? p[(size_t)b * K + lane] : 0.f;
    double d = (double)pk * (double)jv;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) d += __shfl_xor(d, off, 64);
    const float h = scale * (pk * (jv - (float)d));
    if (lane < K) s_h[b * K + lane] = h;
    // data gradient of the row: g_feat[b, :] = sum_k h_k W[k, :]
    float4 acc[CH];
#pragma unroll
    for (int u = 0; u < CH; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k0 = 0; k0 < K; k0 += KB) {
      float4 w[KB][CH];
#pragma unroll
      for (int q = 0; q < KB; ++q) {
        const int k = k0 + q < K ? k0 + q : K - 1;
#pragma unroll
        for (int u = 0; u < CH; ++u) {
          const int j = lane + 64 * u < F4 ? lane + 64 * u : 0;
          w[q][u] = reinterpret_cast<const float4*>(W + (size_t)k * F)[j];
        }
      }
#pragma unroll
      for (int q = 0; q < KB; ++q) {
        const float hk = k0 + q < K ? __shfl(h, k0 + q, 64) : 0.f;
#pragma unroll
        for (int u = 0; u < CH; ++u) {
          acc[u].x += hk * w[q][u].x; acc[u].y += hk * w[q][u].y;
          acc[u].z += hk * w[q][u].z; acc[u].w += hk * w[q][u].w;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      const int j = lane + 64 * u;
      if (j < F4) reinterpret_cast<float4*>(g_feat + (size_t)b * F)[j] = acc[u];
    }
  }
  __syncthreads();
  // weight / bias gradient: sums over the rows, in row order
  for (int e = threadIdx.x; e < K * F4; e += HEAD_T) {
    const int k = e / F4, j = e % F4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int b = 0; b < B; ++b) {
      const float hb = s_h[b * K + k];
      const float4 f = reinterpret_cast<const float4*>(s_feat + (size_t)b * F)[j];
      s.x += hb * f.x; s.y += hb * f.y; s.z += hb * f.z; s.w += hb * f.w;
    }
    reinterpret_cast<float4*>(g_w + (size_t)k * F)[j] = s;
  }
  if (g_b && threadIdx.x < K) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += s_h[b * K + threadIdx.x];
    g_b[threadIdx.x] = s;
  }
}

inline bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

}  // namespace

extern "C" {

int hf_maxpool_tangent_nhwc(void* out, const void* t, const void* idx, int64_t n, int64_t h, int64_t w,
                            int64_t oh, int64_t ow, int64_t c, int64_t out_ld, int dtype, void* stream) {
  if (dtype != HF_F32 || !out || !t || !idx || n < 1 || c < 1) return -1;
  const int64_t total = n * oh * ow * c;
  if (total >= (1LL << 31) || n * h * w * c >= (1LL << 31)) return -1;
  if (out_ld == 0) out_ld = c;
  if (out_ld < c) return -1;
  hipLaunchKernelGGL(k_maxpool_tangent, dim3((unsigned)((total + HB - 1) / HB)), dim3(HB), 0,
                     (hipStream_t)stream, (float*)out, (const float*)t, (const int*)idx, (unsigned)total,
                     (unsigned)c, (unsigned)out_ld, (unsigned)(oh * ow), (unsigned)(h * w));
  return (int)hipGetLastError();
}

int hf_maxpool_adjoint_nhwc(void* g, const void* gy_a, int a_splits, int64_t a_slab, const void* gy_b,
                            int b_splits, int64_t b_slab, const void* idx, int64_t n, int64_t h, int64_t w,
                            int64_t oh, int64_t ow, int64_t c, int64_t kh, int64_t kw, int64_t stride_h,
                            int64_t stride_w, int64_t pad_h, int64_t pad_w, int dtype, void* stream) {
  if (dtype != HF_F32 || !g || !gy_a || !idx || n < 1 || c < 1 || a_splits < 1 || (gy_b && b_splits < 1)) return -1;
  if (kh < 1 || kw < 1 || stride_h < 1 || stride_w < 1 || pad_h < 0 || pad_w < 0) return -1;
  const int64_t total = n * h * w * c;
  if (total >= (1LL << 31)) return -1;
  PoolGeo q{(int)h, (int)w, (int)oh, (int)ow, (int)kh, (int)kw, (int)stride_h, (int)stride_w, (int)pad_h, (int)pad_w};
  hipLaunchKernelGGL(k_maxpool_adjoint, dim3((unsigned)((total + HB - 1) / HB)), dim3(HB), 0,
                     (hipStream_t)stream, (float*)g, (const float*)gy_a, a_splits, (long long)a_slab,
                     (const float*)gy_b, b_splits, (long long)b_slab, (const int*)idx, (unsigned)total,
                     (unsigned)c, q);
  return (int)hipGetLastError();
}

int hf_linear_ce_head(void* g_feat, void* g_w, void* g_b, const void* t_feat, const void* feat, const void* w,
                      const void* v_w, const void* v_b, const void* p, double scale, int64_t rows,
                      int64_t features, int64_t classes, int dtype, void* stream) {
  if (dtype != HF_F32 || !g_feat || !g_w || !t_feat || !feat || !w || !v_w || !p) return -1;
  if (rows < 1 || classes < 1 || classes > 64 || features < 4 || features % 4 || features > 512) return -1;
  const size_t lds = (size_t)(((rows * classes + 3) & ~3LL) + rows * features) * sizeof(float);
  if (lds > 150 * 1024) return -1;  // feat + HJv must fit the CU's LDS
  if (!al16(g_feat) || !al16(g_w) || !al16(t_feat) || !al16(feat) || !al16(w) || !al16(v_w)) return -1;
  const int ch = (int)((features / 4 + 63) / 64);
  hipStream_t s = (hipStream_t)stream;
  // (the attribute is raised once per variant and never during a later stream capture: the
  // engine's first, eager product comes before any capture)
  static size_t lds_allowed[3] = {0, 0, 0};
#define HF_HEAD(CH)                                                                                         \
  do {                                                                                                      \
    if (lds > lds_allowed[CH]) {                                                                            \
      hipError_t e_ = hipFuncSetAttribute((const void*)k_linear_ce_head<CH>,                                \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);          \
      if (e_ != hipSuccess) return (int)e_;                                                                 \
      lds_allowed[CH] = 150 * 1024;                                                                         \
    }                                                                                                       \
    hipLaunchKernelGGL((k_linear_ce_head<CH>), dim3(1), dim3(HEAD_T), lds, s, (float*)g_feat, (float*)g_w,  \
                       (float*)g_b, (const float*)t_feat, (const float*)feat, (const float*)w,              \
                       (const float*)v_w, (const float*)v_b, (const float*)p, (float)scale, (int)rows,      \
                       (int)features, (int)classes);                                                        \
  } while (0)
  if (ch <= 1) HF_HEAD(1);
  else HF_HEAD(2);
#undef HF_HEAD
  return (int)hipGetLastError();
}

}  // extern "C"
