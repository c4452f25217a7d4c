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

// The same for windows that overlap at most 2x2 (kernel <= 2*stride: the 3x3 / stride-2 pool of a
// ResNet stem) and at most MAXS slabs per cotangent, arranged for latency: ONE round of position
// loads (4 windows), then ONE round of slab loads for every matching window -- all of them issued
// before the first addition (the loop nest above waits window by window: 5 dependent round
// trips, 13 us for a 400 K-element map; this form: 2).  Same summation order.
constexpr int POOL_MAXS = 16;

__global__ __launch_bounds__(HB) void k_maxpool_adjoint_2x2(float* __restrict__ g, const float* __restrict__ A,
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
  int oy0 = y + q.ph - q.kh + 1; oy0 = oy0 > 0 ? (oy0 + q.sh - 1) / q.sh : 0;
  int oy1 = (y + q.ph) / q.sh;   oy1 = oy1 < q.OH - 1 ? oy1 : q.OH - 1;
  int ox0 = x + q.pw - q.kw + 1; ox0 = ox0 > 0 ? (ox0 + q.sw - 1) / q.sw : 0;
  int ox1 = (x + q.pw) / q.sw;   ox1 = ox1 < q.OW - 1 ? ox1 : q.OW - 1;
  size_t o[4];
  bool m[4];
  int id[4];
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const int oy = oy0 + (w >> 1), ox = ox0 + (w & 1);
    m[w] = oy <= oy1 && ox <= ox1;
    o[w] = m[w] ? ((size_t)(n * q.OH + oy) * q.OW + ox) * C + c : (size_t)c;  // (a valid address either way)
    id[w] = idx[o[w]];
  }
  float va[4][POOL_MAXS], vb[4][POOL_MAXS];
  const float* __restrict__ Bq = B ? B : A;   // (no second cotangent: its loads alias the first, unused)
  const int b_eff = B ? b_splits : 1;
  // all four match flags are COMPUTED before the first window is entered (the empty asm pins
  // them): a wait for a position inside the window blocks would have to cover the blocks' own
  // loads as well, the counter being in-order
  int mi[4];
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    mi[w] = (m[w] && id[w] == self) ? 1 : 0;
    asm volatile("" : "+v"(mi[w]));
  }
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    m[w] = mi[w] != 0;
    if (m[w]) {  // loads only: nothing below waits before all four windows have issued theirs
      // (slab index clamped instead of a branch per slab -- the surplus loads repeat the last slab
      // and are never added --: branches made hipcc wait between the loads)
#pragma unroll
      for (int s = 0; s < POOL_MAXS; ++s) {
        va[w][s] = A[(size_t)(s < a_splits ? s : a_splits - 1) * a_slab + o[w]];
        vb[w][s] = Bq[(size_t)(s < b_eff ? s : b_eff - 1) * b_slab + o[w]];
      }
    }
  }
  float acc = 0.f;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    if (!m[w]) continue;
    float v = va[w][0];
#pragma unroll
    for (int s = 1; s < POOL_MAXS; ++s)
      if (s < a_splits) v += va[w][s];
    if (B) {
      float u = vb[w][0];
#pragma unroll
      for (int s = 1; s < POOL_MAXS; ++s)
        if (s < b_splits) u += vb[w][s];
      v = v + u;
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
// One round of global loads: W, V_W and feat go to LDS, every wave keeps the feature tangents of
// its rows in registers; everything after the barrier reads LDS (a single workgroup that went
// back to L2 for every class was measured at 26 us: twelve dependent round trips).  A wave
// owns rows b, b+16, ...; the cross-row sums (g_W, g_b) run after a second barrier, rows in
// order.  Sized for small heads (K <= 64, F <= 512, everything within 150 KB of LDS).
// ---------------------------------------------------------------------------------------
constexpr int HEAD_T = 1024, HEAD_W = HEAD_T / 64, HEAD_ROWS = 4;  // rows per wave kept in registers

template <int CH>  // float4 chunks per lane: F <= 256*CH
__global__ __launch_bounds__(HEAD_T) void k_linear_ce_head(
    float* __restrict__ g_feat, float* __restrict__ g_w, float* __restrict__ g_b,
    const float* __restrict__ t_feat, const float* __restrict__ feat, const float* __restrict__ W,
    const float* __restrict__ VW, const float* __restrict__ vb, const float* __restrict__ p, float scale,
    int B, int F, int K) {
  extern __shared__ float lds[];
  const int F4 = F >> 2;
  float4* s_w = reinterpret_cast<float4*>(lds);   // [K][F]
  float4* s_vw = s_w + K * F4;                    // [K][F]
  float4* s_feat = s_vw + K * F4;                 // [B][F]
  float* s_h = reinterpret_cast<float*>(s_feat + B * F4);  // [B][K]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // ---- the only round of global loads ----
  for (int e = threadIdx.x; e < K * F4; e += HEAD_T) {
    s_w[e] = reinterpret_cast<const float4*>(W)[e];
    s_vw[e] = reinterpret_cast<const float4*>(VW)[e];
  }
  for (int e = threadIdx.x; e < B * F4; e += HEAD_T) s_feat[e] = reinterpret_cast<const float4*>(feat)[e];
  float4 tf[HEAD_ROWS][CH];
  float pk[HEAD_ROWS];
#pragma unroll
  for (int q = 0; q < HEAD_ROWS; ++q) {
    const int b = wave + HEAD_W * q;
    pk[q] = (b < B && lane < K) ? p[(size_t)b * K + lane] : 0.f;
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      const int j = lane + 64 * u;
      tf[q][u] = (b < B && j < F4) ? reinterpret_cast<const float4*>(t_feat + (size_t)b * F)[j]
                                   : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  const float bias = (vb && lane < K) ? vb[lane] : 0.f;
  __syncthreads();
  // ---- per row: logits' tangent, loss Hessian, data gradient (LDS only) ----
#pragma unroll
  for (int q = 0; q < HEAD_ROWS; ++q) {
    const int b = wave + HEAD_W * q;
    if (b >= B) break;
    float4 ff[CH];
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      const int j = lane + 64 * u;
      ff[u] = j < F4 ? s_feat[b * F4 + j] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float jv = 0.f;  // lane k keeps Jv[b][k]
    for (int k = 0; k < K; ++k) {
      float part = 0.f;
#pragma unroll
      for (int u = 0; u < CH; ++u) {
        const int j = lane + 64 * u;
        if (j < F4) {
          const float4 w = s_w[k * F4 + j], v = s_vw[k * F4 + j];
          part += tf[q][u].x * w.x + tf[q][u].y * w.y + tf[q][u].z * w.z + tf[q][u].w * w.w;
          part += ff[u].x * v.x + ff[u].y * v.y + ff[u].z * v.z + ff[u].w * v.w;
        }
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
      if (lane == k) jv = part + bias;
    }
    double d = (double)pk[q] * (double)jv;  // <p, Jv> in fp64, as hf_softmax_ce_hvp
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) d += __shfl_xor(d, off, 64);
    const float h = scale * (pk[q] * (jv - (float)d));
    if (lane < K) s_h[b * K + lane] = h;
    float4 acc[CH];
#pragma unroll
    for (int u = 0; u < CH; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < K; ++k) {
      const float hk = __shfl(h, k, 64);
#pragma unroll
      for (int u = 0; u < CH; ++u) {
        const int j = lane + 64 * u;
        if (j < F4) {
          const float4 w = s_w[k * F4 + j];
          acc[u].x += hk * w.x; acc[u].y += hk * w.y; acc[u].z += hk * w.z; acc[u].w += hk * w.w;
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
  // ---- weight / bias gradient: sums over the rows, in row order ----
  for (int e = threadIdx.x; e < K * F4; e += HEAD_T) {
    const int k = e / F4, j = e % F4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int b = 0; b < B; ++b) {
      const float hb = s_h[b * K + k];
      const float4 f = s_feat[b * F4 + j];
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
  const bool small = kh <= 2 * stride_h && kw <= 2 * stride_w && a_splits <= POOL_MAXS && b_splits <= POOL_MAXS;
  if (small)
    hipLaunchKernelGGL(k_maxpool_adjoint_2x2, dim3((unsigned)((total + HB - 1) / HB)), dim3(HB), 0,
                       (hipStream_t)stream, (float*)g, (const float*)gy_a, a_splits, (long long)a_slab,
                       (const float*)gy_b, b_splits, (long long)b_slab, (const int*)idx, (unsigned)total,
                       (unsigned)c, q);
  else
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
  if (rows < 1 || rows > HEAD_W * HEAD_ROWS || classes < 1 || classes > 64 || features < 4 || features % 4 ||
      features > 512)
    return -1;
  const size_t lds = (size_t)(2 * classes * features + rows * features + rows * classes) * sizeof(float);
  if (lds > 150 * 1024) return -1;  // W, V_W, feat and HJv must fit the CU's LDS
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
