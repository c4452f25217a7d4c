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
      // slabs in batches of eight, all in flight before the first addition (one at a time is a dependent round
      // trip per slab: the first block's data gradient arrives as ~15 of them); same order of additions
      float v = A[o];
      float w = B ? B[o] : 0.f;
      for (int s = 1; s < a_splits; s += 8) {
        float t8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t8[u] = A[(size_t)(s + u < a_splits ? s + u : 0) * a_slab + o];
#pragma unroll
        for (int u = 0; u < 8; ++u) v += s + u < a_splits ? t8[u] : 0.f;
      }
      if (B) {
        for (int s = 1; s < b_splits; s += 8) {
          float t8[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) t8[u] = B[(size_t)(s + u < b_splits ? s + u : 0) * b_slab + o];
#pragma unroll
          for (int u = 0; u < 8; ++u) w += s + u < b_splits ? t8[u] : 0.f;
        }
        v = v + w;
      }
      acc += v;
    }
  g[i] = acc;
}

// max-pool, forward with positions (ATen's rule: first maximum in scan order, NaN wins)
__global__ __launch_bounds__(HB) void k_maxpool_forward(float* __restrict__ out, float* __restrict__ out2,
                                                        unsigned out2_ld, int* __restrict__ idx,
                                                        const float* __restrict__ x, unsigned total, unsigned C,
                                                        PoolGeo q) {
  const unsigned i = blockIdx.x * HB + threadIdx.x;
  if (i >= total) return;
  const unsigned c = i % C;
  unsigned pix = i / C;
  const unsigned opix = pix;
  const int ox = pix % q.OW; pix /= q.OW;
  const int oy = pix % q.OH;
  const int n = pix / q.OH;
  int y0 = oy * q.sh - q.ph, x0 = ox * q.sw - q.pw;
  const int y1 = y0 + q.kh < q.H ? y0 + q.kh : q.H, x1 = x0 + q.kw < q.W ? x0 + q.kw : q.W;
  y0 = y0 > 0 ? y0 : 0; x0 = x0 > 0 ? x0 : 0;
  float best = -__builtin_inff();
  int bi = y0 * q.W + x0;
  for (int yy = y0; yy < y1; ++yy)
    for (int xx = x0; xx < x1; ++xx) {
      const float v = x[((size_t)(n * q.H + yy) * q.W + xx) * C + c];
      if (v > best || v != v) { best = v; bi = yy * q.W + xx; }
    }
  if (out) out[i] = best;
  if (out2) out2[(size_t)opix * out2_ld + c] = best;
  idx[i] = bi;
}

// ---------------------------------------------------------------------------------------
// Classifier head of the GGN product in one launch:
//   Jv   = t_feat W^T + feat V_W^T + v_b          tangent of the logits        [B, K]
//   HJv  = scale * p * (Jv - <p, Jv>)             softmax-CE Hessian, row-wise [B, K]
//   g_feat = HJv W    [B, F],   g_W = HJv^T feat  [K, F],   g_b = sum_b HJv    [K]
// One workgroup per HEAD_R rows, one wave per row.  ONE round of global loads: W and V_W go to
// LDS, every wave keeps its row's features and feature tangents in registers; everything after
// the barrier reads LDS.  The sums over the rows (g_W, g_b) are left to the consumer: workgroup
// g writes the partial sums of ITS rows to slab g, hf_pack_ex adds the slabs up in order (as
// it does for split-K weight gradients).  Measured on the way here: a single workgroup that went
// back to L2 for every class 26 us (twelve dependent round trips); a single workgroup with
// everything in LDS 19-20 us (2.7 MB of LDS reads through ONE CU's 128 B/clk); this form spreads
// them over ceil(B / HEAD_R) CUs.
// ---------------------------------------------------------------------------------------
constexpr int HEAD_R = 4, HEAD_T = 64 * HEAD_R, KB = 5;

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
  float4* s_feat = s_vw + K * F4;                 // [HEAD_R][F]
  float* s_h = reinterpret_cast<float*>(s_feat + HEAD_R * F4);  // [HEAD_R][K]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x * HEAD_R + wave;
  const bool row = b < B;
  // ---- the only round of global loads ----
  for (int e = threadIdx.x; e < K * F4; e += HEAD_T) {
    s_w[e] = reinterpret_cast<const float4*>(W)[e];
    s_vw[e] = reinterpret_cast<const float4*>(VW)[e];
  }
  float4 tf[CH], ff[CH];
#pragma unroll
  for (int u = 0; u < CH; ++u) {
    const int j = lane + 64 * u;
    tf[u] = ff[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row && j < F4) {
      tf[u] = reinterpret_cast<const float4*>(t_feat + (size_t)b * F)[j];
      ff[u] = reinterpret_cast<const float4*>(feat + (size_t)b * F)[j];
    }
    if (j < F4) s_feat[wave * F4 + j] = ff[u];  // (zeros for a row past the end)
  }
  const float pk = (row && lane < K) ? p[(size_t)b * K + lane] : 0.f;
  const float bias = (vb && lane < K) ? vb[lane] : 0.f;
  __syncthreads();
  // ---- the row: logits' tangent, KB classes per pass (their wave reductions interleave) ----
  float jv = 0.f;  // lane k keeps Jv[b][k]
  for (int k0 = 0; k0 < K; k0 += KB) {
    float part[KB];
#pragma unroll
    for (int r = 0; r < KB; ++r) {
      const int k = k0 + r < K ? k0 + r : K - 1;  // (clamped: the surplus sums are discarded)
      float p0 = 0.f, p1 = 0.f;
#pragma unroll
      for (int u = 0; u < CH; ++u) {
        const int j = lane + 64 * u;
        if (j < F4) {
          const float4 w = s_w[k * F4 + j], v = s_vw[k * F4 + j];
          // (explicit fma: the file is built with -ffp-contract=off)
          p0 = fmaf(tf[u].x, w.x, p0); p1 = fmaf(tf[u].y, w.y, p1);
          p0 = fmaf(tf[u].z, w.z, p0); p1 = fmaf(tf[u].w, w.w, p1);
          p0 = fmaf(ff[u].x, v.x, p0); p1 = fmaf(ff[u].y, v.y, p1);
          p0 = fmaf(ff[u].z, v.z, p0); p1 = fmaf(ff[u].w, v.w, p1);
        }
      }
      part[r] = p0 + p1;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
      for (int r = 0; r < KB; ++r) part[r] += __shfl_xor(part[r], off, 64);
    }
#pragma unroll
    for (int r = 0; r < KB; ++r)
      if (lane == k0 + r && k0 + r < K) jv = part[r] + bias;
  }
  double d = (double)pk * (double)jv;  // <p, Jv> in fp64, as hf_softmax_ce_hvp
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) d += __shfl_xor(d, off, 64);
  const float h = row ? scale * (pk * (jv - (float)d)) : 0.f;
  if (lane < K) s_h[wave * K + lane] = h;
  // ---- the row's data gradient ----
  float4 acc[CH];
#pragma unroll
  for (int u = 0; u < CH; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int k0 = 0; k0 < K; k0 += KB) {
#pragma unroll
    for (int r = 0; r < KB; ++r) {
      const int k = k0 + r < K ? k0 + r : K - 1;
      const float hk = k0 + r < K ? __shfl(h, k, 64) : 0.f;
#pragma unroll
      for (int u = 0; u < CH; ++u) {
        const int j = lane + 64 * u;
        if (j < F4) {
          const float4 w = s_w[k * F4 + j];
          acc[u].x = fmaf(hk, w.x, acc[u].x); acc[u].y = fmaf(hk, w.y, acc[u].y);
          acc[u].z = fmaf(hk, w.z, acc[u].z); acc[u].w = fmaf(hk, w.w, acc[u].w);
        }
      }
    }
  }
  if (row) {
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      const int j = lane + 64 * u;
      if (j < F4) reinterpret_cast<float4*>(g_feat + (size_t)b * F)[j] = acc[u];
    }
  }
  __syncthreads();
  // ---- this workgroup's share of the weight / bias gradient (its rows, in order) -> slab ----
  float* gw = g_w + (size_t)blockIdx.x * K * F;
  for (int e = threadIdx.x; e < K * F4; e += HEAD_T) {
    const int k = e / F4, j = e % F4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int r = 0; r < HEAD_R; ++r) {
      const float hb = s_h[r * K + k];
      const float4 f = s_feat[r * F4 + j];
      s.x = fmaf(hb, f.x, s.x); s.y = fmaf(hb, f.y, s.y); s.z = fmaf(hb, f.z, s.z); s.w = fmaf(hb, f.w, s.w);
    }
    reinterpret_cast<float4*>(gw)[e] = s;
  }
  if (g_b && threadIdx.x < K) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < HEAD_R; ++r) s += s_h[r * K + threadIdx.x];
    g_b[(size_t)blockIdx.x * K + threadIdx.x] = s;
  }
}

// ---------------------------------------------------------------------------------------
// Head of a network ending in global average pooling, inside J^T H_L J v (one workgroup per sample):
//   Jv[k] = mean_hw t[hw][k];  h = scale * p * (Jv - <p, Jv>);  g[hw][k] = h[k] / hw
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(HB) void k_pool_ce_head(float* __restrict__ g, float* __restrict__ jv_out,
                                                     const float* __restrict__ t, const float* __restrict__ p,
                                                     float scale, int HW, int K) {
  __shared__ double red[HB / 64];
  __shared__ float s_h[1024];
  const int n = blockIdx.x;
  const float* tn = t + (size_t)n * HW * K;
  double part = 0.0;
  float jv_mine[4];  // K <= 1024: up to 4 classes per thread
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int k = threadIdx.x + u * HB;
    float s = 0.f;
    if (k < K) {
      for (int q = 0; q < HW; ++q) s += tn[(size_t)q * K + k];
      s = s / (float)HW;
      if (jv_out) jv_out[(size_t)n * K + k] = s;
      part += (double)p[(size_t)n * K + k] * (double)s;
    }
    jv_mine[u] = s;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
  __syncthreads();
  double d = red[0];
#pragma unroll
  for (int w = 1; w < HB / 64; ++w) d += red[w];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int k = threadIdx.x + u * HB;
    if (k < K) s_h[k] = (scale * (p[(size_t)n * K + k] * (jv_mine[u] - (float)d))) / (float)HW;
  }
  __syncthreads();
  float* gn = g + (size_t)n * HW * K;
  for (int e = threadIdx.x; e < HW * K; e += HB) gn[e] = s_h[e % K];
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

int hf_maxpool_forward_nhwc(void* out, void* out2, int64_t out2_ld, void* idx, const void* x, int64_t n,
                            int64_t h, int64_t w, int64_t oh, int64_t ow, int64_t c, int64_t kh, int64_t kw,
                            int64_t stride_h, int64_t stride_w, int64_t pad_h, int64_t pad_w, int dtype,
                            void* stream) {
  if (dtype != HF_F32 || !idx || !x || (!out && !out2) || n < 1 || c < 1 || oh < 1 || ow < 1) return -1;
  if (kh < 1 || kw < 1 || stride_h < 1 || stride_w < 1 || pad_h < 0 || pad_w < 0) return -1;
  const int64_t total = n * oh * ow * c;
  if (total >= (1LL << 31) || n * h * w * c >= (1LL << 31)) return -1;
  if (out2 && out2_ld < c) return -1;
  if (n * oh * ow * (out2 ? out2_ld : c) >= (1LL << 31)) return -1;
  PoolGeo q{(int)h, (int)w, (int)oh, (int)ow, (int)kh, (int)kw, (int)stride_h, (int)stride_w, (int)pad_h, (int)pad_w};
  hipLaunchKernelGGL(k_maxpool_forward, dim3((unsigned)((total + HB - 1) / HB)), dim3(HB), 0,
                     (hipStream_t)stream, (float*)out, (float*)out2, (unsigned)out2_ld, (int*)idx,
                     (const float*)x, (unsigned)total, (unsigned)c, q);
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
  // (a variant that issued the slab loads of all matching windows before the first addition --
  // two dependent round trips instead of five -- was measured slower, 17.7 vs 13.4 us: 138 VGPRs
  // and 32 loads per match instead of 11)
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
  if (rows < 1 || rows > 4096 || classes < 1 || classes > 64 || features < 4 || features % 4 || features > 512)
    return -1;
  const size_t lds = (size_t)(2 * classes * features + HEAD_R * features + HEAD_R * classes) * sizeof(float);
  if (lds > 64 * 1024) return -1;  // W, V_W and the workgroup's rows must fit the default LDS allowance
  if (!al16(g_feat) || !al16(g_w) || !al16(t_feat) || !al16(feat) || !al16(w) || !al16(v_w)) return -1;
  const int ch = (int)((features / 4 + 63) / 64);
  const unsigned groups = (unsigned)((rows + HEAD_R - 1) / HEAD_R);
  hipStream_t s = (hipStream_t)stream;
#define HF_HEAD(CH)                                                                                        \
  hipLaunchKernelGGL((k_linear_ce_head<CH>), dim3(groups), dim3(HEAD_T), lds, s, (float*)g_feat,           \
                     (float*)g_w, (float*)g_b, (const float*)t_feat, (const float*)feat, (const float*)w,  \
                     (const float*)v_w, (const float*)v_b, (const float*)p, (float)scale, (int)rows,       \
                     (int)features, (int)classes)
  if (ch <= 1) HF_HEAD(1);
  else HF_HEAD(2);
#undef HF_HEAD
  return (int)hipGetLastError();
}

int hf_pool_ce_head(void* g, void* jv_out, const void* t, const void* p, double scale, int64_t n, int64_t hw,
                    int64_t k, int dtype, void* stream) {
  if (dtype != HF_F32 || !g || !t || !p || n < 1 || hw < 1 || k < 1 || k > 1024) return -1;
  if (n * hw * k >= (1LL << 31)) return -1;
  hipLaunchKernelGGL(k_pool_ce_head, dim3((unsigned)n), dim3(HB), 0, (hipStream_t)stream, (float*)g,
                     (float*)jv_out, (const float*)t, (const float*)p, (float)scale, (int)hw, (int)k);
  return (int)hipGetLastError();
}

int hf_linear_ce_head_slabs(int64_t rows) { return rows < 1 ? 0 : (int)((rows + HEAD_R - 1) / HEAD_R); }

}  // extern "C"
