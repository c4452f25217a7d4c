// hf_conv.hip -- implicit-GEMM convolution kernels for the curvature product's conv
// layers on SMALL feature maps (gfx950, fp32 MFMA, NHWC), deterministic split-K.
//
// Why these exist.  A GGN product of a ResNet-18 on 28x28 inputs runs 48 convolutions
// (tangent, data-gradient and weight-gradient of 16 layers) whose GEMM view has a tiny
// M (N*OH*OW = 32..1568 rows) and a long K (9*Cin up to 4608).  The only way to fill
// 256 CUs is to split K.  MIOpen's kernels do that with a zero-fill launch in front
// and atomic accumulation: 48 extra launches per product (16 % of the GPU time) and a
// product that is not bitwise repeatable.  Here every K-split writes its partial tile
// to a workspace and the LAST workgroup to arrive at a tile (ticket counter) sums the
// partials in split order: one launch, no zero-fill, no float atomics, bitwise
// deterministic.  Taps of the kernel window that can never meet data (3x3 kernels on
// 1x1 / 2x2 maps) are dropped on the host, so such layers cost what a GEMM costs.
//
// Three directions, all "activation rows gathered on the fly" (implicit im2col):
//   F  out[m=(n,oh,ow)][k]  = sum_{tap,c} X[n, oh*s-p+r, ow*s-p+q][c] * Wt[k][tap][c]
//   D  dX [m=(n,ih,iw)][c]  = sum_{tap,k} dY[n,(ih+p-r)/s,(iw+p-q)/s][k] * WT[c][tap][k]
//   W  dW [k][tap][c]       = sum_{m}     dY[m][k] * X[pix(m,tap)][c]
// F and D are one kernel (operands contiguous along the reduction index: "NT"); W reduces
// over the rows ("TN").  v_mfma_f32_32x32x2_f32, 64x64 block tiles, 4 waves of 32x32,
// BK = 32, double-buffered LDS, register prefetch of the next tile.
//
// MFMA operand maps (cdna_hip_programming.md section 3): lane l supplies A[i=l&31][k=l>>5],
// B[k=l>>5][j=l&31]; C/D: col = l&31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5).
// The 32 k of a step are assigned as k = 16*(l>>5) + step (A and B alike), so that a lane's
// 16 operands are 64 contiguous bytes of an LDS row.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "hf_pcg.h"
#include "hf_unpack.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int CT = 256;            // threads per block
#ifndef HF_CONV_BK
#define HF_CONV_BK 32
#endif
constexpr int MAX_TAPS = 64;
#ifndef HF_CONV_FLAT96
#define HF_CONV_FLAT96 1  // (A/B knob of round 5: 0 builds without the 96 x 128 weight-gradient configuration)
#endif
#ifndef HF_DCLASS_NOSPLIT
#define HF_DCLASS_NOSPLIT 1
#endif
#ifndef HF_FLAT96_MIN_WORK
#define HF_FLAT96_MIN_WORK 2048  // (tiles x K-steps from which a weight gradient takes Flat96; A/B: 6144 = the Big rule)
#endif

// Tile configuration: 4 waves as WM x WN, each wave owns TM x TN MFMA tiles of 32x32 (block tile
// 32*WM*TM x 32*WN*TN), BK reduction elements per step.  Three configurations are built:
//   Small (2x2 waves, 1x1 tiles, BK 32): 64x64 -- the latency-bound small-map problems of the ResNet-18
//     product (a few K-steps per workgroup; BK = 64 measured the same);
//   Big   (2x2 waves, 2x2 tiles, BK 16): 128x128, four accumulators per wave -- problems with thousands
//     of GEMM rows (All-CNN-C, ResNet-50-size maps), where the 64x64 kernel is issue-bound: per K-step
//     it feeds 16 MFMAs per wave from 2+2 staged float4 per thread, this one 32 MFMAs from the same 2+2;
//   Big96 (4x1 waves, 1x3 tiles, BK 16): 128x96 -- the same for column counts that are multiples of 96
//     (All-CNN-C's 96 / 192 channels would waste a quarter of every 128-wide tile).  Staged like Big
//     (128 LDS rows per operand, the last 32 of B unused).
// All stage 2 + 2 float4 per thread and step, so the hand-counted prefetch ring below is shared.
template <int WM_, int WN_, int TM_, int TN_, int BK_>
struct Cfg {
  static constexpr int WM = WM_, WN = WN_, TM = TM_, TN = TN_, BK = BK_;
  static constexpr int BM = 32 * WM * TM, BN = 32 * WN * TN;
  static constexpr int SM = (BM + 63) / 64 * 64, SN = (BN + 63) / 64 * 64;  // staged rows / columns per operand
  static constexpr int KQ = BK / 4;          // float4 per row of a k-contiguous tile
  static constexpr int RPT = CT / KQ;        // NT: rows covered per pass of the staging threads
  static constexpr int HK = BK / 2;          // k per lane half
  static constexpr int LDK = BK + 4;         // NT: LDS row stride (floats): 16-B aligned, conflict-free b128 reads
  static constexpr int LDA = SM + 4, LDB = SN + 4;  // TN: LDS row strides
  static constexpr int NT_FLOATS = 2 * (SM + SN) * LDK;
  static constexpr int TN_FLOATS = 2 * BK * (LDA + LDB);
  static constexpr int LDS_FLOATS = (NT_FLOATS > TN_FLOATS ? NT_FLOATS : TN_FLOATS) + 4;  // + the last-arriver flag
  static_assert(WM * WN == 4, "four waves");
  static_assert(SM / RPT == 2 && SN / RPT == 2, "the hand-counted prefetch ring assumes 2 + 2 loads per step (NT)");
  static_assert(SM == SN, "both operands are staged at the same width");
  static_assert(LDS_FLOATS * 4 <= 64 * 1024, "static LDS limit");
};
typedef Cfg<2, 2, 1, 1, HF_CONV_BK> Small;
typedef Cfg<2, 2, 2, 2, 16> Big;
typedef Cfg<4, 1, 1, 3, 16> Big96;
// Flat96 (1x4 waves, 3x1 tiles, BK 16): 96 x 128 -- WEIGHT GRADIENTS whose dY channel count is a multiple of 96
// (All-CNN-C's 96 / 192): the 96 output rows fill the tile exactly, and the output COLUMNS are enumerated flat over
// (kernel tap, input channel) -- dW[k][tap][c] is one contiguous [k][taps * c] matrix when every tap is live -- so a
// 128-wide column tile may straddle taps (each staging thread keeps its own tap) and only the last of the
// ceil(taps * c / 128) tiles is partial: 9 x 96 = 864 columns fill 96.4 % of seven tiles, where the per-tap
// enumeration of the other configurations fills 75 % (128 x 96 rows x columns for 96 x 96).  Staged like Big.
typedef Cfg<1, 4, 3, 1, 16> Flat96;

// Division by a launch constant as multiply-high + shift (Granlund-Montgomery, n < 2^31): the host fills
// (mul, shift) per divisor.  The prologue of a convolution workgroup decodes its tile, its K range and the
// (n, y, x) of its staged rows with ~12 divisions by runtime values; as emulated 32-bit divisions (~35 VALU
// instructions each, the uniform ones included: there is no scalar divide) they were ~40 % of the ~800
// instructions in front of the first load of launches that take 7-10 us in all.
struct FastDiv {
  unsigned mul, shift;
};
__host__ __device__ __forceinline__ unsigned fdiv(unsigned n, FastDiv f) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (__umulhi(n, f.mul) + n) >> f.shift;
#else
  return (unsigned)((((unsigned long long)n * f.mul) >> 32) + n) >> f.shift;
#endif
}

struct ConvArgs {
  const float* src;    // gathered activations (F: X, D: dY, W: X)
  const float* mat;    // F: Wt [Nout][RS][Cs]; D: WT [Nout][RS][Cs]; W: dY [M][Kout]
  float* out;
  float* ws;           // split-K partial tiles
  int* tickets;        // one counter per output tile, zero between launches
  int dgrad;           // gather rule: 0 = forward window, 1 = transposed (data gradient)
  // geometry of the gathered tensor and of the row index space
  int rows;            // M: F: N*OH*OW, D: N*H*W, W: N*OH*OW
  int rh, rw;          // spatial extent the row index decodes over (F/W: OH,OW; D: H,W)
  int sh_, sw_;        // spatial extent of the gathered tensor (F/W: H,W; D: OH,OW)
  int cs, cs_ld;       // channels gathered per pixel, pixel stride of src
  int mat_ld;          // NT: floats between consecutive taps of `mat` (>= cs: `mat` may be the first-channels
                       //     slice of a wider [Nout][RS][mat_ld] buffer -- the W half of a [W | v_W] operand)
  int nout, ldc;       // output columns (F: K, D: C, W: unused), output row stride
  int kout;            // W: number of dY channels (output rows of dW)
  int R, S, stride_h, stride_w, pad_h, pad_w;
  int ntaps;           // live taps
  // live taps as (r | s << 8).  DWORDS on purpose: a dynamically indexed BYTE array in the kernel arguments
  // compiles to global_load_ubyte + s_waitcnt vmcnt(0) (gfx950 has no scalar byte load), which drained the
  // three-deep prefetch ring at every step; a dword is one s_load on the scalar (lgkm) counter
  int tap_rs[MAX_TAPS];
  int tiles_m, tiles_n, splits, steps;  // steps = reduction steps in total
  int scalar;          // 1: channel counts not multiples of 4 -> element-wise gathers
  int shift_h, shift_w;  // log2 of the strides when both are powers of two, else -1
  int out_c;           // W: channels of X that get an output column (<= cs: X may be zero-padded)
  int slabs;           // 1: split s writes its partial result, in the output's own layout, to
  long long slab_stride;  //    out + s*slab_stride; the CONSUMER kernel sums the slabs in its prologue
  int big;             // tile configuration this problem was set up for (0: Small, 1: Big, 2: Big96, 3: Flat96)
  // Data gradient of a STRIDED convolution, decomposed by residue class of the input pixel: pixel (y, x) only
  // receives the taps with (y + pad - r) % stride == 0, so rows are enumerated class by class ((y % sh, x % sw)
  // fixed within a row tile) and every tile loops over ITS taps only -- a stride-2 3x3 layer does 9/4 taps per
  // pixel instead of 9, of which 3/4 would gather nothing.  ncls == 0: plain enumeration.
  int ncls;
  int cls_tile0[5];    // first row tile of class i ([ncls] = tiles_m)
  int cls_tap0[5];     // taps of class i: tap_r/tap_s[cls_tap0[i] .. cls_tap0[i+1])
  int cls_h[4], cls_w[4], cls_py[4], cls_px[4];  // pixels of the class: y = y'*stride_h + py, y' < cls_h
  // launch constants as divisors (filled by seal() right before the launch)
  FastDiv fd_tiles_n, fd_tiles_m, fd_splits, fd_rw, fd_rh, fd_csteps, fd_cblocks, fd_cs;
  // BNSUM instantiations (tangent convolution in front of a TRAIN-mode BatchNorm): per-channel partial sums of this
  // workgroup's tile, sum(t) and sum(t * xhat), xhat = (bn_x - bn_mean) * bn_rstd, to row (tile_m * splits + split)
  // of bn_part_1 / bn_part_x ([tiles_m * splits][nout]) -- what the reduction launch between the convolution and the
  // elementwise pass computed (hf_chan_affine_bwd_ex with gx = NULL); the elementwise pass adds the rows up
  const float *bn_x, *bn_mean, *bn_rstd;
  float *bn_part_x, *bn_part_1;
};

// One scalar load per 64-byte line of the argument block, all in flight together, before anything else: the
// compiler loads kernel arguments where they are first used -- a chain of ~5 dependent scalar-cache misses
// (~0.5 us each) through a ~650-byte block in the prologue of launches that take 7-10 us.  After this the
// later loads hit the scalar cache.
__device__ __forceinline__ void touch_args(const ConvArgs& a) {
  const int* ka = reinterpret_cast<const int*>(&a);
  int sink = 0;
#pragma unroll
  for (unsigned l = 0; l < (sizeof(ConvArgs) + 63) / 64; ++l) {
    const unsigned w = l * 16 < sizeof(ConvArgs) / 4 ? l * 16 : (unsigned)(sizeof(ConvArgs) / 4 - 1);
    sink ^= ka[w];
  }
  asm volatile("" ::"s"(sink));
}

// Pixel index into the gathered tensor for row coordinates (n, y, x) and tap (r, q); `valid` is
// false for padding.  Branch-free on purpose (selects only): the staging loads must not sit
// behind per-lane branches, or hipcc stops pipelining them.
__device__ __forceinline__ int gather_pixel(const ConvArgs& a, int n, int y, int x, int r, int q, bool& valid) {
  int sy, sx;
  bool v = true;
  if (!a.dgrad) {  // (uniform: a scalar branch)
    sy = y * a.stride_h - a.pad_h + r;
    sx = x * a.stride_w - a.pad_w + q;
  } else {
    const int ty = y + a.pad_h - r, tx = x + a.pad_w - q;
    if (a.shift_h >= 0) { sy = ty >> a.shift_h; sx = tx >> a.shift_w; }  // power-of-two strides
    else { sy = ty / a.stride_h; sx = tx / a.stride_w; }
    v = (ty >= 0) & (tx >= 0) & (sy * a.stride_h == ty) & (sx * a.stride_w == tx);
  }
  v = v & ((unsigned)sy < (unsigned)a.sh_) & ((unsigned)sx < (unsigned)a.sw_);
  valid = v;
  return v ? (n * a.sh_ + sy) * a.sw_ + sx : 0;
}

__device__ __forceinline__ float4 ldg4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// Staging loads the compiler does not see as loads: hipcc schedules the first use of a loaded
// register right behind the load (and waits there), which serialises the three-deep prefetch
// ring below.  Issued as inline asm they stay in flight until the matching HF_WAIT_SLOT, whose
// "+v" operands tie every later use of the registers to the wait (cdna_hip_programming.md
// section 5.7: count the queue by hand).
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void gload4(f32x4& dst, const float* p) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
#define HF_WAIT_SLOT(N, A, B) \
  asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(A[0]), "+v"(A[1]), "+v"(B[0]), "+v"(B[1]) : : "memory")

// four consecutive floats of which only the first `valid` exist, no alignment assumed
// (channel counts that are not multiples of 4: the 49-tap im2col of a 1-channel stem)
__device__ __forceinline__ float4 ldg4s(const float* p, int valid) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (valid > 0) v.x = p[0];
  if (valid > 1) v.y = p[1];
  if (valid > 2) v.z = p[2];
  if (valid > 3) v.w = p[3];
  return v;
}

// Last-arriver reduction of the split-K partials of one block tile (fixed order).  The wave at
// (wm, wn) holds TM x TN MFMA tiles: tile (im, in) covers rows wm*32*TM + im*32 .., cols likewise.
template <typename C, bool CLS = false, bool BNSUM = false>
__device__ __forceinline__ void finish_tile(const ConvArgs& a, const f32x16 (&acc)[C::TM][C::TN], int tile, int split,
                                            int wm, int wn, int lane, float* out_tile_base,
                                            int row0, int col0, int row_lim, int col_lim, int ldc,
                                            int* flag_lds, int cls = -1, float* lds = nullptr) {
  constexpr int BM = C::BM, BN = C::BN;
  const int i = lane & 31, h = lane >> 5;
  if constexpr (BNSUM) {
    static_assert(C::TM == 1 && C::TN == 1 && !CLS, "per-tile column sums: 64x64 tiles, plain row enumeration");
    // column `col` of this tile: 2 waves (wm) x 2 lane halves (h) x 16 accumulator rows each.  Every load first
    // (the tile's xhat operand), fp64 sums per lane, the four shares combined through LDS in a fixed order.
    const int col = wn * 32 + i, gcol = col0 + col;
    const bool cok = gcol < col_lim;
    const float mu = cok ? a.bn_mean[gcol] : 0.f, rs = cok ? a.bn_rstd[gcol] : 0.f;
    float xv[16];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int orow = row0 + wm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
      xv[reg] = (cok && orow < row_lim) ? a.bn_x[(size_t)orow * ldc + gcol] : mu;
    }
    double s1 = 0.0, sx = 0.0;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int orow = row0 + wm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
      if (cok && orow < row_lim) {
        const float v = acc[0][0][reg];
        s1 += (double)v;
        sx += (double)v * (double)(float)((xv[reg] - mu) * rs);
      }
    }
    double* red = reinterpret_cast<double*>(lds);  // [2][4][BN]; the caller synchronised after its last LDS read
    red[(0 * 4 + wm * 2 + h) * BN + col] = s1;
    red[(1 * 4 + wm * 2 + h) * BN + col] = sx;
    __syncthreads();
    const int t = threadIdx.x;
    if (t < 2 * BN) {
      const int kind = t / BN, c = t - kind * BN;
      const double sum = ((red[(kind * 4 + 0) * BN + c] + red[(kind * 4 + 1) * BN + c]) +
                          red[(kind * 4 + 2) * BN + c]) + red[(kind * 4 + 3) * BN + c];
      const int tile_m = (int)fdiv((unsigned)tile, a.fd_tiles_n);
      if (col0 + c < col_lim)
        (kind ? a.bn_part_x : a.bn_part_1)[(size_t)(tile_m * a.splits + split) * a.nout + col0 + c] = (float)sum;
    }
  }
  if (a.splits == 1 || a.slabs) {
    float* dst = out_tile_base + (a.slabs ? (size_t)split * (size_t)a.slab_stride : 0);
#pragma unroll
    for (int im = 0; im < C::TM; ++im)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int row = (wm * C::TM + im) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
        int orow = row0 + row;  // output row (= pixel index) of this accumulator row
        bool rok = orow < row_lim;
        if (CLS && cls >= 0) {  // (uniform) class-major enumeration: row -> (n, y', x') -> pixel (n, y'*sh + py, x'*sw + px)
          const int wc = a.cls_w[cls], hc = a.cls_h[cls];
          const int xq = orow % wc, tq = orow / wc;
          const int yq = tq % hc, nq = tq / hc;
          orow = (nq * a.rh + yq * a.stride_h + a.cls_py[cls]) * a.rw + xq * a.stride_w + a.cls_px[cls];
        }
#pragma unroll
        for (int in = 0; in < C::TN; ++in) {
          const int col = (wn * C::TN + in) * 32 + i;
          if (rok && col0 + col < col_lim) dst[(size_t)orow * ldc + col0 + col] = acc[im][in][reg];
        }
      }
    return;
  }
  const int tiles = a.tiles_m * a.tiles_n;
  float* mine = a.ws + ((size_t)split * tiles + tile) * (BM * BN);
#pragma unroll
  for (int im = 0; im < C::TM; ++im)
#pragma unroll
    for (int in = 0; in < C::TN; ++in)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int row = (wm * C::TM + im) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
        const int col = (wn * C::TN + in) * 32 + i;
        // write-through (sc1) store: visible device-wide once drained, no release fence needed
        __hip_atomic_store(mine + row * BN + col, acc[im][in][reg], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
  // publish (cdna_hip_programming.md, in-launch split-K reduction, write-through form):
  // every wave drains its sc1 stores, ONE lane draws the ticket; the last arriver's lane 0
  // acquires once (drops stale lines), then all its waves read the slabs with plain loads
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const int old = __hip_atomic_fetch_add(a.tickets + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = old == a.splits - 1;
    if (last) {
      __hip_atomic_store(a.tickets + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // next launch
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    *flag_lds = last;
  }
  __syncthreads();
  if (!*flag_lds) return;
  const float* base = a.ws + (size_t)tile * (BM * BN);
  const size_t slab = (size_t)tiles * (BM * BN);
  for (int e = threadIdx.x; e < BM * BN / 4; e += CT) {
    float4 s = *reinterpret_cast<const float4*>(base + 4 * e);
    int sp = 1;
    for (; sp + 4 <= a.splits; sp += 4) {  // four slabs in flight, summed in split order
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(base + (sp + u) * slab + 4 * e);
#pragma unroll
      for (int u = 0; u < 4; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
    for (; sp < a.splits; ++sp) {
      const float4 v = *reinterpret_cast<const float4*>(base + sp * slab + 4 * e);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    const int row = (4 * e) / BN, col = (4 * e) % BN;
    if (row0 + row < row_lim) {
      float* o = out_tile_base + (size_t)(row0 + row) * ldc + col0 + col;
      const float v[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (col0 + col + c < col_lim) o[c] = v[c];
    }
  }
}

// ---------------------------------------------------------------------------------
// NT kernel: F and D.   out[m][j] = sum_{tap, c} src[pix(m,tap)][c] * mat[j][tap][c]
// ---------------------------------------------------------------------------------
// CLS: the instantiation that understands the residue-class enumeration of strided data gradients (ConvArgs::ncls);
// the plain one keeps the prologue of the latency-bound small-map launches short (+1 us per launch otherwise).
template <bool SCALAR, typename C, bool CLS = false, bool BNSUM = false>
__device__ __forceinline__ void conv_nt_body(const ConvArgs& a, float* lds, int bid) {
  constexpr int BM = C::BM, BN = C::BN, BK = C::BK, KQ = C::KQ, RPT = C::RPT, HK = C::HK, LDK = C::LDK;
  constexpr int NU = 2;
  constexpr int SM = C::SM, SN = C::SN;
  float (*As)[SM][LDK] = reinterpret_cast<float (*)[SM][LDK]>(lds);
  float (*Bs)[SN][LDK] = reinterpret_cast<float (*)[SN][LDK]>(lds + 2 * SM * LDK);
  int& flag = *reinterpret_cast<int*>(lds + C::LDS_FLOATS - 4);
  touch_args(a);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave / C::WN, wn = wave % C::WN;
  const int bq = (int)fdiv((unsigned)bid, a.fd_tiles_n), tile_n = bid - bq * a.tiles_n;
  const int split = (int)fdiv((unsigned)bq, a.fd_tiles_m), tile_m = bq - split * a.tiles_m;
  const int tile = tile_m * a.tiles_n + tile_n;
  const int csteps = (a.cs + BK - 1) / BK;
  // residue class of this row tile (strided data gradients, see ConvArgs): its rows, taps and step count
  int cls = -1, row_base = tile_m * BM, rows_c = a.rows, tap0 = 0, steps = a.steps;
  if (CLS && a.ncls > 0) {  // (uniform)
    cls = 0;
    while (cls + 1 < a.ncls && tile_m >= a.cls_tile0[cls + 1]) ++cls;
    row_base = (tile_m - a.cls_tile0[cls]) * BM;
    rows_c = (a.rows / (a.rh * a.rw)) * a.cls_h[cls] * a.cls_w[cls];  // images x pixels of the class
    tap0 = a.cls_tap0[cls];
    steps = (a.cls_tap0[cls + 1] - tap0) * csteps;
  }
  const int per = (int)fdiv((unsigned)(steps + a.splits - 1), a.fd_splits);
  const int j0 = split * per < steps ? split * per : steps, j1 = (j0 + per < steps) ? j0 + per : steps;

  // staging assignment: thread -> rows lr + RPT*u and the float4 at k offset 4*kq
  const int lr = t / KQ, kq = t % KQ;
  int rn[NU], ry[NU], rx[NU];
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int m = row_base + lr + RPT * u;
    if (m < rows_c) {
      if (CLS && cls >= 0) {
        const int wc = a.cls_w[cls], hc = a.cls_h[cls];
        const int xx = m % wc, tq = m / wc;
        rx[u] = xx * a.stride_w + a.cls_px[cls]; ry[u] = (tq % hc) * a.stride_h + a.cls_py[cls]; rn[u] = tq / hc;
      } else {
        const int tq = (int)fdiv((unsigned)m, a.fd_rw), xx = m - tq * a.rw;
        rn[u] = (int)fdiv((unsigned)tq, a.fd_rh);
        rx[u] = xx; ry[u] = tq - rn[u] * a.rh;
      }
    } else {
      rn[u] = -1; ry[u] = rx[u] = 0;
    }
  }
  const int RS = a.R * a.S;
  const float* brow[NU];
  bool bok[NU];
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int j = tile_n * BN + lr + RPT * u;
    bok[u] = (j < a.nout) & (lr + RPT * u < BN);  // (rows past BN belong to the next column tile)
    brow[u] = a.mat + (size_t)(bok[u] ? j : 0) * RS * a.mat_ld;
  }

  // Global -> register -> LDS staging with THREE steps of loads in flight: these kernels run
  // one or two workgroups per CU on operands that the previous launch has just written (cold
  // in this XCD's L2), so a step is bound by the ~2 us round trip unless several are
  // outstanding.  LDS stays double-buffered; the register ring is three deep.
  // (loads are issued UNCONDITIONALLY -- an out-of-range element reads a valid dummy address
  // and is zeroed when the registers are written to LDS: a branch around a load makes hipcc
  // drain the whole queue, vmcnt(0), before the next use; cdna_hip_programming.md trap 4c)
  auto fetch = [&](int step, float4 (&ra)[NU], float4 (&rb)[NU]) -> unsigned {
    const int tl = step / csteps, c = (step - tl * csteps) * BK + 4 * kq, ti = tap0 + tl;
    const int rs_ = a.tap_rs[ti], r = rs_ & 255, q = rs_ >> 8;
    const bool cok = c < a.cs;
    unsigned ok = 0;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      bool va;
      const int pix = gather_pixel(a, rn[u], ry[u], rx[u], r, q, va);
      va = va & cok & (rn[u] >= 0);
      const bool vb = cok & bok[u];
      const float* pa = a.src + (va ? (size_t)pix * a.cs_ld + c : 0);
      const float* pb = vb ? brow[u] + (size_t)(r * a.S + q) * a.mat_ld + c : a.mat;
      ra[u] = ldg4s(pa, va ? a.cs - c : 0);
      rb[u] = ldg4s(pb, vb ? a.cs - c : 0);
      ok |= (va ? 1u : 0u) << (2 * u) | (vb ? 2u : 0u) << (2 * u);
    }
    return ok;
  };
  auto stash = [&](int buf, const float4 (&ra)[NU], const float4 (&rb)[NU], unsigned ok) {
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      *reinterpret_cast<float4*>(&As[buf][lr + RPT * u][4 * kq]) = (ok >> (2 * u)) & 1u ? ra[u] : zero;
      *reinterpret_cast<float4*>(&Bs[buf][lr + RPT * u][4 * kq]) = (ok >> (2 * u)) & 2u ? rb[u] : zero;
    }
  };

  f32x16 acc[C::TM][C::TN];
#pragma unroll
  for (int im = 0; im < C::TM; ++im)
#pragma unroll
    for (int in = 0; in < C::TN; ++in)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[im][in][r] = 0.f;
  const int i = lane & 31, h = lane >> 5;
  auto compute = [&](int cur) {
    float av[C::TM][HK], bv[C::TN][HK];
#pragma unroll
    for (int im = 0; im < C::TM; ++im) {
      const float* ap = &As[cur][(wm * C::TM + im) * 32 + i][HK * h];
#pragma unroll
      for (int v = 0; v < HK / 4; ++v)
        *reinterpret_cast<float4*>(av[im] + 4 * v) = *reinterpret_cast<const float4*>(ap + 4 * v);
    }
#pragma unroll
    for (int in = 0; in < C::TN; ++in) {
      const float* bp = &Bs[cur][(wn * C::TN + in) * 32 + i][HK * h];
#pragma unroll
      for (int v = 0; v < HK / 4; ++v)
        *reinterpret_cast<float4*>(bv[in] + 4 * v) = *reinterpret_cast<const float4*>(bp + 4 * v);
    }
#pragma unroll
    for (int s = 0; s < HK; ++s)
#pragma unroll
      for (int im = 0; im < C::TM; ++im)
#pragma unroll
        for (int in = 0; in < C::TN; ++in)
          acc[im][in] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[im][s], bv[in][s], acc[im][in], 0, 0, 0);
  };
  const int jl = j1 - 1;
  int cur = 0;
  if (!SCALAR) {
    // hand-counted ring: always three steps (12 loads) in flight; steps beyond the split's range
    // re-fetch step j1-1 and are never written to LDS, so the count at every wait is 8
    // steps are fetched in increasing order (the clamped re-fetches at the end repeat the last
    // one): the (tap, channel block) of the next fetch is kept incrementally -- no division per step
    int f_step = j0, f_ti = (int)fdiv((unsigned)j0, a.fd_csteps), f_cb = j0 - f_ti * csteps;
    // the tap table entry of the NEXT tap is loaded when a tap begins: the scalar load (a dynamic index into
    // the kernel arguments) has a whole tap's steps to land instead of stalling the step that needs it
    auto tap_at = [&](int ti) { return a.tap_rs[ti < MAX_TAPS ? ti : MAX_TAPS - 1]; };
    int f_rs = tap_at(tap0 + f_ti), f_rs_next = tap_at(tap0 + f_ti + 1);
    auto fetch_v = [&](int step, f32x4 (&ra)[NU], f32x4 (&rb)[NU]) -> unsigned {
      if (step > f_step) {  // (uniform; steps advance by one)
        f_step = step;
        if (++f_cb == csteps) {
          f_cb = 0;
          ++f_ti;
          f_rs = f_rs_next;
          f_rs_next = tap_at(tap0 + f_ti + 1);
        }
      }
      const int c = f_cb * BK + 4 * kq;
      const int rs_ = f_rs, r = rs_ & 255, q = rs_ >> 8;
      const bool cok = c < a.cs;
      unsigned ok = 0;
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        bool va;
        const int pix = gather_pixel(a, rn[u], ry[u], rx[u], r, q, va);
        va = va & cok & (rn[u] >= 0);
        const bool vb = cok & bok[u];
        gload4(ra[u], a.src + (va ? (size_t)pix * a.cs_ld + c : 0));
        gload4(rb[u], vb ? brow[u] + (size_t)(r * a.S + q) * a.mat_ld + c : a.mat);
        ok |= (va ? 1u : 0u) << (2 * u) | (vb ? 2u : 0u) << (2 * u);
      }
      return ok;
    };
    auto stash_v = [&](int buf, const f32x4 (&ra)[NU], const f32x4 (&rb)[NU], unsigned ok) {
      const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        *reinterpret_cast<f32x4*>(&As[buf][lr + RPT * u][4 * kq]) = (ok >> (2 * u)) & 1u ? ra[u] : zero;
        *reinterpret_cast<f32x4*>(&Bs[buf][lr + RPT * u][4 * kq]) = (ok >> (2 * u)) & 2u ? rb[u] : zero;
      }
    };
    f32x4 a0[NU], b0[NU], a1[NU], b1[NU], a2[NU], b2[NU];
    unsigned k0 = 0, k1 = 0, k2 = 0;
    if (j0 < j1) {
      k0 = fetch_v(j0, a0, b0);
      k1 = fetch_v(j0 + 1 < j1 ? j0 + 1 : jl, a1, b1);
      k2 = fetch_v(j0 + 2 < j1 ? j0 + 2 : jl, a2, b2);
      for (int j = j0; j < j1; j += 3) {
        HF_WAIT_SLOT(8, a0, b0);
        stash_v(cur, a0, b0, k0);
        __syncthreads();
        k0 = fetch_v(j + 3 < j1 ? j + 3 : jl, a0, b0);
        compute(cur);
        cur ^= 1;
        HF_WAIT_SLOT(8, a1, b1);
        if (j + 1 < j1) {
          stash_v(cur, a1, b1, k1);
          __syncthreads();
          compute(cur);
          cur ^= 1;
        }
        k1 = fetch_v(j + 4 < j1 ? j + 4 : jl, a1, b1);
        HF_WAIT_SLOT(8, a2, b2);
        if (j + 2 < j1) {
          stash_v(cur, a2, b2, k2);
          __syncthreads();
          compute(cur);
          cur ^= 1;
        }
        k2 = fetch_v(j + 5 < j1 ? j + 5 : jl, a2, b2);
      }
      HF_WAIT_SLOT(0, a0, b0);  // drain: the registers may be reused from here on
      HF_WAIT_SLOT(0, a1, b1);
      HF_WAIT_SLOT(0, a2, b2);
    }
  } else {
    float4 a0[NU], b0[NU];
    for (int j = j0; j < j1; ++j) {  // element-wise gathers: plain one-step loop
      const unsigned k0 = fetch(j, a0, b0);
      stash(cur, a0, b0, k0);
      __syncthreads();
      compute(cur);
      cur ^= 1;
    }
  }
  __syncthreads();  // the LDS array is reused below (last-arriver flag)
  finish_tile<C, CLS, BNSUM>(a, acc, tile, split, wm, wn, lane, a.out, row_base, tile_n * BN, rows_c, a.nout, a.ldc,
                             &flag, cls, lds);
}

// ---------------------------------------------------------------------------------
// TN kernel: W.   dW[k][tap][c] = sum_m dY[m][k] * X[pix(m,tap)][c]
//   tile_m over k (dY channels), tile_n over (live tap, BN-channel block of X)
// ---------------------------------------------------------------------------------
// FLAT: the output columns are enumerated flat over (tap, channel) (see Flat96): column tile tile_n covers the
// flattened columns [tile_n * BN, ...), every staging thread decodes the tap of ITS column quad once.
template <bool SCALAR, typename C, bool FLAT = false>
__device__ __forceinline__ void conv_tn_body(const ConvArgs& a, float* lds, int bid) {
  constexpr int BM = C::BM, BN = C::BN, BK = C::BK, HK = C::HK, LDA = C::LDA, LDB = C::LDB;
  float (*As)[BK][LDA] = reinterpret_cast<float (*)[BK][LDA]>(lds);
  float (*Bs)[BK][LDB] = reinterpret_cast<float (*)[BK][LDB]>(lds + 2 * BK * LDA);
  int& flag = *reinterpret_cast<int*>(lds + C::LDS_FLOATS - 4);
  touch_args(a);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave / C::WN, wn = wave % C::WN;
  const int bq = (int)fdiv((unsigned)bid, a.fd_tiles_n), tile_n = bid - bq * a.tiles_n;
  const int split = (int)fdiv((unsigned)bq, a.fd_tiles_m), tile_m = bq - split * a.tiles_m;
  const int tile = tile_m * a.tiles_n + tile_n;
  const int per = (int)fdiv((unsigned)(a.steps + a.splits - 1), a.fd_splits);
  const int j0 = split * per, j1 = (j0 + per < a.steps) ? j0 + per : a.steps;

  const int cblocks = (a.cs + BN - 1) / BN;
  int ti, c0;
  if constexpr (FLAT) { ti = 0; c0 = tile_n * BN; }
  else { ti = (int)fdiv((unsigned)tile_n, a.fd_cblocks); c0 = (tile_n - ti * cblocks) * BN; }

  // staging: thread -> rows kr, kr + RP of the step and the float4 at column 4*cq
  constexpr int CQ = C::SM / 4;     // float4 per staged row (both operands are staged SM == SN wide)
  constexpr int RP = CT / CQ;       // rows per pass
  constexpr int TU = BK / RP;       // passes
  static_assert(TU == 2, "the hand-counted prefetch ring assumes 2 + 2 loads per step (TN)");
  const int kr = t / CQ, cq = t % CQ;
  const int ka = tile_m * BM + 4 * cq;  // dY channel
  int cb = c0 + 4 * cq;                 // X channel
  bool bkok = (cb < a.cs) & (4 * cq < BN);
  if constexpr (FLAT) {                 // (flattened column -> this thread's tap and channel; quads never straddle taps)
    const int jq = c0 + 4 * cq;
    bkok = (jq < a.ntaps * a.cs) & (4 * cq < BN);
    ti = bkok ? (int)fdiv((unsigned)jq, a.fd_cs) : 0;
    cb = jq - ti * a.cs;
  }
  const int rs_ = a.tap_rs[ti], r = rs_ & 255, q = rs_ >> 8;
  const bool aok = (ka < a.kout) & (4 * cq < BM);

  auto fetch = [&](int step, float4 (&ra)[TU], float4 (&rb)[TU]) -> unsigned {
    unsigned ok = 0;
#pragma unroll
    for (int u = 0; u < TU; ++u) {
      const int m = step * BK + kr + RP * u;
      const bool mok = m < a.rows;
      const int mm = mok ? m : 0;
      const int xx = mm % a.rw, tq = mm / a.rw;
      bool vb;
      const int pix = gather_pixel(a, tq / a.rh, tq % a.rh, xx, r, q, vb);
      vb = vb & mok & bkok;
      const bool va = mok & aok;
      const float* pa = a.mat + (va ? (size_t)m * a.kout + ka : 0);
      const float* pb = a.src + (vb ? (size_t)pix * a.cs_ld + cb : 0);
      ra[u] = ldg4s(pa, va ? a.kout - ka : 0);
      rb[u] = ldg4s(pb, vb ? a.cs - cb : 0);
      ok |= (va ? 1u : 0u) << (2 * u) | (vb ? 2u : 0u) << (2 * u);
    }
    return ok;
  };
  auto stash = [&](int buf, const float4 (&ra)[TU], const float4 (&rb)[TU], unsigned ok) {
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < TU; ++u) {
      *reinterpret_cast<float4*>(&As[buf][kr + RP * u][4 * cq]) = (ok >> (2 * u)) & 1u ? ra[u] : zero;
      *reinterpret_cast<float4*>(&Bs[buf][kr + RP * u][4 * cq]) = (ok >> (2 * u)) & 2u ? rb[u] : zero;
    }
  };

  f32x16 acc[C::TM][C::TN];
#pragma unroll
  for (int im = 0; im < C::TM; ++im)
#pragma unroll
    for (int in = 0; in < C::TN; ++in)
#pragma unroll
      for (int x = 0; x < 16; ++x) acc[im][in][x] = 0.f;
  const int i = lane & 31, h = lane >> 5;
  auto compute = [&](int cur) {
    float av[C::TM][HK], bv[C::TN][HK];
#pragma unroll
    for (int s = 0; s < HK; ++s) {
#pragma unroll
      for (int im = 0; im < C::TM; ++im) av[im][s] = As[cur][HK * h + s][(wm * C::TM + im) * 32 + i];
#pragma unroll
      for (int in = 0; in < C::TN; ++in) bv[in][s] = Bs[cur][HK * h + s][(wn * C::TN + in) * 32 + i];
    }
#pragma unroll
    for (int s = 0; s < HK; ++s)
#pragma unroll
      for (int im = 0; im < C::TM; ++im)
#pragma unroll
        for (int in = 0; in < C::TN; ++in)
          acc[im][in] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[im][s], bv[in][s], acc[im][in], 0, 0, 0);
  };
  const int jl = j1 - 1;
  int cur = 0;
  if (!SCALAR) {
    // the rows of consecutive steps advance by BK: (n, y, x) of each staged row is kept
    // incrementally (the per-step divisions were ~300 VALU instructions per thread and step)
    int f_step = j0, fn_[TU], fy_[TU], fx_[TU];
#pragma unroll
    for (int u = 0; u < TU; ++u) {
      const int m = j0 * BK + kr + RP * u;
      const int tq = (int)fdiv((unsigned)m, a.fd_rw), xx = m - tq * a.rw;
      fn_[u] = (int)fdiv((unsigned)tq, a.fd_rh);
      fx_[u] = xx; fy_[u] = tq - fn_[u] * a.rh;
    }
    auto fetch_v = [&](int step, f32x4 (&ra)[TU], f32x4 (&rb)[TU]) -> unsigned {
      if (step > f_step) {  // (uniform; steps advance by one)
        f_step = step;
#pragma unroll
        for (int u = 0; u < TU; ++u) {
          fx_[u] += BK;
          while (fx_[u] >= a.rw) { fx_[u] -= a.rw; if (++fy_[u] == a.rh) { fy_[u] = 0; ++fn_[u]; } }
        }
      }
      unsigned ok = 0;
#pragma unroll
      for (int u = 0; u < TU; ++u) {
        const int m = step * BK + kr + RP * u;
        const bool mok = m < a.rows;
        bool vb;
        const int pix = gather_pixel(a, fn_[u], fy_[u], fx_[u], r, q, vb);
        vb = vb & mok & bkok;
        const bool va = mok & aok;
        gload4(ra[u], a.mat + (va ? (size_t)m * a.kout + ka : 0));
        gload4(rb[u], a.src + (vb ? (size_t)pix * a.cs_ld + cb : 0));
        ok |= (va ? 1u : 0u) << (2 * u) | (vb ? 2u : 0u) << (2 * u);
      }
      return ok;
    };
    auto stash_v = [&](int buf, const f32x4 (&ra)[TU], const f32x4 (&rb)[TU], unsigned ok) {
      const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < TU; ++u) {
        *reinterpret_cast<f32x4*>(&As[buf][kr + RP * u][4 * cq]) = (ok >> (2 * u)) & 1u ? ra[u] : zero;
        *reinterpret_cast<f32x4*>(&Bs[buf][kr + RP * u][4 * cq]) = (ok >> (2 * u)) & 2u ? rb[u] : zero;
      }
    };
    f32x4 a0[TU], b0[TU], a1[TU], b1[TU], a2[TU], b2[TU];
    unsigned k0 = 0, k1 = 0, k2 = 0;
    if (j0 < j1) {
      k0 = fetch_v(j0, a0, b0);
      k1 = fetch_v(j0 + 1 < j1 ? j0 + 1 : jl, a1, b1);
      k2 = fetch_v(j0 + 2 < j1 ? j0 + 2 : jl, a2, b2);
      for (int j = j0; j < j1; j += 3) {
        HF_WAIT_SLOT(8, a0, b0);
        stash_v(cur, a0, b0, k0);
        __syncthreads();
        k0 = fetch_v(j + 3 < j1 ? j + 3 : jl, a0, b0);
        compute(cur);
        cur ^= 1;
        HF_WAIT_SLOT(8, a1, b1);
        if (j + 1 < j1) {
          stash_v(cur, a1, b1, k1);
          __syncthreads();
          compute(cur);
          cur ^= 1;
        }
        k1 = fetch_v(j + 4 < j1 ? j + 4 : jl, a1, b1);
        HF_WAIT_SLOT(8, a2, b2);
        if (j + 2 < j1) {
          stash_v(cur, a2, b2, k2);
          __syncthreads();
          compute(cur);
          cur ^= 1;
        }
        k2 = fetch_v(j + 5 < j1 ? j + 5 : jl, a2, b2);
      }
      HF_WAIT_SLOT(0, a0, b0);
      HF_WAIT_SLOT(0, a1, b1);
      HF_WAIT_SLOT(0, a2, b2);
    }
  } else {
    float4 a0[TU], b0[TU];
    for (int j = j0; j < j1; ++j) {
      const unsigned k0 = fetch(j, a0, b0);
      stash(cur, a0, b0, k0);
      __syncthreads();
      compute(cur);
      cur ^= 1;
    }
  }
  __syncthreads();
  // output element (k, tap, c) at (k*RS + tap)*cs + c: rows = k, "columns" = c within this tap
  if constexpr (FLAT) {
    // every tap live and out_c == cs: the columns ARE the flattened (tap, c) index.  A zero-padded operand
    // (out_c < cs) is only let through with a single tap (apply_out_c): the columns are then the channels, and the
    // padded ones are cut off here -- with several taps they would land in the next tap's / row's first entries
    finish_tile<C>(a, acc, tile, split, wm, wn, lane, a.out, tile_m * BM, c0, a.kout,
                   a.ntaps == 1 ? a.out_c : a.ntaps * a.cs, a.R * a.S * a.out_c, &flag);
  } else {
    float* base = a.out + (size_t)(r * a.S + q) * a.out_c;
    finish_tile<C>(a, acc, tile, split, wm, wn, lane, base, tile_m * BM, c0, a.kout, a.out_c,
                   a.R * a.S * a.out_c, &flag);
  }
}

template <bool SCALAR, typename C, bool CLS = false, bool BNSUM = false>
__global__ __launch_bounds__(CT) void k_conv_nt(const ConvArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[C::LDS_FLOATS];
  conv_nt_body<SCALAR, C, CLS, BNSUM>(a, lds, blockIdx.x);
}

// A forward / tangent convolution whose launch CARRIES the tangent sweep's weight scatter (hf_unpack_weights) as
// extra workgroups behind its own: the stem's convolution does not read any scattered operand (its v_W is a slice
// of the vector itself), the scatter only has to land before the NEXT launch -- as one launch the 14 us scatter
// hides behind the 12 us latency-bound convolution (they were 26 us in sequence).
template <bool SCALAR, typename C>
__global__ __launch_bounds__(CT) void k_conv_nt_unpack(const ConvArgs a, const hf_shared::UnpackArgs u,
                                                       const float* __restrict__ usrc, int conv_blocks) {
  __shared__ __attribute__((aligned(16))) float lds[C::LDS_FLOATS];
  if ((int)blockIdx.x < conv_blocks) conv_nt_body<SCALAR, C, false>(a, lds, blockIdx.x);
  else hf_shared::unpack_block<float>(usrc, u, blockIdx.x - (unsigned)conv_blocks);
}

template <bool SCALAR, typename C, bool FLAT = false>
__global__ __launch_bounds__(CT) void k_conv_tn(const ConvArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[C::LDS_FLOATS];
  conv_tn_body<SCALAR, C, FLAT>(a, lds, blockIdx.x);
}

// Data gradient AND weight gradient of one layer in ONE launch: both read the same dY,
// neither depends on the other; the first `nblocks_d` workgroups run the NT body.
// ANYBIG: some problem of the launch runs in the Big configuration (decided per problem, `a.big`); the
// all-Small instantiation keeps the register allocation of the small-map launches what it was.
template <bool ANYBIG, bool CLS = false>
__global__ __launch_bounds__(CT) void k_conv_dw(const ConvArgs d, const ConvArgs w, int nblocks_d) {
  constexpr int LDSB = Big::LDS_FLOATS > Big96::LDS_FLOATS ? Big::LDS_FLOATS : Big96::LDS_FLOATS;
  constexpr int LDSF = ANYBIG ? (LDSB > Small::LDS_FLOATS ? LDSB : Small::LDS_FLOATS) : Small::LDS_FLOATS;
  __shared__ __attribute__((aligned(16))) float lds[LDSF];
  if ((int)blockIdx.x < nblocks_d) {
    if (ANYBIG && d.big == 1) conv_nt_body<false, Big, CLS>(d, lds, blockIdx.x);
    else if (ANYBIG && d.big == 2) conv_nt_body<false, Big96, CLS>(d, lds, blockIdx.x);
    else conv_nt_body<false, Small, CLS>(d, lds, blockIdx.x);
  } else {
    if (ANYBIG && w.big == 1) conv_tn_body<false, Big>(w, lds, blockIdx.x - nblocks_d);
    else if (ANYBIG && w.big == 2) conv_tn_body<false, Big96>(w, lds, blockIdx.x - nblocks_d);
    else if (ANYBIG && w.big == 3) conv_tn_body<false, Flat96, true>(w, lds, blockIdx.x - nblocks_d);
    else conv_tn_body<false, Small>(w, lds, blockIdx.x - nblocks_d);
  }
}

// Up to GROUP_MAX independent convolutions (any mix of directions) in ONE launch: workgroups
// [start[p], start[p+1]) run problem p.  The curvature engine uses it where two layers read the
// same tensor (a residual block's first convolution and its downsample branch): one launch for
// both tangent convolutions, one for both layers' data + weight gradients.
constexpr int GROUP_MAX = 4;
struct GroupArgs {
  ConvArgs a[GROUP_MAX];
  int tn[GROUP_MAX];       // 1: weight gradient (TN body), 0: forward / data gradient (NT body)
  int start[GROUP_MAX + 1];
  int n;
};

// (the problems arrive as SEPARATE by-value arguments: indexing an array of them inside one by-value
// struct made hipcc copy all 1.2 KB to scratch memory, and every field read in the K loops a scratch
// load -- measured 3x on the 128-wide configurations)
struct GroupMeta {
  int tn[GROUP_MAX];       // 1: weight gradient (TN body), 0: forward / data gradient (NT body)
  int start[GROUP_MAX + 1];
  int n;
};

template <bool ANYBIG, bool CLS, bool BNSUM = false>
__device__ __forceinline__ void group_run(const ConvArgs& a, int tn, float* lds, int local) {
  if constexpr (BNSUM) {  // (tangent convolutions on 64x64 tiles only: checked on the host)
    if (a.bn_part_1) conv_nt_body<false, Small, false, true>(a, lds, local);
    else conv_nt_body<false, Small, false, false>(a, lds, local);
    return;
  }
  if (tn) {
    if (ANYBIG && a.big == 1) conv_tn_body<false, Big>(a, lds, local);
    else if (ANYBIG && a.big == 2) conv_tn_body<false, Big96>(a, lds, local);
    else if (ANYBIG && a.big == 3) conv_tn_body<false, Flat96, true>(a, lds, local);
    else conv_tn_body<false, Small>(a, lds, local);
  } else {
    if (ANYBIG && a.big == 1) conv_nt_body<false, Big, CLS>(a, lds, local);
    else if (ANYBIG && a.big == 2) conv_nt_body<false, Big96, CLS>(a, lds, local);
    else conv_nt_body<false, Small, CLS>(a, lds, local);
  }
}

template <bool ANYBIG, bool CLS = false, bool BNSUM = false>
__global__ __launch_bounds__(CT) void k_conv_group(const ConvArgs a0, const ConvArgs a1, const ConvArgs a2,
                                                   const ConvArgs a3, const GroupMeta m) {
  constexpr int LDSB = Big::LDS_FLOATS > Big96::LDS_FLOATS ? Big::LDS_FLOATS : Big96::LDS_FLOATS;
  constexpr int LDSF = ANYBIG ? (LDSB > Small::LDS_FLOATS ? LDSB : Small::LDS_FLOATS) : Small::LDS_FLOATS;
  __shared__ __attribute__((aligned(16))) float lds[LDSF];
  const int b = (int)blockIdx.x;
  if (b < m.start[1]) group_run<ANYBIG, CLS, BNSUM>(a0, m.tn[0], lds, b);
  else if (b < m.start[2]) group_run<ANYBIG, CLS, BNSUM>(a1, m.tn[1], lds, b - m.start[1]);
  else if (b < m.start[3]) group_run<ANYBIG, CLS, BNSUM>(a2, m.tn[2], lds, b - m.start[2]);
  else group_run<ANYBIG, CLS, BNSUM>(a3, m.tn[3], lds, b - m.start[3]);
}

inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

}  // namespace

#define HF_HIP(expr)                         \
  do {                                       \
    hipError_t e_ = (expr);                  \
    if (e_ != hipSuccess) return (int)e_;    \
  } while (0)

namespace {

// Number of K-splits.  target_blocks > 0: split until about that many workgroups exist
// (>= 2 steps per split).  Otherwise a cost model in microseconds (measured on MI355X):
// one 64x64xBK step costs a global->LDS round trip that the one-step prefetch only partly
// hides (~1.0 us at BK = 32, ~1.5 us at BK = 64; the fp32 MFMA work itself is 0.43 / 0.86 us),
// a split run pays ~3 us to publish and collect tickets plus ~0.15 us per slab the last
// arriver sums; workgroups beyond two per CU queue.
#ifndef HF_BIG_TARGET_BLOCKS
#define HF_BIG_TARGET_BLOCKS 512  // (whole-bench A/B, profiles/r05_big_target_blocks_ab*.jsonl: 256 / 384 / 512 / 768 / 1024 ->
#endif                            //  configs[3] 576 / 579 / 601 / 582 / 544, All-CNN-C GGN 761 / 782 / 817 / 817 / 754)
#ifndef HF_BIG_MIN_WORK
#define HF_BIG_MIN_WORK 6144  // (output tiles x K-steps from which a problem takes a 128-wide configuration)
#endif
constexpr int64_t BIG_TARGET_BLOCKS = HF_BIG_TARGET_BLOCKS;  // workgroups a 128-wide launch is split towards (measured: below)
constexpr int64_t FEW_TILES = 8;            // up to this many output tiles ...
constexpr int64_t FEW_TILES_CAP = 96;       // ... a weight gradient may be split this deep (default measured on ResNet-18)

int choose_splits(int64_t tiles, int64_t steps, int target_blocks, int64_t ws_bytes, bool slabs = false,
                  int bk = Small::BK, int64_t tile_elems = Small::BM * Small::BN) {
  int64_t best = 1;
  if (slabs && target_blocks <= 0) {
    // no in-launch reduction: a split costs its consumer one more read per element; split
    // until one workgroup per CU exists (measured in the ResNet-18 product: 256 beats 384,
    // 512 and 768 -- the prefetch ring keeps a workgroup with several steps busy) or a split
    // is down to 2 steps
    // The 128-wide configurations (bk == 16) run one MFMA-heavy workgroup per CU at 256: a second and
    // third resident workgroup fill the matrix pipe while the first stages its next tile (measured on
    // the All-CNN-C shapes, scripts/conv_kernel_bench.py --big 1: 256 / 512 / 768 workgroups ->
    // tangent 76 / 93 / 99, data gradient 68 / 83 / 87, weight gradient 39 / 52 / 65 TFLOP/s stand-alone.  INSIDE the
    // product every further split is one more slab for the consumer kernel to read, and a Hessian product runs a
    // second launch on a parallel graph branch: the whole bench prefers 512 to 768 -- HF_BIG_TARGET_BLOCKS)
    const bool big = bk != Small::BK;
    best = ((big ? BIG_TARGET_BLOCKS : 256) + tiles - 1) / tiles;
    if (big && best > steps / 8) best = steps / 8;  // (>= 8 steps per workgroup)
    if (best > steps / 2) best = steps / 2;
    // (one or two output tiles -- the stem's weight gradient: 6272 rows into a 64 x 52 matrix --
    // would leave most of the chip idle at 32 splits)
    // (FEW_TILES: up to how many output tiles the larger cap applies.  Measured, round 4,
    // profiles/r04_conv_few_tiles.jsonl: 2 / 4 / 8 tiles -> ResNet-50 topology 314 / 323 / 323 matvecs/s,
    // ResNet-18 1507 / 1512 / 1507, All-CNN-C 742 / 751 / 755.)
    const int64_t cap = tiles <= FEW_TILES ? FEW_TILES_CAP : (big ? 128 : 32);
    if (best > cap) best = cap;
    if (best < 1) best = 1;
    while (best > 1 && ((steps + best - 1) / best) * (best - 1) >= steps) --best;
    return (int)best;
  }
  if (target_blocks > 0) {
    best = (target_blocks + tiles - 1) / tiles;
    if (best > steps / 2) best = steps / 2;
    if (best < 1) best = 1;
  } else {
    double best_cost = 1e30;
    const int64_t cap = steps < 64 ? steps : 64;
    for (int64_t sp = 1; sp <= cap; ++sp) {
      const int64_t per = (steps + sp - 1) / sp;
      if (sp > 1 && per * (sp - 1) >= steps) continue;  // would leave a split empty
      const double rounds = (double)(tiles * sp) / 512.0;
      double cost = (0.6 + 0.014 * bk) * (double)per * (rounds > 1.0 ? rounds : 1.0);
      if (sp > 1) cost += 3.0 + 0.15 * (double)sp;
      if (cost < best_cost - 1e-9) { best_cost = cost; best = sp; }
    }
  }
  while (!slabs && best > 1 && best * tiles * tile_elems * (int64_t)sizeof(float) > ws_bytes) --best;
  while (best > 1 && ((steps + best - 1) / best) * (best - 1) >= steps) --best;
  return (int)best;
}

FastDiv make_fastdiv(int64_t d) {
  if (d < 1) d = 1;
  unsigned l = 0;
  while ((1LL << l) < d) ++l;  // ceil(log2 d)
  FastDiv f;
  f.mul = (unsigned)((((unsigned long long)((1ULL << l) - (unsigned long long)d)) << 32) / (unsigned long long)d + 1);
  f.shift = l;
  return f;
}

// the divisors of the kernels' prologues (see FastDiv); called on the copy that is handed to the launch
void seal(ConvArgs& a, int direction) {
  const int bk = a.big ? Big::BK : Small::BK;
  const int bn = a.big == 1 ? Big::BN : a.big == 2 ? Big96::BN : Small::BN;
  a.fd_tiles_n = make_fastdiv(a.tiles_n);
  a.fd_tiles_m = make_fastdiv(a.tiles_m);
  a.fd_splits = make_fastdiv(a.splits);
  a.fd_rw = make_fastdiv(a.rw);
  a.fd_rh = make_fastdiv(a.rh);
  a.fd_csteps = make_fastdiv((a.cs + bk - 1) / bk);
  a.fd_cblocks = make_fastdiv((a.cs + bn - 1) / bn);
  a.fd_cs = make_fastdiv(a.cs);
  (void)direction;
}

// Fill `a` for one direction; returns the number of workgroups (<= 0: error code).
void launch_one(int direction, const ConvArgs& a_in, int64_t blocks, hipStream_t stream) {
  ConvArgs a = a_in;
  seal(a, direction);
  const dim3 grid((unsigned)blocks), block(CT);
  if (direction <= 1) {
    if (a.scalar) hipLaunchKernelGGL((k_conv_nt<true, Small>), grid, block, 0, stream, a);
    else if (a.ncls > 0 && a.big == 1) hipLaunchKernelGGL((k_conv_nt<false, Big, true>), grid, block, 0, stream, a);
    else if (a.ncls > 0 && a.big == 2) hipLaunchKernelGGL((k_conv_nt<false, Big96, true>), grid, block, 0, stream, a);
    else if (a.ncls > 0) hipLaunchKernelGGL((k_conv_nt<false, Small, true>), grid, block, 0, stream, a);
    else if (a.big == 1) hipLaunchKernelGGL((k_conv_nt<false, Big>), grid, block, 0, stream, a);
    else if (a.big == 2) hipLaunchKernelGGL((k_conv_nt<false, Big96>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((k_conv_nt<false, Small>), grid, block, 0, stream, a);
  } else {
    if (a.scalar) hipLaunchKernelGGL((k_conv_tn<true, Small>), grid, block, 0, stream, a);
    else if (a.big == 1) hipLaunchKernelGGL((k_conv_tn<false, Big>), grid, block, 0, stream, a);
    else if (a.big == 2) hipLaunchKernelGGL((k_conv_tn<false, Big96>), grid, block, 0, stream, a);
    else if (a.big == 3) hipLaunchKernelGGL((k_conv_tn<false, Flat96, true>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((k_conv_tn<false, Small>), grid, block, 0, stream, a);
  }
}

void launch_dw(const ConvArgs& d_in, const ConvArgs& w_in, int64_t bd, int64_t bw, hipStream_t stream) {
  ConvArgs d = d_in, w = w_in;
  seal(d, 1);
  seal(w, 2);
  const dim3 grid((unsigned)(bd + bw)), block(CT);
  const bool big = d.big || w.big, cls = d.ncls > 0;
  if (big && cls) hipLaunchKernelGGL((k_conv_dw<true, true>), grid, block, 0, stream, d, w, (int)bd);
  else if (big) hipLaunchKernelGGL((k_conv_dw<true, false>), grid, block, 0, stream, d, w, (int)bd);
  else if (cls) hipLaunchKernelGGL((k_conv_dw<false, true>), grid, block, 0, stream, d, w, (int)bd);
  else hipLaunchKernelGGL((k_conv_dw<false, false>), grid, block, 0, stream, d, w, (int)bd);
}

// Which tile configuration a problem runs in: a pure function of its geometry (hf_conv2d_nhwc_plan and
// every launch path must agree).  Big (128x128) needs both output dimensions to fill most of a tile
// and enough GEMM rows that the 64x64 kernel would be issue-bound.
inline int hf_env_dclass() { return 1; }  // (strided data gradients are enumerated by residue class: 1 268 -> 1 292, round 3)

inline int hf_env_big() { return -1; }  // (-1: by geometry; 0 / 1 force a configuration when bisecting)

int want_big(int direction, int64_t rows, int64_t dim_m, int64_t dim_n, int64_t red, int64_t mult, bool scalar,
             bool all_taps = false, bool slab_mode = false) {
  // dim_m x dim_n: the output matrix (NT: rows x nout; TN: kout x cs per tap, `mult` = live taps of them);
  // red: length of the reduction; rows: GEMM rows of the layer.
  // Returns 0 (Small), 1 (Big: 128x128) or 2 (Big96: 128x96), whichever wastes less of its tiles.
  if (scalar) return 0;
  const int force = hf_env_big();
  if (force == 0) return 0;
  auto fill = [](int64_t d, int64_t t) { return (double)d / (double)(((d + t - 1) / t) * t); };
  const double f128 = fill(dim_m, 128) * fill(dim_n, 128), f96 = fill(dim_m, 128) * fill(dim_n, 96);
  int kind = f96 > f128 + 1e-9 ? 2 : 1;
  double fbest = kind == 2 ? f96 : f128;
  // weight gradients with every tap live and a dY channel count that is a multiple of 96: Flat96 (see its typedef)
  int64_t flat_tiles = 0;
  if (HF_CONV_FLAT96 && direction == 2 && all_taps && dim_m % 96 == 0 && dim_n % 4 == 0) {
    const double fflat = fill(mult * dim_n, 128);
    if (fflat > fbest + 0.05) { kind = 3; fbest = fflat; flat_tiles = (dim_m / 96) * ((mult * dim_n + 127) / 128); }
  }
  const bool fits = fbest >= 0.7;
  if (force > 0) return fits ? kind : 0;
  // ... and only where the launch has enough work to give ~3 workgroups per CU a K loop of >= 8 steps of 16 each
  // (a looser rule -- any reduction of >= 64 steps, splits capped at 8 steps per workgroup -- was measured to put
  // ResNet-50 layers on the 128-wide tiles and lose: 289 -> 268 matvecs/s): with a handful of steps per workgroup the 128-wide
  // tiles lose to the 64x64 ones, whose prologue / epilogue are a quarter the size (ResNet-50 on 64x64
  // images, batch 32: 260 matvecs/s with the 64x64 tiles everywhere, 228 -> 198 with the 128-wide ones wherever
  // they fit; All-CNN-C's 8192-row layers, 9 steps per workgroup: 128x96 tiles 37 us, 64x64 tiles 64 us)
  const int64_t tiles = kind == 3 ? flat_tiles
                                  : ((dim_m + 127) / 128) * ((dim_n + (kind == 2 ? 95 : 127)) / (kind == 2 ? 96 : 128)) * mult;
  const int64_t steps = (red + Big::BK - 1) / Big::BK;
  // (Flat96 weight gradients: a third of that work already pays -- All-CNN-C's stride-2 layers, 8192 / 2048 output rows,
  // ran 58 / 35 us on 36 / 72 64x64 tiles with 8 / 4 splits against MIOpen's 24 / 23 us)
  // (slab mode only: with the in-launch ticket reduction of the stand-alone calls a deep split makes the last
  // arriver's sum the longest chain of the launch -- 52 -> 87 us measured)
  const int64_t least = (kind == 3 && slab_mode) ? HF_FLAT96_MIN_WORK : HF_BIG_MIN_WORK;
  return (fits && rows >= 2048 && tiles * steps >= least) ? kind : 0;
}

// The `out_c` of a weight-gradient problem (X zero-padded to cs channels, dW has out_c <= cs of them) is applied AFTER
// setup() has chosen the tile configuration: Flat96 enumerates its output columns flat over (tap, channel), which a
// padded operand breaks unless there is a single tap (the padded columns of one row raced with the next
// row's first entries and the last row wrote past the slab) -- refused, loudly, instead.
static int apply_out_c(ConvArgs& a, int64_t out_c) {
  if (!out_c) return HF_OK;
  if (a.big == 3 && out_c != a.cs && a.ntaps != 1) return HF_ERR_ARG;
  a.out_c = (int)out_c;
  return HF_OK;
}

int64_t setup(ConvArgs& a, int direction, void* out, const void* act, const void* mat, int64_t n, int64_t h,
              int64_t w, int64_t c, int64_t k, int64_t r, int64_t s, int64_t stride_h, int64_t stride_w,
              int64_t pad_h, int64_t pad_w, int64_t act_ld, float* ws, int64_t ws_bytes, int* tickets,
              int64_t n_tickets, int target_blocks, int slab_splits = -1, int64_t slab_stride = 0,
              int64_t mat_ld = 0) {
  const int64_t oh = (h + 2 * pad_h - r) / stride_h + 1, ow = (w + 2 * pad_w - s) / stride_w + 1;
  memset(&a, 0, sizeof(a));
  a.src = (const float*)act;
  a.mat = (const float*)mat;
  a.out = (float*)out;
  a.ws = ws;
  a.tickets = tickets;
  a.R = (int)r; a.S = (int)s;
  a.stride_h = (int)stride_h; a.stride_w = (int)stride_w; a.pad_h = (int)pad_h; a.pad_w = (int)pad_w;
  a.shift_h = a.shift_w = -1;
  if ((stride_h & (stride_h - 1)) == 0 && (stride_w & (stride_w - 1)) == 0) {
    a.shift_h = a.shift_w = 0;
    while ((1 << a.shift_h) < stride_h) ++a.shift_h;
    while ((1 << a.shift_w) < stride_w) ++a.shift_w;
  }
  // taps that meet data for at least one output position
  for (int rr = 0; rr < r; ++rr) {
    bool live_r = false;
    for (int64_t y = 0; y < oh && !live_r; ++y) { const int64_t iy = y * stride_h - pad_h + rr; live_r = iy >= 0 && iy < h; }
    if (!live_r) continue;
    for (int qq = 0; qq < s; ++qq) {
      bool live_q = false;
      for (int64_t x = 0; x < ow && !live_q; ++x) { const int64_t ix = x * stride_w - pad_w + qq; live_q = ix >= 0 && ix < w; }
      if (live_q) { a.tap_rs[a.ntaps] = rr | (qq << 8); a.ntaps++; }
    }
  }
  if (a.ntaps == 0) return HF_ERR_ARG;
  int64_t rows, red_steps;
  if (direction == 0) {        // F: act = X [n,h,w,c] (pixel stride act_ld), mat = Wt [k][r*s][c], out [n,oh,ow,k]
    a.dgrad = 0; rows = n * oh * ow; a.rh = (int)oh; a.rw = (int)ow; a.sh_ = (int)h; a.sw_ = (int)w;
    a.cs = (int)c; a.nout = (int)k; a.ldc = (int)k;
  } else if (direction == 1) { // D: act = dY [n,oh,ow,k], mat = WT [c][r*s][k], out = dX [n,h,w,c]
    a.dgrad = 1; rows = n * h * w; a.rh = (int)h; a.rw = (int)w; a.sh_ = (int)oh; a.sw_ = (int)ow;
    a.cs = (int)k; a.nout = (int)c; a.ldc = (int)c;
  } else {                     // W: act = X [n,h,w,c], mat = dY [n*oh*ow][k], out = dW [k][r*s][c]
    a.dgrad = 0; rows = n * oh * ow; a.rh = (int)oh; a.rw = (int)ow; a.sh_ = (int)h; a.sw_ = (int)w;
    a.cs = (int)c; a.kout = (int)k;
  }
  a.out_c = a.cs;
  a.rows = (int)rows;
  a.cs_ld = (int)(act_ld > 0 ? act_ld : a.cs);
  a.mat_ld = (int)(mat_ld > 0 ? mat_ld : a.cs);
  a.scalar = ((c % 4) || (k % 4) || (a.cs_ld % 4) || (a.mat_ld % 4)) ? 1 : 0;
  if (a.cs_ld < a.cs || a.mat_ld < a.cs || (mat_ld > 0 && direction == 2)) return HF_ERR_ARG;
  // strided data gradient in slab mode: residue classes of the input pixel (see ConvArgs)
  int cls_taps_max = a.ntaps;
  // (large maps only: on ResNet-18's <= 1568-row layers the launch is latency-bound either way and the extra
  // partially-filled row tiles cost ~1 % of the bench)
  const bool classes = direction == 1 && slab_splits >= 0 && (stride_h > 1 || stride_w > 1) &&
                       stride_h * stride_w <= 4 && !a.scalar && n * h * w >= 2048 && hf_env_dclass();
  struct { int py, px, hc, wc, ntaps; unsigned char r[MAX_TAPS], q[MAX_TAPS]; } cl[4];
  int ncls = 0;
  if (classes) {
    cls_taps_max = 0;
    for (int py = 0; py < stride_h; ++py)
      for (int px = 0; px < stride_w; ++px) {
        const int hc = (int)((h - py + stride_h - 1) / stride_h), wc = (int)((w - px + stride_w - 1) / stride_w);
        if (hc <= 0 || wc <= 0) continue;
        auto& c_ = cl[ncls];
        c_.py = py; c_.px = px; c_.hc = hc; c_.wc = wc; c_.ntaps = 0;
        for (int t_ = 0; t_ < a.ntaps; ++t_) {
          const int rr = a.tap_rs[t_] & 255, qq = a.tap_rs[t_] >> 8;
          const int64_t dy = py + pad_h - rr, dx = px + pad_w - qq;
          if (((dy % stride_h) + stride_h) % stride_h == 0 && ((dx % stride_w) + stride_w) % stride_w == 0) {
            c_.r[c_.ntaps] = (unsigned char)rr; c_.q[c_.ntaps] = (unsigned char)qq; c_.ntaps++;
          }
        }
        if (c_.ntaps > cls_taps_max) cls_taps_max = c_.ntaps;
        ++ncls;
      }
  }
  a.big = direction <= 1 ? want_big(direction, rows, rows, a.nout, (int64_t)cls_taps_max * a.cs, 1, a.scalar)
                         : want_big(direction, rows, a.kout, a.cs, rows, a.ntaps, a.scalar, a.ntaps == (int)(r * s),
                                    slab_splits >= 0);
  const int BM = a.big == 3 ? Flat96::BM : a.big ? Big::BM : Small::BM;
  const int BN = a.big == 2 ? Big96::BN : a.big ? Big::BN : Small::BN;
  const int BK = a.big ? Big::BK : Small::BK;
  if (direction <= 1) {
    a.tiles_m = (int)((rows + BM - 1) / BM);
    a.tiles_n = (a.nout + BN - 1) / BN;
    red_steps = (int64_t)cls_taps_max * ((a.cs + BK - 1) / BK);
    if (ncls > 0) {  // rows class by class, taps reordered class-major
      a.ncls = ncls;
      int tile0 = 0, tap0 = 0;
      for (int i = 0; i < ncls; ++i) {
        a.cls_tile0[i] = tile0; a.cls_tap0[i] = tap0;
        a.cls_h[i] = cl[i].hc; a.cls_w[i] = cl[i].wc; a.cls_py[i] = cl[i].py; a.cls_px[i] = cl[i].px;
        for (int t_ = 0; t_ < cl[i].ntaps; ++t_) a.tap_rs[tap0 + t_] = cl[i].r[t_] | (cl[i].q[t_] << 8);
        tap0 += cl[i].ntaps;
        tile0 += (int)((n * cl[i].hc * cl[i].wc + BM - 1) / BM);
      }
      a.cls_tile0[ncls] = tile0; a.cls_tap0[ncls] = tap0;
      a.tiles_m = tile0;
      if (red_steps < 1) red_steps = 1;
    }
  } else {
    a.tiles_m = (a.kout + BM - 1) / BM;
    a.tiles_n = a.big == 3 ? (a.ntaps * a.cs + BN - 1) / BN : a.ntaps * ((a.cs + BN - 1) / BN);
    red_steps = (rows + BK - 1) / BK;
  }
  a.steps = (int)red_steps;
  const int64_t tiles = (int64_t)a.tiles_m * a.tiles_n;
  if (slab_splits >= 0) {  // consumer-side reduction: no workspace, no tickets
    a.slabs = 1;
    a.slab_stride = slab_stride;
    int sp = slab_splits > 0 ? slab_splits : choose_splits(tiles, red_steps, target_blocks, 0, true, BK);
    // strided data gradients by residue class: the classes' K loops differ by up to 4x (1 ... 4 taps of a 3x3 / stride 2
    // kernel) and a uniform split leaves the short ones with 2 steps per workgroup -- with one workgroup per CU already
    // there, do not split (A/B knob HF_DCLASS_NOSPLIT)
    if (HF_DCLASS_NOSPLIT && slab_splits == 0 && ncls > 0 && a.big && tiles >= 256) sp = 1;
    if (sp > red_steps) sp = (int)red_steps;
    while (sp > 1 && ((red_steps + sp - 1) / sp) * (sp - 1) >= red_steps) --sp;
    a.splits = sp < 1 ? 1 : sp;
    return tiles * a.splits;
  }
  // more output tiles than ticket counters (large batches / maps): such a launch fills the chip
  // without a K split, and an unsplit launch draws no tickets
  a.splits = tiles > n_tickets ? 1 : choose_splits(tiles, red_steps, target_blocks, ws_bytes, false, BK,
                                                    (int64_t)BM * BN);
  return tiles * a.splits;
}

int check_common(const void* out, const void* act, const void* mat, const void* workspace, const void* tickets,
                 int dtype, int64_t n, int64_t h, int64_t w, int64_t c, int64_t k, int64_t r, int64_t s,
                 int64_t stride_h, int64_t stride_w, int64_t pad_h, int64_t pad_w) {
  if (!out || !act || !mat || !workspace || !tickets) return HF_ERR_ARG;
  if (dtype != HF_F32) return HF_ERR_ARG;
  if (n <= 0 || h <= 0 || w <= 0 || c <= 0 || k <= 0 || r <= 0 || s <= 0 || stride_h <= 0 ||
      stride_w <= 0 || pad_h < 0 || pad_w < 0 || r * s > MAX_TAPS)
    return HF_ERR_ARG;
  const int64_t oh = (h + 2 * pad_h - r) / stride_h + 1, ow = (w + 2 * pad_w - s) / stride_w + 1;
  if (oh <= 0 || ow <= 0) return HF_ERR_ARG;
  // (channel counts that are not multiples of 4 run the element-wise gather variant)
  if (!aligned16(workspace) || (((c % 4) == 0 && (k % 4) == 0) &&
                                (!aligned16(out) || !aligned16(act) || !aligned16(mat))))
    return HF_ERR_ALIGN;
  if (n * h * w * (c > k ? c : k) > 0x7fffffffLL || n * oh * ow * (c > k ? c : k) > 0x7fffffffLL ||
      k * r * s * c > 0x7fffffffLL)
    return HF_ERR_ARG;
  return HF_OK;
}

}  // namespace

// C linkage comes from hf_pcg.h
int hf_conv2d_nhwc(int direction, void* out, const void* act, const void* mat, int64_t n,
                   int64_t h, int64_t w, int64_t c, int64_t k, int64_t r, int64_t s,
                   int64_t stride_h, int64_t stride_w, int64_t pad_h, int64_t pad_w,
                   int64_t act_ld, void* workspace, int64_t workspace_bytes, void* tickets,
                   int64_t n_tickets, int target_blocks, int dtype, void* stream) {
  if (direction < 0 || direction > 2) return HF_ERR_ARG;
  const int rc = check_common(out, act, mat, workspace, tickets, dtype, n, h, w, c, k, r, s, stride_h,
                              stride_w, pad_h, pad_w);
  if (rc) return rc;
  ConvArgs a;
  const int64_t blocks = setup(a, direction, out, act, mat, n, h, w, c, k, r, s, stride_h, stride_w, pad_h,
                               pad_w, act_ld, (float*)workspace, workspace_bytes, (int*)tickets, n_tickets,
                               target_blocks);
  if (blocks <= 0) return (int)blocks;
  launch_one(direction, a, blocks, (hipStream_t)stream);
  HF_HIP(hipGetLastError());
  return HF_OK;
}

int hf_conv2d_nhwc_backward(void* dx, void* dw, const void* dy, const void* x, const void* w_t, int64_t n,
                            int64_t h, int64_t w, int64_t c, int64_t k, int64_t r, int64_t s,
                            int64_t stride_h, int64_t stride_w, int64_t pad_h, int64_t pad_w,
                            void* workspace, int64_t workspace_bytes, void* tickets, int64_t n_tickets,
                            int target_blocks, int dtype, void* stream) {
  if (!dx || !dw) return HF_ERR_ARG;
  int rc = check_common(dx, dy, w_t, workspace, tickets, dtype, n, h, w, c, k, r, s, stride_h, stride_w,
                        pad_h, pad_w);
  if (rc) return rc;
  if (!x || !aligned16(x) || !aligned16(dw)) return x ? HF_ERR_ALIGN : HF_ERR_ARG;
  // the two halves get disjoint halves of the scratch (they run concurrently)
  const int64_t half_ws = (workspace_bytes / 2) & ~(int64_t)15, half_t = n_tickets / 2;
  ConvArgs d, g;
  const int64_t bd = setup(d, 1, dx, dy, w_t, n, h, w, c, k, r, s, stride_h, stride_w, pad_h, pad_w, 0,
                           (float*)workspace, half_ws, (int*)tickets, half_t, target_blocks);
  if (bd <= 0) return (int)bd;
  const int64_t bw = setup(g, 2, dw, x, dy, n, h, w, c, k, r, s, stride_h, stride_w, pad_h, pad_w, 0,
                           (float*)((char*)workspace + half_ws), half_ws, (int*)tickets + half_t, half_t,
                           target_blocks);
  if (bw <= 0) return (int)bw;
  if (d.scalar || g.scalar) return HF_ERR_ARG;  // the merged launch has the 16-byte gather variant only
  launch_dw(d, g, bd, bw, (hipStream_t)stream);
  HF_HIP(hipGetLastError());
  return HF_OK;
}

// ---- consumer-side reduction ("slab") variants ---------------------------------------
// Split s writes its partial result, in the output's own layout, to out + s*slab_stride and
// the launch ends there: the kernel that consumes the tensor sums the `splits` slabs in its
// prologue (hf_chan_affine_ex, hf_bn_adjoint_pre, hf_pack_ex).  No workspace, no tickets, no
// fences: the launch boundary publishes the slabs.
int hf_conv2d_nhwc_plan(int direction, int64_t n, int64_t h, int64_t w, int64_t c, int64_t k, int64_t r,
                        int64_t s, int64_t stride_h, int64_t stride_w, int64_t pad_h, int64_t pad_w,
                        int target_blocks) {
  if (direction < 0 || direction > 2) return HF_ERR_ARG;
  if (n <= 0 || h <= 0 || w <= 0 || c <= 0 || k <= 0 || r <= 0 || s <= 0 || stride_h <= 0 || stride_w <= 0 ||
      pad_h < 0 || pad_w < 0 || r * s > MAX_TAPS)
    return HF_ERR_ARG;
  ConvArgs a;
  float dummy = 0.f;
  const int64_t blocks = setup(a, direction, &dummy, &dummy, &dummy, n, h, w, c, k, r, s, stride_h, stride_w,
                               pad_h, pad_w, 0, nullptr, 0, nullptr, 0, target_blocks, 0, 0);
  if (blocks <= 0) return (int)blocks;
  return a.splits;
}

int hf_conv2d_nhwc_slabs(int direction, void* out, const void* act, const void* mat, int64_t n, int64_t h,
                         int64_t w, int64_t c, int64_t k, int64_t r, int64_t s, int64_t stride_h,
                         int64_t stride_w, int64_t pad_h, int64_t pad_w, int64_t act_ld, int64_t mat_ld,
                         int64_t out_c, int splits, int64_t slab_stride, int dtype, void* stream) {
  if (direction < 0 || direction > 2 || splits < 1 || slab_stride < 0 || mat_ld < 0) return HF_ERR_ARG;
  if (out_c < 0 || out_c > c || (out_c && direction != 2)) return HF_ERR_ARG;
  alignas(16) float dummy_ws[4];
  const int rc = check_common(out, act, mat, dummy_ws, dummy_ws, dtype, n, h, w, c, k, r, s, stride_h, stride_w,
                              pad_h, pad_w);
  if (rc) return rc;
  ConvArgs a;
  const int64_t blocks = setup(a, direction, out, act, mat, n, h, w, c, k, r, s, stride_h, stride_w, pad_h,
                               pad_w, act_ld, nullptr, 0, nullptr, 0, 0, splits, slab_stride, mat_ld);
  if (blocks <= 0) return (int)blocks;
  if (a.splits != splits) return HF_ERR_ARG;  // ask hf_conv2d_nhwc_plan first
  if (apply_out_c(a, out_c)) return HF_ERR_ARG;
  launch_one(direction, a, blocks, (hipStream_t)stream);
  HF_HIP(hipGetLastError());
  return HF_OK;
}

int hf_conv2d_nhwc_slabs_unpack(void* out, const void* act, const void* mat, int64_t n, int64_t h, int64_t w,
                                int64_t c, int64_t k, int64_t r, int64_t s, int64_t stride_h, int64_t stride_w,
                                int64_t pad_h, int64_t pad_w, int64_t act_ld, int64_t mat_ld, int splits,
                                int64_t slab_stride, const void* usrc, void* const* udsts, const int64_t* usrc_offs,
                                const int64_t* unumels, const int64_t* uslabs, const int64_t* uinners,
                                const int64_t* ulive, const int64_t* uhalves, int n_tensors, int dtype,
                                void* stream) {
  if (splits < 1 || slab_stride < 0 || mat_ld < 0 || dtype != HF_F32) return HF_ERR_ARG;
  if (!usrc || !udsts || !usrc_offs || !unumels || !uslabs || !uinners || n_tensors < 1) return HF_ERR_ARG;
  if (uhalves)  // (transposed copies run as LDS-tiled workgroups of hf_unpack_weights' own kernel only)
    for (int t_ = 0; t_ < n_tensors; ++t_)
      if (uhalves[t_] == 2) return HF_ERR_ARG;
  alignas(16) float dummy_ws[4];
  const int rc = check_common(out, act, mat, dummy_ws, dummy_ws, dtype, n, h, w, c, k, r, s, stride_h, stride_w,
                              pad_h, pad_w);
  if (rc) return rc;
  ConvArgs a;
  const int64_t blocks = setup(a, 0, out, act, mat, n, h, w, c, k, r, s, stride_h, stride_w, pad_h, pad_w, act_ld,
                               nullptr, 0, nullptr, 0, 0, splits, slab_stride, mat_ld);
  if (blocks <= 0) return (int)blocks;
  if (a.splits != splits || a.big || a.ncls > 0) return HF_ERR_ARG;  // (small-map launches only: ask the plan first)
  hf_shared::UnpackArgs u;
  int ublocks = 0;
  const int next = hf_shared::fill_unpack_args<float>(u, &ublocks, 0, udsts, usrc_offs, unumels, uslabs, uinners,
                                                      ulive, uhalves, n_tensors, /*allow_transposed=*/false);
  if (next < 0) return next;
  if (next != n_tensors || ublocks < 1) return HF_ERR_ARG;  // more tensors than one argument block holds
  seal(a, 0);
  const dim3 grid((unsigned)(blocks + ublocks)), block(CT);
  if (a.scalar)
    hipLaunchKernelGGL((k_conv_nt_unpack<true, Small>), grid, block, 0, (hipStream_t)stream, a, u, (const float*)usrc,
                       (int)blocks);
  else
    hipLaunchKernelGGL((k_conv_nt_unpack<false, Small>), grid, block, 0, (hipStream_t)stream, a, u,
                       (const float*)usrc, (int)blocks);
  HF_HIP(hipGetLastError());
  return HF_OK;
}

int hf_conv2d_nhwc_backward_slabs(void* dx, void* dw, const void* dy, const void* x, const void* w_t,
                                  int64_t n, int64_t h, int64_t w, int64_t c, int64_t k, int64_t r,
                                  int64_t s, int64_t stride_h, int64_t stride_w, int64_t pad_h,
                                  int64_t pad_w, int splits_d, int64_t slab_stride_d, int splits_w,
                                  int64_t slab_stride_w, int dtype, void* stream) {
  if (!dx || !dw || !x || splits_d < 1 || splits_w < 1) return HF_ERR_ARG;
  alignas(16) float dummy_ws[4];
  int rc = check_common(dx, dy, w_t, dummy_ws, dummy_ws, dtype, n, h, w, c, k, r, s, stride_h, stride_w, pad_h,
                        pad_w);
  if (rc) return rc;
  if (!aligned16(x) || !aligned16(dw)) return HF_ERR_ALIGN;
  ConvArgs d, g;
  const int64_t bd = setup(d, 1, dx, dy, w_t, n, h, w, c, k, r, s, stride_h, stride_w, pad_h, pad_w, 0, nullptr,
                           0, nullptr, 0, 0, splits_d, slab_stride_d);
  if (bd <= 0) return (int)bd;
  const int64_t bw = setup(g, 2, dw, x, dy, n, h, w, c, k, r, s, stride_h, stride_w, pad_h, pad_w, 0, nullptr, 0,
                           nullptr, 0, 0, splits_w, slab_stride_w);
  if (bw <= 0) return (int)bw;
  if (d.splits != splits_d || g.splits != splits_w) return HF_ERR_ARG;
  if (d.scalar || g.scalar) return HF_ERR_ARG;  // the merged launch has the 16-byte gather variant only
  launch_dw(d, g, bd, bw, (hipStream_t)stream);
  HF_HIP(hipGetLastError());
  return HF_OK;
}

int hf_conv2d_nhwc_dw_slabs(const hf_conv_problem* d, const hf_conv_problem* w, int dtype, void* stream) {
  if (!d || !w || d->direction > 1 || d->direction < 0 || w->direction != 2) return HF_ERR_ARG;
  ConvArgs a[2];
  int64_t blocks[2];
  alignas(16) float dummy_ws[4];
  const hf_conv_problem* pr[2] = {d, w};
  for (int i = 0; i < 2; ++i) {
    const hf_conv_problem& q = *pr[i];
    if (q.splits < 1 || q.slab_stride < 0 || q.mat_ld < 0 || q.out_c < 0 || q.out_c > q.c || (q.out_c && i == 0))
      return HF_ERR_ARG;
    const int rc = check_common(q.out, q.act, q.mat, dummy_ws, dummy_ws, dtype, q.n, q.h, q.w, q.c, q.k, q.r, q.s,
                                q.stride_h, q.stride_w, q.pad_h, q.pad_w);
    if (rc) return rc;
    blocks[i] = setup(a[i], q.direction, q.out, q.act, q.mat, q.n, q.h, q.w, q.c, q.k, q.r, q.s, q.stride_h,
                      q.stride_w, q.pad_h, q.pad_w, q.act_ld, nullptr, 0, nullptr, 0, 0, q.splits, q.slab_stride,
                      q.mat_ld);
    if (blocks[i] <= 0) return (int)blocks[i];
    if (a[i].splits != q.splits || a[i].scalar) return HF_ERR_ARG;
    if (apply_out_c(a[i], q.out_c)) return HF_ERR_ARG;
  }
  const dim3 grid((unsigned)(blocks[0] + blocks[1]));
  (void)grid;
  launch_dw(a[0], a[1], blocks[0], blocks[1], (hipStream_t)stream);
  HF_HIP(hipGetLastError());
  return HF_OK;
}

// Tangent convolutions (direction 0, slab mode, 64x64 tiles) whose epilogue also writes the per-channel partial sums
// a TRAIN-mode BatchNorm behind them needs (ConvArgs::bn_*): one launch for up to GROUP_MAX problems, each with or
// without sums.  HF_ERR_ARG for anything the BNSUM instantiations do not cover -- the caller then issues the plain
// launch and the reduction launch.
int hf_conv2d_nhwc_group_slabs_bnsum(const hf_conv_problem* problems, int n_problems, const hf_conv_bnsum* sums,
                                     int dtype, void* stream) {
  if (!problems || !sums || n_problems < 1 || n_problems > GROUP_MAX) return HF_ERR_ARG;
  GroupArgs q;
  memset(&q, 0, sizeof(q));
  int64_t total = 0;
  alignas(16) float dummy_ws[4];
  for (int i = 0; i < n_problems; ++i) {
    const hf_conv_problem& pr = problems[i];
    if (pr.direction != 0 || pr.splits < 1 || pr.slab_stride < 0 || pr.mat_ld < 0 || pr.out_c) return HF_ERR_ARG;
    const int rc = check_common(pr.out, pr.act, pr.mat, dummy_ws, dummy_ws, dtype, pr.n, pr.h, pr.w, pr.c, pr.k,
                                pr.r, pr.s, pr.stride_h, pr.stride_w, pr.pad_h, pr.pad_w);
    if (rc) return rc;
    ConvArgs& a = q.a[i];
    const int64_t blocks = setup(a, 0, pr.out, pr.act, pr.mat, pr.n, pr.h, pr.w, pr.c, pr.k, pr.r, pr.s,
                                 pr.stride_h, pr.stride_w, pr.pad_h, pr.pad_w, pr.act_ld, nullptr, 0, nullptr, 0, 0,
                                 pr.splits, pr.slab_stride, pr.mat_ld);
    if (blocks <= 0) return (int)blocks;
    if (a.splits != pr.splits || a.scalar || a.big || a.ncls > 0 || !a.slabs) return HF_ERR_ARG;
    const hf_conv_bnsum& b = sums[i];
    if (b.part_1) {
      if (!b.part_x || !b.x || !b.mean || !b.rstd) return HF_ERR_ARG;
      if (b.part_rows != (int64_t)a.tiles_m * a.splits) return HF_ERR_ARG;  // (= ceil(rows / 64) * splits)
      a.bn_x = (const float*)b.x; a.bn_mean = (const float*)b.mean; a.bn_rstd = (const float*)b.rstd;
      a.bn_part_x = (float*)b.part_x; a.bn_part_1 = (float*)b.part_1;
    }
    q.start[i] = (int)total;
    total += blocks;
    if (total > 0x7fffffffLL) return HF_ERR_ARG;
  }
  GroupMeta m;
  memset(&m, 0, sizeof(m));
  for (int i = 0; i < GROUP_MAX; ++i) m.start[i] = i < n_problems ? q.start[i] : (int)total;
  m.start[GROUP_MAX] = (int)total;
  m.n = n_problems;
  for (int i = 0; i < n_problems; ++i) seal(q.a[i], 0);
  const dim3 grid((unsigned)total), block(CT);
  hipStream_t st = (hipStream_t)stream;
  if (n_problems == 1 && q.a[0].bn_part_1)
    hipLaunchKernelGGL((k_conv_nt<false, Small, false, true>), grid, block, 0, st, q.a[0]);
  else
    hipLaunchKernelGGL((k_conv_group<false, false, true>), grid, block, 0, st, q.a[0], q.a[1], q.a[2], q.a[3], m);
  HF_HIP(hipGetLastError());
  return HF_OK;
}

int hf_conv2d_nhwc_group_slabs(const hf_conv_problem* problems, int n_problems, int dtype, void* stream) {
  if (!problems || n_problems < 1 || n_problems > GROUP_MAX) return HF_ERR_ARG;
  GroupArgs q;  // (~1.2 KB, passed to the kernel by value)
  memset(&q, 0, sizeof(q));
  int64_t total = 0;
  alignas(16) float dummy_ws[4];
  for (int i = 0; i < n_problems; ++i) {
    const hf_conv_problem& pr = problems[i];
    if (pr.direction < 0 || pr.direction > 2 || pr.splits < 1 || pr.slab_stride < 0 || pr.mat_ld < 0) return HF_ERR_ARG;
    if (pr.out_c < 0 || pr.out_c > pr.c || (pr.out_c && pr.direction != 2)) return HF_ERR_ARG;
    const int rc = check_common(pr.out, pr.act, pr.mat, dummy_ws, dummy_ws, dtype, pr.n, pr.h, pr.w, pr.c, pr.k,
                                pr.r, pr.s, pr.stride_h, pr.stride_w, pr.pad_h, pr.pad_w);
    if (rc) return rc;
    const int64_t blocks = setup(q.a[i], pr.direction, pr.out, pr.act, pr.mat, pr.n, pr.h, pr.w, pr.c, pr.k, pr.r,
                                 pr.s, pr.stride_h, pr.stride_w, pr.pad_h, pr.pad_w, pr.act_ld, nullptr, 0,
                                 nullptr, 0, 0, pr.splits, pr.slab_stride, pr.mat_ld);
    if (blocks <= 0) return (int)blocks;
    if (q.a[i].splits != pr.splits) return HF_ERR_ARG;  // ask hf_conv2d_nhwc_plan first
    if (q.a[i].scalar) return HF_ERR_ARG;               // the grouped launch has the 16-byte gather variant only
    if (apply_out_c(q.a[i], pr.out_c)) return HF_ERR_ARG;
    q.tn[i] = pr.direction == 2;
    q.start[i] = (int)total;
    total += blocks;
    if (total > 0x7fffffffLL) return HF_ERR_ARG;
  }
  q.start[n_problems] = (int)total;
  q.n = n_problems;
  bool anybig = false;
  for (int i = 0; i < n_problems; ++i) anybig = anybig || q.a[i].big;
  GroupMeta m;
  memset(&m, 0, sizeof(m));
  for (int i = 0; i < GROUP_MAX; ++i) {
    m.tn[i] = q.tn[i];
    m.start[i] = i <= n_problems ? q.start[i] : (int)total;  // (unused slots: empty ranges at the end)
  }
  m.start[GROUP_MAX] = (int)total;
  m.n = n_problems;
  bool anycls = false;
  for (int i = 0; i < n_problems; ++i) anycls = anycls || q.a[i].ncls > 0;
  for (int i = 0; i < n_problems; ++i) seal(q.a[i], 0);
  const dim3 grid((unsigned)total), block(CT);
  hipStream_t st = (hipStream_t)stream;
  if (anybig && anycls) hipLaunchKernelGGL((k_conv_group<true, true>), grid, block, 0, st, q.a[0], q.a[1], q.a[2], q.a[3], m);
  else if (anybig) hipLaunchKernelGGL((k_conv_group<true, false>), grid, block, 0, st, q.a[0], q.a[1], q.a[2], q.a[3], m);
  else if (anycls) hipLaunchKernelGGL((k_conv_group<false, true>), grid, block, 0, st, q.a[0], q.a[1], q.a[2], q.a[3], m);
  else hipLaunchKernelGGL((k_conv_group<false, false>), grid, block, 0, st, q.a[0], q.a[1], q.a[2], q.a[3], m);
  HF_HIP(hipGetLastError());
  return HF_OK;
}
