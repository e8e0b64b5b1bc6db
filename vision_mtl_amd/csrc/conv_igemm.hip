// Implicit-GEMM convolution on exact-fp32 MFMA (v_mfma_f32_32x32x2_f32), NHWC.
//
// Replaces the ATen conv2d / conv_transpose2d dispatches reached from
//   reference vision_mtl/utils/model_utils.py:71,74 (DoubleConv 3x3),
//   reference vision_mtl/models/mtan_model.py:31-46,105-129,214-216,369 (1x1, 3x3, ConvT),
//   smp UnetDecoder Conv2dReLU / SegmentationHead and timm pointwise convs
//   (reference vision_mtl/utils/model_utils.py:25-34, models/basic_model.py:30-41).
//
// One kernel serves forward and data-gradient: both are "gather rows of an
// NHWC tensor per filter tap, contract against a packed [row][tap*Cs+c] weight
// matrix".  dgrad of a stride-1 conv is the same contraction over dY with the
// tap-flipped, transposed packing (see pack.hip).  A second kernel computes the
// weight gradient as a split-K GEMM over pixels.
//
// GEMM view (forward):  Y[m][n] = sum_kk  Xcol[m][kk] * Wp[n][kk]
//   m  = (b, ho, wo)            M    = B*Ho*Wo
//   kk = tap*Cs + ci            Ktot = KH*KW*Cs   (Cs % 4 == 0, pad channels are 0)
//   n  = output channel         rows n >= Nw of Wp are treated as 0
//
// LDS tiles are [row][BK + 4] with kk contiguous, so one ds_read_b128 gives a
// lane 4 consecutive kk; lanes 0-31 take kk 0..3 and lanes 32-63 kk 4..7 of each
// 8-wide k-group and the 4 elements feed 4 MFMAs (k pairs (j, j+4)).  Row stride
// 36 floats makes those reads conflict-free (9*i mod 16 is a bijection).
#include "common.h"

#define BK 32
#define LDT (BK + 4)

struct ConvP {
  const float* x;     // [B][H][W][Cs]
  const float* wp;    // [Nw][Ktot]
  const float* bias;  // [Nw] or nullptr
  float* y;           // [B][Ho][Wo][ldy]  (or pixel-shuffled, see shuffle)
  float* stats;       // optional [gridM][2][ldy] per-row-block column mean / M2 (or nullptr)
  int B, H, W, Cs;
  int Ho, Wo, ldy;
  int Nw;             // valid weight rows (Cout, or 4*Cout for shuffle)
  int Cout;           // logical channels written non-zero per output pixel
  int KH, KW, stride, pad;
  int Ktot, M;
  int act;
  int shuffle;        // 1: rows n=(u*2+v)*Cout+co are scattered to (2h+u, 2w+v, co)
  int tiles_m, tiles_n;
};

template <int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvP p) {
  constexpr int TM = BM / WAVES_M / 32;
  constexpr int TN = BN / WAVES_N / 32;
  constexpr int RA = BM / 32;  // A rows per thread
  constexpr int RB = BN / 32;  // B rows per thread
  static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                    // [2][BM][LDT]
  float* Bs = smem + 2 * BM * LDT;     // [2][BN][LDT]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = tid >> 6;
  const int wm = wv / WAVES_N, wn = wv % WAVES_N;
  const int l31 = lane & 31, hi = lane >> 5;

  const int nwg = p.tiles_m * p.tiles_n;
  const int bid = xcd_remap(blockIdx.x, nwg);
  const int tile_m = bid / p.tiles_n, tile_n = bid % p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  // ---- per-thread loader geometry ----
  const int k4 = tid & 7;   // float4 column inside the BK chunk
  const int r0 = tid >> 3;  // 0..31
  int pixbase[RA], hb[RA], wb[RA];
#pragma unroll
  for (int i = 0; i < RA; ++i) {
    const int m = m0 + r0 + 32 * i;
    if (m < p.M) {
      const int hw = p.Ho * p.Wo;
      const int b = m / hw;
      const int rem = m - b * hw;
      const int ho = rem / p.Wo;
      const int wo = rem - ho * p.Wo;
      pixbase[i] = b * p.H * p.W;
      hb[i] = ho * p.stride - p.pad;
      wb[i] = wo * p.stride - p.pad;
    } else {
      pixbase[i] = 0;
      hb[i] = -(1 << 20);  // forces the bounds test to fail
      wb[i] = 0;
    }
  }
  // running (tap, ci) of this thread's float4 column
  int kk = k4 * 4;
  int tap = kk / p.Cs;
  int ci = kk - tap * p.Cs;
  int dh = tap / p.KW;
  int dw = tap - dh * p.KW;

  f32x4 ra[RA], rb[RB];

  auto load_tile = [&]() {
    const bool kok = kk < p.Ktot;
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      const int h = hb[i] + dh, w = wb[i] + dw;
      const bool ok = kok && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (ok) {
        const size_t off = (size_t)(pixbase[i] + h * p.W + w) * p.Cs + ci;
        v = *reinterpret_cast<const f32x4*>(p.x + off);
      }
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const int n = n0 + r0 + 32 * i;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (kok && n < p.Nw) v = *reinterpret_cast<const f32x4*>(p.wp + (size_t)n * p.Ktot + kk);
      rb[i] = v;
    }
    // advance to the next BK chunk
    kk += BK;
    ci += BK;
    while (ci >= p.Cs) {
      ci -= p.Cs;
      ++dw;
      if (dw == p.KW) { dw = 0; ++dh; }
    }
  };
  auto store_tile = [&](int buf) {
    float* a = As + buf * BM * LDT;
    float* b = Bs + buf * BN * LDT;
#pragma unroll
    for (int i = 0; i < RA; ++i) *reinterpret_cast<f32x4*>(a + (r0 + 32 * i) * LDT + k4 * 4) = ra[i];
#pragma unroll
    for (int i = 0; i < RB; ++i) *reinterpret_cast<f32x4*>(b + (r0 + 32 * i) * LDT + k4 * 4) = rb[i];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = (p.Ktot + BK - 1) / BK;
  load_tile();
  store_tile(0);
  __syncthreads();

  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    const bool more = kt + 1 < nk;
    if (more) load_tile();  // global loads stay in flight under the MFMAs
    const float* a = As + cur * BM * LDT + (wm * TM * 32 + l31) * LDT + hi * 4;
    const float* b = Bs + cur * BN * LDT + (wn * TN * 32 + l31) * LDT + hi * 4;
#pragma unroll
    for (int kg = 0; kg < BK / 8; ++kg) {
      f32x4 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const f32x4*>(a + i * 32 * LDT + kg * 8);
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const f32x4*>(b + j * 32 * LDT + kg * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][e], fb[j][e], acc[i][j], 0, 0, 0);
    }
    if (more) store_tile(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

  // ---- epilogue: bias + activation, zero the pad channels, store ----
  const int hw = p.Ho * p.Wo;
  float bv[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + (wn * TN + j) * 32 + l31;
    bv[j] = (p.bias != nullptr && n < p.Nw) ? p.bias[p.shuffle ? n % p.Cout : n] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (m >= p.M) continue;
        float v = act_fwd(acc[i][j][r] + bv[j], p.act);
        if (!p.shuffle) {
          if (n < p.ldy) {
            if (n >= p.Cout) v = 0.f;
            p.y[(size_t)m * p.ldy + n] = v;
          }
        } else if (n < p.Nw) {
          const int q = n / p.Cout, co = n - q * p.Cout;
          const int b_ = m / hw, rem = m - b_ * hw;
          const int h_ = rem / p.Wo, w_ = rem - h_ * p.Wo;
          const size_t o =
              ((size_t)(b_ * 2 * p.Ho + 2 * h_ + (q >> 1)) * (2 * p.Wo) + 2 * w_ + (q & 1)) * p.ldy + co;
          p.y[o] = v;
        }
      }
    }
  }

  // ---- BatchNorm partials of this row block, straight from the accumulators: per column the
  // block-local MEAN and M2 = sum (v - mean)^2 (two in-register passes).  The finalize kernel
  // merges blocks with Chan's parallel-variance formula in fp64, so the variance never goes
  // through E[x^2] - E[x]^2 (which loses everything when |mean| >> std).
  if (p.stats != nullptr) {
    float* red = smem;  // [2][WAVES_M][BN]; the staging tiles are dead after the last barrier
    const int nvalid = min(BM, p.M - m0);
    auto colval = [&](int i, int j, int r, float& v) -> bool {
      const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      const int n = n0 + (wn * TN + j) * 32 + l31;
      v = (n < p.Cout) ? act_fwd(acc[i][j][r] + bv[j], p.act) : 0.f;
      return m < p.M;
    };
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      float s1 = 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v;
          if (colval(i, j, r, v)) s1 += v;
        }
      s1 += __shfl_xor(s1, 32, 64);
      if (hi == 0) red[wm * BN + (wn * TN + j) * 32 + l31] = s1;
    }
    __syncthreads();
    float mean[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < WAVES_M; ++w) s += red[w * BN + (wn * TN + j) * 32 + l31];
      mean[j] = s / (float)nvalid;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      float s2 = 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v;
          if (colval(i, j, r, v)) s2 += (v - mean[j]) * (v - mean[j]);
        }
      s2 += __shfl_xor(s2, 32, 64);
      if (hi == 0) red[(WAVES_M + wm) * BN + (wn * TN + j) * 32 + l31] = s2;
    }
    __syncthreads();
    for (int c = tid; c < BN; c += 256) {
      float s = 0.f, m2 = 0.f;
#pragma unroll
      for (int w = 0; w < WAVES_M; ++w) {
        s += red[w * BN + c];
        m2 += red[(WAVES_M + w) * BN + c];
      }
      const int n = n0 + c;
      if (n < p.ldy) {
        p.stats[((size_t)tile_m * 2 + 0) * p.ldy + n] = s / (float)nvalid;
        p.stats[((size_t)tile_m * 2 + 1) * p.ldy + n] = m2;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// weight gradient:  dWp[n][kk] += sum_m dY[m][n] * Xcol[m][kk]   (split over m)
// A operand = dY rows (i = n), B operand = im2col(X) (j = kk), k = pixel.
// LDS tiles are [pixel][channel] exactly as they sit in HBM, so fragment reads
// are conflict-free ds_read_b32 (32 consecutive floats per half-wave).
// ---------------------------------------------------------------------------
struct WgradP {
  const float* x;   // [B][H][W][Cs]
  const float* dy;  // [B][Ho][Wo][ldy]
  float* dwp;       // [Nw][Ktot], zeroed by the caller (this launch accumulates atomically)
  int B, H, W, Cs;
  int Ho, Wo, ldy;
  int Nw;
  int KH, KW, stride, pad;
  int Ktot, M;
  int chunk;        // pixels per z-slice (multiple of BP)
};

#define BP 32

template <int BMC, int BNK, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradP p) {
  constexpr int TM = BMC / WAVES_M / 32;
  constexpr int TN = BNK / WAVES_N / 32;
  constexpr int LDY = BMC + 4;
  constexpr int LDX = BNK + 4;
  constexpr int YQ = BMC / 4;              // float4 per dY row
  constexpr int XQ = BNK / 4;              // float4 per X row
  constexpr int YROWS = 256 / YQ;          // rows covered per pass
  constexpr int XROWS = 256 / XQ;
  constexpr int YP = BP / YROWS;           // passes
  constexpr int XP = BP / XROWS;
  static_assert(WAVES_M * WAVES_N == 4, "4 waves");
  static_assert(YP >= 1 && XP >= 1, "tile too wide for BP");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ys = smem;                 // [2][BP][LDY]
  float* Xs = smem + 2 * BP * LDY;  // [2][BP][LDX]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
  const int wm = wv / WAVES_N, wn = wv % WAVES_N;
  const int l31 = lane & 31, hi = lane >> 5;

  const int kk0 = blockIdx.x * BNK;
  const int co0 = blockIdx.y * BMC;
  const int p_begin = blockIdx.z * p.chunk;
  const int p_end = min(p.M, p_begin + p.chunk);
  if (p_begin >= p_end) return;

  // loader geometry: fixed channel column per thread, rows advance with the chunk
  const int yq = tid % YQ, yr = tid / YQ;
  const int xq = tid % XQ, xr = tid / XQ;
  const int yco = co0 + yq * 4;
  const bool yok = yco < p.ldy;  // ldy % 4 == 0 -> whole float4 in range
  const int kk = kk0 + xq * 4;
  const bool xok = kk < p.Ktot;
  const int tap = xok ? kk / p.Cs : 0;
  const int ci = kk - tap * p.Cs;
  const int dh = tap / p.KW - p.pad;
  const int dw = tap % p.KW - p.pad;
  const int hw = p.Ho * p.Wo;

  f32x4 ry[YP], rx[XP];
  auto load_tile = [&](int pp) {
#pragma unroll
    for (int i = 0; i < YP; ++i) {
      const int m = pp + yr + YROWS * i;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (yok && m < p_end) v = *reinterpret_cast<const f32x4*>(p.dy + (size_t)m * p.ldy + yco);
      ry[i] = v;
    }
#pragma unroll
    for (int i = 0; i < XP; ++i) {
      const int m = pp + xr + XROWS * i;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (xok && m < p_end) {
        const int b = m / hw;
        const int rem = m - b * hw;
        const int ho = rem / p.Wo;
        const int wo = rem - ho * p.Wo;
        const int h = ho * p.stride + dh, w = wo * p.stride + dw;
        if ((unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W)
          v = *reinterpret_cast<const f32x4*>(p.x + ((size_t)(b * p.H + h) * p.W + w) * p.Cs + ci);
      }
      rx[i] = v;
    }
  };
  auto store_tile = [&](int buf) {
    float* ys = Ys + buf * BP * LDY;
    float* xs = Xs + buf * BP * LDX;
#pragma unroll
    for (int i = 0; i < YP; ++i) *reinterpret_cast<f32x4*>(ys + (yr + YROWS * i) * LDY + yq * 4) = ry[i];
#pragma unroll
    for (int i = 0; i < XP; ++i) *reinterpret_cast<f32x4*>(xs + (xr + XROWS * i) * LDX + xq * 4) = rx[i];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  load_tile(p_begin);
  store_tile(0);
  __syncthreads();
  int cur = 0;
  for (int pp = p_begin; pp < p_end; pp += BP) {
    const bool more = pp + BP < p_end;
    if (more) load_tile(pp + BP);
    const float* ys = Ys + cur * BP * LDY + hi * LDY + wm * TM * 32 + l31;
    const float* xs = Xs + cur * BP * LDX + hi * LDX + wn * TN * 32 + l31;
#pragma unroll
    for (int s = 0; s < BP / 2; ++s) {
      float fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i] = ys[2 * s * LDY + i * 32];
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[j] = xs[2 * s * LDX + j * 32];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    if (more) store_tile(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

  // atomics: lanes 0-31 / 32-63 each add 128 contiguous bytes of one dWp row
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = kk0 + (wn * TN + j) * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = co0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (row < p.Nw && col < p.Ktot) atomicAdd(p.dwp + (size_t)row * p.Ktot + col, acc[i][j][r]);
      }
    }
}

// ---------------------------------------------------------------------------
// host-side launchers
// ---------------------------------------------------------------------------
template <int BM, int BN, int WMV, int WNV>
static int launch_conv(ConvP& p, hipStream_t st) {
  p.tiles_m = cdiv(p.M, BM);
  p.tiles_n = cdiv(p.shuffle ? p.Nw : p.ldy, BN);
  const size_t lds = (size_t)2 * (BM + BN) * LDT * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<BM, BN, WMV, WNV>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WMV, WNV>), dim3(p.tiles_m * p.tiles_n), dim3(256), lds, st, p);
  return vmtl_check_launch();
}

// number of row-blocks the forward kernel will use for a given problem (needed by
// callers that want the fused BatchNorm column partials: stats is [tiles_m][2][ldy]).
static void conv_pick_tile(int M, int ncols, int* bm, int* bn) {
  if (ncols <= 32) { *bm = 128; *bn = 32; }
  else if (ncols <= 64) { *bm = 128; *bn = 64; }
  else if (ncols <= 96 || (long long)cdiv(M, 128) * cdiv(ncols, 128) < 384) { *bm = 64; *bn = 64; }
  else { *bm = 128; *bn = 128; }
  if (*bm == 128 && *bn == 64 && (long long)cdiv(M, 128) < 256) { *bm = 64; *bn = 64; }
}

extern "C" int vmtl_conv2d_stats_rows(int B, int Ho, int Wo, int ldy) {
  int bm, bn;
  conv_pick_tile(B * Ho * Wo, ldy, &bm, &bn);
  return cdiv(B * Ho * Wo, bm);
}

// output rows covered by each stats row block (the last block may be partial)
extern "C" int vmtl_conv2d_stats_block(int B, int Ho, int Wo, int ldy) {
  int bm, bn;
  conv_pick_tile(B * Ho * Wo, ldy, &bm, &bn);
  return bm;
}

extern "C" int vmtl_conv2d_fwd(const float* x, const float* wp, const float* bias, float* y, float* stats,
                               int B, int H, int W, int Cs, int Ho, int Wo, int ldy, int Nw, int Cout,
                               int KH, int KW, int stride, int pad, int act, int shuffle, void* stream) {
  if (!x || !wp || !y) return VMTL_ERR_ARG;
  if (Cs <= 0 || (Cs & 3) || B <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0) return VMTL_ERR_ARG;
  if (KH <= 0 || KW <= 0 || stride <= 0 || pad < 0 || Nw <= 0 || Cout <= 0 || ldy <= 0) return VMTL_ERR_ARG;
  if (!shuffle && (Cout > ldy || Nw > ldy)) return VMTL_ERR_ARG;
  if (shuffle && (Nw != 4 * Cout || Cout > ldy || stats)) return VMTL_ERR_ARG;
  // every gathered input coordinate must be expressible; output extent must match the conv arithmetic
  if ((H + 2 * pad - KH) / stride + 1 != Ho || (W + 2 * pad - KW) / stride + 1 != Wo) return VMTL_ERR_ARG;
  if ((long long)B * Ho * Wo > 0x7fffffffLL || (long long)B * H * W > 0x7fffffffLL) return VMTL_ERR_ARG;
  ConvP p;
  p.x = x; p.wp = wp; p.bias = bias; p.y = y; p.stats = stats;
  p.B = B; p.H = H; p.W = W; p.Cs = Cs; p.Ho = Ho; p.Wo = Wo; p.ldy = ldy; p.Nw = Nw; p.Cout = Cout;
  p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad; p.Ktot = KH * KW * Cs; p.M = B * Ho * Wo;
  p.act = act; p.shuffle = shuffle;
  hipStream_t st = (hipStream_t)stream;
  if (shuffle && ldy > Cout &&  // the scatter only writes co < Cout: keep the pad-channel invariant
      hipMemsetAsync(y, 0, (size_t)B * 4 * Ho * Wo * ldy * sizeof(float), st) != hipSuccess)
    return VMTL_ERR_LAUNCH;
  int bm, bn;
  conv_pick_tile(p.M, shuffle ? Nw : ldy, &bm, &bn);
  if (bm == 128 && bn == 32) return launch_conv<128, 32, 4, 1>(p, st);
  if (bm == 128 && bn == 64) return launch_conv<128, 64, 2, 2>(p, st);
  if (bm == 64 && bn == 64) return launch_conv<64, 64, 2, 2>(p, st);
  return launch_conv<128, 128, 2, 2>(p, st);
}

template <int BMC, int BNK, int WMV, int WNV>
static int launch_wgrad(WgradP& p, hipStream_t st) {
  const int tk = cdiv(p.Ktot, BNK), tc = cdiv(p.Nw, BMC);
  // split the pixel axis until the grid comfortably fills 256 CUs
  long long tiles = (long long)tk * tc;
  int splits = (int)((2048 + tiles - 1) / tiles);
  const int max_splits = cdiv(p.M, 4 * BP);
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  int chunk = cdiv(cdiv(p.M, splits), BP) * BP;
  splits = cdiv(p.M, chunk);
  p.chunk = chunk;
  const size_t lds = (size_t)2 * BP * ((BMC + 4) + (BNK + 4)) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_kernel<BMC, BNK, WMV, WNV>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((conv_wgrad_kernel<BMC, BNK, WMV, WNV>), dim3(tk, tc, splits), dim3(256), lds, st, p);
  return vmtl_check_launch();
}

extern "C" int vmtl_conv2d_wgrad(const float* x, const float* dy, float* dwp, int B, int H, int W, int Cs,
                                 int Ho, int Wo, int ldy, int Nw, int KH, int KW, int stride, int pad,
                                 void* stream) {
  if (!x || !dy || !dwp) return VMTL_ERR_ARG;
  if (Cs <= 0 || (Cs & 3) || (ldy & 3) || Nw <= 0 || Nw > ldy) return VMTL_ERR_ARG;
  if ((H + 2 * pad - KH) / stride + 1 != Ho || (W + 2 * pad - KW) / stride + 1 != Wo) return VMTL_ERR_ARG;
  if ((long long)B * Ho * Wo > 0x7fffffffLL || (long long)B * H * W > 0x7fffffffLL) return VMTL_ERR_ARG;
  WgradP p;
  p.x = x; p.dy = dy; p.dwp = dwp; p.B = B; p.H = H; p.W = W; p.Cs = Cs; p.Ho = Ho; p.Wo = Wo; p.ldy = ldy;
  p.Nw = Nw; p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad; p.Ktot = KH * KW * Cs; p.M = B * Ho * Wo;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(dwp, 0, (size_t)Nw * p.Ktot * sizeof(float), st);
  if (e != hipSuccess) return VMTL_ERR_LAUNCH;
  if (Nw <= 32) return launch_wgrad<32, 128, 1, 4>(p, st);
  if (Nw <= 64) return launch_wgrad<64, 128, 2, 2>(p, st);
  return launch_wgrad<128, 128, 2, 2>(p, st);
}
