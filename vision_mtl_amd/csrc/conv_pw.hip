// Pointwise (1x1, stride 1) convolution = plain GEMM  Y[M][ldy] = X[M][Ks] * Wp[Nw][Ks]^T  for the MobileNetV3
// encoder's expand / project convs (timm pointwise convs of the `basic` / `csnet` encoders, reference
// vision_mtl/utils/model_utils.py:25-34) and the 1x1 convs of MTAN's attention modules (reference
// models/mtan_model.py:31,39,105,113,369), forward and data gradient (the data gradient of a 1x1 conv is the same
// GEMM over dY with the transposed packing).
//
// Why not the implicit-GEMM kernel (conv_igemm.hip): these launches are SMALL (M = B*H*W <= 262144, K = 16..960,
// N = 16..960; 0.03-0.6 GFLOP) and sit on a dependent chain.  There they measured 12-50 us each - a floor of ~12 us
// from the LDS-staged pipeline (global -> registers -> LDS -> barrier -> fragments, plus three barriers of statistics
// epilogue), and K-latency-bound above it when M*N yields fewer workgroups than CUs (each K step then pays a full
// global-load latency for a few MFMAs).  Here:
//   * no LDS in the K loop and no barrier: every wave loads its own MFMA fragments straight from global memory /
//     L2 (rows are K-contiguous: a lane reads 16 bytes, the four lane quarters of a row read one 64-byte segment),
//     two k-groups in flight;
//   * when the tile grid alone cannot fill the chip the four waves of a workgroup SPLIT K (KW = 4) and combine
//     through LDS once at the end - 4x the waves for the tile-starved deep layers (M = 1024 / 4096);
//   * the output tile goes through LDS so that y is written as coalesced float4 rows, with the bias add and the
//     BatchNorm (mean, M2) partials of the tile taken on the way.
#include "common.h"

struct PwP {
  const float* x;     // [M][Ks]
  const float* wp;    // [Nw][Ks]
  const float* bias;  // [Nw] or null
  float* y;           // [M][ldy]
  float* stats;       // optional [tiles_m][2][ldy]: per-row-block column (mean, M2)
  int M, Ks, ldy, Nw, Cout;
  int tiles_m, tiles_n;
  // virtual concat (MTAN attention, reference models/mtan_model.py:57-59,139-141): the K axis is [x | x2] - columns
  // [0, K1) of a row come from x (row stride K1), [K1, Ks) from x2 (row stride Ks - K1); and its mirror for the data
  // gradient: output columns [0, N1) go to y (row stride N1), [N1, ldy) to y2 (row stride ldy - N1)
  const float* x2;  // null: single source
  float* y2;        // null: single destination
  int K1, N1;
  // pre-activation node (PRO): the A operand is act(pa[k] * x + pc[k]) - BatchNorm + activation of the layer that
  // produced x, applied to the fragments on their way into the MFMAs; a_out (nullable) receives the activated matrix
  // once (workgroups of column tile 0), for the weight-gradient kernel
  const float* pa;
  const float* pc;
  const float* res;  // nullable [M][Ks]: added after the activation (a block's residual branch: a = act(pa*x + pc) + res)
  float* a_out;
  int act_in;
  // BatchNorm-backward epilogue (ez_x != null): the launch is the data gradient of a conv whose input was
  // act(BN(ez_x)); it stores dz = acc * act'(gamma * xhat + beta) instead of acc and per-row-block (sum dz,
  // sum dz * xhat) in `stats` - the BatchNorm backward then needs no reduction pass of its own
  const float* ez_x;
  const float* ez_mean;
  const float* ez_invstd;
  const float* ez_gamma;
  const float* ez_beta;
  const float* ez_add;  // nullable [M][ldy]: a second gradient of the differentiated tensor, added to acc first
  int ez_act;
};

// wave tile 32 rows x (16*TN) columns; KW waves of the workgroup split K, the other 4/KW stack along M
template <int TN, int KW, bool SRC2 = false, bool PRO = false>
__global__ __launch_bounds__(256) void pw_gemm_kernel(PwP p) {
  constexpr int TM = 2;
  constexpr int RG = 4 / KW;        // row groups (waves along M)
  constexpr int BM = RG * 32;       // rows per workgroup
  constexpr int BN = 16 * TN;       // columns per workgroup
  constexpr int OS = BN + 4;        // LDS row stride (floats): = 4 (mod 8) -> conflict-free ds_write_b32 from the C layout
  __shared__ __attribute__((aligned(16))) float tile[KW][BM][OS];
  __shared__ f32x4 red[2][256];

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int l15 = lane & 15, lq = lane >> 4;
  const int ks = wv % KW, rg = wv / KW;
  const int bid = xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);
  const int tile_m = bid / p.tiles_n, tile_n = bid - tile_m * p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  // fragment sources: A rows m0 + rg*32 + 16 i + l15, B rows n0 + 16 j + l15; k = 16 g + 4 lq .. + 3
  const float* ap[TM];
  const float* ap2[TM];  // SRC2: the same row of the second source, shifted so that index k addresses column k - K1
  const float* bp[TN];
  bool aok[TM], bok[TN];
  const int lda = SRC2 ? p.K1 : p.Ks;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + rg * 32 + 16 * i + l15;
    aok[i] = m < p.M;
    ap[i] = p.x + (size_t)(aok[i] ? m : 0) * lda + 4 * lq;
    ap2[i] = SRC2 ? p.x2 + (size_t)(aok[i] ? m : 0) * (p.Ks - p.K1) + 4 * lq - p.K1 : nullptr;
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + 16 * j + l15;
    bok[j] = n < p.Nw;
    bp[j] = p.wp + (size_t)(bok[j] ? n : 0) * p.Ks + 4 * lq;
  }
  const int G = (p.Ks + 15) >> 4;
  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  struct Frag { f32x4 a[TM], b[TN], pa, pc, r[TM]; };
  auto load = [&](int g, Frag& f) {
    const bool kok = g < G && 16 * g + 4 * lq < p.Ks;
    if (PRO) {
      f.pa = kok ? *reinterpret_cast<const f32x4*>(p.pa + 16 * g + 4 * lq) : (f32x4){0.f, 0.f, 0.f, 0.f};
      f.pc = kok ? *reinterpret_cast<const f32x4*>(p.pc + 16 * g + 4 * lq) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const float* src = (SRC2 && 16 * g + 4 * lq >= p.K1) ? ap2[i] : ap[i];  // K1 % 4 == 0: a quad has one source
      f.a[i] = (kok && aok[i]) ? *reinterpret_cast<const f32x4*>(src + 16 * g) : (f32x4){0.f, 0.f, 0.f, 0.f};
      if (PRO)
        f.r[i] = (p.res != nullptr && kok && aok[i]) ? *reinterpret_cast<const f32x4*>(p.res + (ap[i] - p.x) + 16 * g)
                                                     : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int j = 0; j < TN; ++j)
      f.b[j] = (kok && bok[j]) ? *reinterpret_cast<const f32x4*>(bp[j] + 16 * g) : (f32x4){0.f, 0.f, 0.f, 0.f};
  };
  auto mma = [&](Frag& f, int g) {
    if (PRO) {  // the producer's BatchNorm + activation, on the fragments (k >= Ks: pa = pc = 0 and B is zero there)
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        f32x4 v = f.a[i] * f.pa + f.pc;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = act_fwd(v[e], p.act_in);
        v += f.r[i];
        f.a[i] = v;
        if (p.a_out != nullptr && tile_n == 0 && aok[i] && g < G && 16 * g + 4 * lq < p.Ks)
          *reinterpret_cast<f32x4*>(p.a_out + (size_t)(m0 + rg * 32 + 16 * i + l15) * p.Ks + 16 * g + 4 * lq) = v;
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[i][e], f.b[j][e], acc[i][j], 0, 0, 0);
  };
  // this wave's k-groups: ks, ks + KW, ...; three in flight
  Frag f0, f1, f2;
  int g = ks;
  load(g, f0);
  load(g + KW, f1);
  for (; g < G; g += 3 * KW) {
    load(g + 2 * KW, f2);
    mma(f0, g);
    if (g + KW >= G) break;
    load(g + 3 * KW, f0);
    mma(f1, g + KW);
    if (g + 2 * KW >= G) break;
    load(g + 4 * KW, f1);
    mma(f2, g + 2 * KW);
  }

  // ---- C layout -> LDS (one plane per K slice); C: column = lane & 15, row = 4 * (lane >> 4) + reg ----
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) tile[ks][rg * 32 + 16 * i + 4 * lq + r][16 * j + l15] = acc[i][j][r];
  __syncthreads();

  // ---- epilogue: thread <-> (row, column quad); K slices summed, bias, coalesced float4 stores, statistics ----
  constexpr int Q = BN / 4;            // quads per row
  constexpr int RPP = 256 / Q;         // rows per pass
  constexpr int PASSES = (BM + RPP - 1) / RPP;
  const int q = tid % Q, r0 = tid / Q;
  const int n4 = n0 + 4 * q;
  f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
  if (p.bias != nullptr) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (n4 + e < p.Nw) bias4[e] = p.bias[n4 + e];
  }
  f32x4 val[PASSES];
  f32x4 s1 = {0.f, 0.f, 0.f, 0.f};
  f32x4 s2x = {0.f, 0.f, 0.f, 0.f};  // BatchNorm-backward epilogue: sum dz * xhat
  const bool ez = p.ez_x != nullptr;
  f32x4 e_mean = {0.f, 0.f, 0.f, 0.f}, e_is = e_mean, e_g = e_mean, e_b = e_mean;
  if (ez && n4 < p.ldy) {
    e_mean = *reinterpret_cast<const f32x4*>(p.ez_mean + n4);
    e_is = *reinterpret_cast<const f32x4*>(p.ez_invstd + n4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      e_g[e] = n4 + e < p.Cout ? (p.ez_gamma ? p.ez_gamma[n4 + e] : 1.f) : 0.f;
      e_b[e] = n4 + e < p.Cout ? (p.ez_beta ? p.ez_beta[n4 + e] : 0.f) : 0.f;
    }
  }
  int cnt = 0;
#pragma unroll
  for (int ps = 0; ps < PASSES; ++ps) {
    const int r = r0 + ps * RPP;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (r < BM) {
      v = *reinterpret_cast<const f32x4*>(&tile[0][r][4 * q]);
#pragma unroll
      for (int k = 1; k < KW; ++k) v += *reinterpret_cast<const f32x4*>(&tile[k][r][4 * q]);
      v += bias4;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (n4 + e >= p.Cout) v[e] = 0.f;
      const int m = m0 + r;
      if (ez && m < p.M && n4 < p.ldy) {
        if (p.ez_add != nullptr) v += *reinterpret_cast<const f32x4*>(p.ez_add + (size_t)m * p.ldy + n4);
        const f32x4 xh = (*reinterpret_cast<const f32x4*>(p.ez_x + (size_t)m * p.ldy + n4) - e_mean) * e_is;
        const f32x4 z = e_g * xh + e_b;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= act_grad(z[e], p.ez_act);
        s2x += v * xh;
      }
      if (m < p.M) {
        if (n4 < p.ldy) {
          if (p.y2 == nullptr) *reinterpret_cast<f32x4*>(p.y + (size_t)m * p.ldy + n4) = v;
          else if (n4 < p.N1) *reinterpret_cast<f32x4*>(p.y + (size_t)m * p.N1 + n4) = v;
          else *reinterpret_cast<f32x4*>(p.y2 + (size_t)m * (p.ldy - p.N1) + (n4 - p.N1)) = v;
        }
        s1 += v;
        ++cnt;
      } else {
        v = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    }
    val[ps] = v;
  }
  if (p.stats != nullptr && ez) {
    // per-tile (sum dz, sum dz * xhat) of every column
    red[0][tid] = s1;
    red[1][tid] = s2x;
    __syncthreads();
    if (tid < Q && n4 < p.ldy) {
      f32x4 a = {0.f, 0.f, 0.f, 0.f}, c = a;
#pragma unroll
      for (int k = 0; k < RPP; ++k) {
        a += red[0][tid + Q * k];
        c += red[1][tid + Q * k];
      }
      *reinterpret_cast<f32x4*>(p.stats + ((size_t)tile_m * 2 + 0) * p.ldy + n4) = a;
      *reinterpret_cast<f32x4*>(p.stats + ((size_t)tile_m * 2 + 1) * p.ldy + n4) = c;
    }
  } else if (p.stats != nullptr) {
    // per-tile (mean, M2) of every column in one reduction round: sums of d = v - pivot and d*d, pivot = the column's
    // value in the tile's first row (still in the LDS tile: shifted sums stay accurate when |mean| >> std).  Rows of a
    // column live in the RPP threads tid = q + Q * r0.
    const int nvalid = min(BM, p.M - m0);
    f32x4 pv = *reinterpret_cast<const f32x4*>(&tile[0][0][4 * q]);
#pragma unroll
    for (int k = 1; k < KW; ++k) pv += *reinterpret_cast<const f32x4*>(&tile[k][0][4 * q]);
    pv += bias4;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (n4 + e >= p.Cout) pv[e] = 0.f;
    f32x4 d1 = {0.f, 0.f, 0.f, 0.f}, d2 = d1;
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps)
      if (r0 + ps * RPP < BM && m0 + r0 + ps * RPP < p.M) {
        const f32x4 d = val[ps] - pv;
        d1 += d;
        d2 += d * d;
      }
    red[0][tid] = d1;
    red[1][tid] = d2;
    __syncthreads();
    if (tid < Q && n4 < p.ldy) {  // r0 == 0: q == tid
      f32x4 a = {0.f, 0.f, 0.f, 0.f}, c = a;
#pragma unroll
      for (int k = 0; k < RPP; ++k) {
        a += red[0][tid + Q * k];
        c += red[1][tid + Q * k];
      }
      const float inv = 1.f / (float)nvalid;
      *reinterpret_cast<f32x4*>(p.stats + ((size_t)tile_m * 2 + 0) * p.ldy + n4) = pv + a * inv;
      *reinterpret_cast<f32x4*>(p.stats + ((size_t)tile_m * 2 + 1) * p.ldy + n4) = c - a * a * inv;
    }
  }
  (void)cnt;
}


// ---------------------------------------------------------------------------------------------- large-M kernel
// The same GEMM for LARGE row counts (MTAN's attention 1x1 convs at 128^2 / 256^2 pixels, bs 16: M = 262144 / 1048576,
// K = 64..384, N = 32..192; reference models/mtan_model.py:31,39,57-66,105,113,139-148).  There the kernel above runs at
// 1.7-2.5 TB/s and 60-70 TF - neither roofline (profiles/r02_mtan_*): fragments straight from global memory read A in
// 64-byte row segments, twice for N = 128 (two 64-column tiles), and every workgroup re-fetches B.  Here
//   * PERSISTENT workgroups (one per CU) keep the whole weight matrix B [BN][K] in LDS, loaded once;
//   * A is streamed ONE ROW TILE [BM][K] AT A TIME: every thread fetches its 128-byte-coalesced share of tile t+1 into
//     registers BEFORE the K loop of tile t and stores it to LDS after tile t's epilogue - one full tile (32-128 KB per
//     CU) is in flight under the MFMAs, and the K loop itself has NO barrier and no global traffic (all of K is in LDS);
//   * full-width tiles: N <= 128 is one column tile, A is read exactly once;
//   * barriers are LDS-only (s_waitcnt lgkmcnt(0); s_barrier): __syncthreads() would also wait for the prefetch;
//   * the pre-activation prologue act(pa*x + pc) runs when a tile is STAGED (once per element, not per fragment use) and
//     the activated matrix goes back to HBM as the coalesced rows it was loaded in;
//   * epilogue as above (bias, pad-channel zeros, per-tile BatchNorm (mean, M2), two-destination split store).
// LDS images are [K chunk of 32][row][32 floats] with the 16-byte slot XOR-swizzled by (row & 7): the layout of
// conv_igemm.hip's staging tiles (0 bank-conflict cycles measured there), for A and B alike.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// wave tile (TM*16) x (TN*16), WM x WN waves; KC = K chunks of 32 the instantiation can hold (Ks <= 32 * KC)
// EZ: BatchNorm-backward epilogue (see PwP::ez_x; no addend): dz = acc * act'(gamma * xhat + beta) and (sum dz, sum dz * xhat);
// the ez_x values of a tile's epilogue are prefetched one tile ahead like A.
template <int TM, int TN, int WM, int WN, int KC, bool SRC2, bool PRO, bool EZ = false>
__global__ __launch_bounds__(WM* WN * 64) void pw_big_kernel(PwP p, int nprog) {
  static_assert(WM * WN == 4 || WM * WN == 8, "4 or 8 waves");
  constexpr int NTHR = WM * WN * 64;
  constexpr int SR = NTHR / 8;  // rows one staging pass covers (8 threads x 16 bytes = one 128-byte row chunk)
  constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
  static_assert(BM % SR == 0, "whole staging passes");
  constexpr int RA = BM / SR, RB = (BN + SR - 1) / SR;  // staging rows per thread and chunk
  constexpr int OS = BN + 4;                         // output tile row stride (floats)
  constexpr int A_FLOATS = KC * BM * 32, O_FLOATS = BM * OS;
  constexpr int AO_FLOATS = A_FLOATS > O_FLOATS ? A_FLOATS : O_FLOATS;  // the output tile aliases the A image
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Bs = smem;                 // [KC][BN][32]
  float* As = smem + KC * BN * 32;  // [KC][BM][32]  | output tile [BM][OS]
  f32x4* red = reinterpret_cast<f32x4*>(As + AO_FLOATS);  // [2][NTHR]

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int l15 = lane & 15, lq = lane >> 4;
  const int wm = wv / WN, wn = wv % WN;
  const int tile_n = blockIdx.y, n0 = tile_n * BN;
  const int nkc = (p.Ks + 31) >> 5;   // K chunks in use (<= KC)
  const int nkg = (p.Ks + 15) >> 4;   // 16-wide k-groups in use
  const int r0 = tid >> 3, kq = tid & 7;
  const int ks = (kq ^ (r0 & 7)) * 4;  // swizzled slot of this thread's k-quad ((r0 + SR i) & 7 == r0 & 7)
  const int lda = SRC2 ? p.K1 : p.Ks;

  // ---- B: the whole [BN][Ks] weight tile, once ----
#pragma unroll
  for (int c = 0; c < KC; ++c)
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const int row = r0 + SR * i, n = n0 + row, k = 32 * c + 4 * kq;
      if (BN % SR == 0 || row < BN) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (c < nkc && n < p.Nw && k < p.Ks) v = *reinterpret_cast<const f32x4*>(p.wp + (size_t)n * p.Ks + k);
        *reinterpret_cast<f32x4*>(Bs + (c * BN + row) * 32 + ks) = v;
      }
    }
  // prologue coefficients of this thread's k-quads (fixed per thread: one quad per chunk)
  f32x4 cpa[PRO ? KC : 1], cpc[PRO ? KC : 1];
  if (PRO) {
#pragma unroll
    for (int c = 0; c < KC; ++c) {
      const int k = 32 * c + 4 * kq;
      const bool ok = c < nkc && k < p.Ks;
      cpa[c] = ok ? *reinterpret_cast<const f32x4*>(p.pa + k) : (f32x4){0.f, 0.f, 0.f, 0.f};
      cpc[c] = ok ? *reinterpret_cast<const f32x4*>(p.pc + k) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  }

  // ---- A tile prefetch: pre[c][i] = x[m0 + r0 + SR i][32 c + 4 kq .. +3] ----
  f32x4 pre[KC][RA];
  auto fetch = [&](int tile_m) {
    const int m0 = tile_m * BM;
#pragma unroll
    for (int c = 0; c < KC; ++c) {
      const int k = 32 * c + 4 * kq;
#pragma unroll
      for (int i = 0; i < RA; ++i) {
        const int m = m0 + r0 + SR * i;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (c < nkc && m < p.M && k < p.Ks) {
          const float* src = (SRC2 && k >= p.K1) ? p.x2 + (size_t)m * (p.Ks - p.K1) + (k - p.K1) : p.x + (size_t)m * lda + k;
          v = *reinterpret_cast<const f32x4*>(src);
        }
        pre[c][i] = v;
      }
    }
  };
  auto stage = [&](int tile_m) {  // registers -> LDS (+ prologue, + activated copy back to HBM)
    const int m0 = tile_m * BM;
#pragma unroll
    for (int c = 0; c < KC; ++c) {
      if (c < nkc) {
        const int k = 32 * c + 4 * kq;
#pragma unroll
        for (int i = 0; i < RA; ++i) {
          f32x4 v = pre[c][i];
          if (PRO) {
            v = v * cpa[c] + cpc[c];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = act_fwd(v[e], p.act_in);
            const int m = m0 + r0 + SR * i;
            if (p.a_out != nullptr && tile_n == 0 && m < p.M && k < p.Ks)
              *reinterpret_cast<f32x4*>(p.a_out + (size_t)m * p.Ks + k) = v;
            if (m >= p.M) v = (f32x4){0.f, 0.f, 0.f, 0.f};  // rows past M: act(pc) need not be 0
          }
          *reinterpret_cast<f32x4*>(As + (c * BM + r0 + SR * i) * 32 + ks) = v;
        }
      }
    }
  };

  f32x4 acc[TM][TN];
  // epilogue geometry: thread <-> (row, column quad)
  constexpr int Q = BN / 4, RPP = NTHR / Q, PASSES = (BM + RPP - 1) / RPP;
  const int q = tid % Q, er0 = tid / Q;
  const int n4 = n0 + 4 * q;
  const bool ethread = er0 < RPP;
  f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
  if (p.bias != nullptr) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (n4 + e < p.Nw) bias4[e] = p.bias[n4 + e];
  }

  // EZ: per-column constants of this thread's quad, and the ez_x prefetch registers (next tile / current tile)
  f32x4 e_mean = {0.f, 0.f, 0.f, 0.f}, e_is = e_mean, e_g = e_mean, e_b = e_mean;
  f32x4 ez_next[EZ ? PASSES : 1], ez_cur[EZ ? PASSES : 1];
  if (EZ && ethread && n4 < p.ldy) {
    e_mean = *reinterpret_cast<const f32x4*>(p.ez_mean + n4);
    e_is = *reinterpret_cast<const f32x4*>(p.ez_invstd + n4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      e_g[e] = n4 + e < p.Cout ? (p.ez_gamma ? p.ez_gamma[n4 + e] : 1.f) : 0.f;
      e_b[e] = n4 + e < p.Cout ? (p.ez_beta ? p.ez_beta[n4 + e] : 0.f) : 0.f;
    }
  }
  auto fetch_ez = [&](int tm) {
    if (EZ) {
#pragma unroll
      for (int ps = 0; ps < PASSES; ++ps) {
        const int r = er0 + ps * RPP, m = tm * BM + r;
        ez_next[ps] = (ethread && r < BM && m < p.M && n4 < p.ldy) ? *reinterpret_cast<const f32x4*>(p.ez_x + (size_t)m * p.ldy + n4)
                                                                   : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    }
  };
  // BatchNorm partials: ONE statistics row per workgroup when every workgroup walks the same number of full tiles
  // (shifted sums around the pivot of its first tile, kept per thread across tiles and folded once at the end): M = 1 M
  // rows are 256 statistics rows instead of 16384 (the finalize kernel read 17 MB of partials per such layer), and the
  // per-tile reduction barrier disappears.  Otherwise one row per tile.
  const bool acc_stats = p.stats != nullptr && p.M % BM == 0 && p.tiles_m % nprog == 0;
  f32x4 pv_acc = {0.f, 0.f, 0.f, 0.f}, d1_acc = pv_acc, d2_acc = pv_acc;
  int tile_m = blockIdx.x;
  if (tile_m < p.tiles_m) {
    fetch(tile_m);
    fetch_ez(tile_m);
  }
  bool first = true;
  for (; tile_m < p.tiles_m; tile_m += nprog) {
    // tile_m's rows are in `pre` (in flight or landed): LDS image is free (first tile: B stores above need the barrier too)
    stage(tile_m);
    if (EZ) {
#pragma unroll
      for (int ps = 0; ps < PASSES; ++ps) ez_cur[ps] = ez_next[ps];  // fetched a whole tile ago
    }
    lds_barrier();
    const bool first_tile = first;
    first = false;
    const int next = tile_m + nprog;
    if (next < p.tiles_m) {  // stays in flight under the K loop, the epilogue and its stores
      fetch(next);
      fetch_ez(next);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // ---- K loop: no barrier, no global traffic ----
    const float* a = As + ((wm * TM * 16 + l15) * 32);
    const float* b = Bs + ((wn * TN * 16 + l15) * 32);
    for (int g = 0; g < nkg; ++g) {
      const int c = g >> 1;
      const int so = ((((g & 1) << 2) + lq) ^ (l15 & 7)) * 4;
      f32x4 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const f32x4*>(a + (c * BM + i * 16) * 32 + so);
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const f32x4*>(b + (c * BN + j * 16) * 32 + so);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i][e], fb[j][e], acc[i][j], 0, 0, 0);
    }
    lds_barrier();  // every wave is done reading the A image: it becomes the output tile
    float* tile = As;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          tile[((wm * TM + i) * 16 + 4 * lq + r) * OS + (wn * TN + j) * 16 + l15] = acc[i][j][r];
    lds_barrier();
    // ---- epilogue: bias, pad zeros, coalesced float4 stores, per-tile (mean, M2) ----
    const int m0 = tile_m * BM;
    f32x4 val[PASSES];
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2x = s1;  // EZ: this tile's (sum dz, sum dz * xhat) of the thread's rows
    if (ethread) {
#pragma unroll
      for (int ps = 0; ps < PASSES; ++ps) {
        const int r = er0 + ps * RPP;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (r < BM && m0 + r < p.M) {
          v = *reinterpret_cast<const f32x4*>(tile + r * OS + 4 * q) + bias4;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (n4 + e >= p.Cout) v[e] = 0.f;
          if (EZ && n4 < p.ldy) {
            const f32x4 xh = (ez_cur[ps] - e_mean) * e_is;
            const f32x4 z = e_g * xh + e_b;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= act_grad(z[e], p.ez_act);
            s1 += v;
            s2x += v * xh;
          }
          const int m = m0 + r;
          if (n4 < p.ldy) {
            if (p.y2 == nullptr) *reinterpret_cast<f32x4*>(p.y + (size_t)m * p.ldy + n4) = v;
            else if (n4 < p.N1) *reinterpret_cast<f32x4*>(p.y + (size_t)m * p.N1 + n4) = v;
            else *reinterpret_cast<f32x4*>(p.y2 + (size_t)m * (p.ldy - p.N1) + (n4 - p.N1)) = v;
          }
        }
        val[ps] = v;
      }
    }
    if (EZ && p.stats != nullptr) {
      // plain column sums (sum dz, sum dz * xhat): per workgroup across its tiles, or per tile
      d1_acc += s1;
      d2_acc += s2x;
      if (!acc_stats) {
        red[tid] = d1_acc;
        red[NTHR + tid] = d2_acc;
        d1_acc = d2_acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        lds_barrier();
        if (tid < Q && n4 < p.ldy) {
          f32x4 sa = {0.f, 0.f, 0.f, 0.f}, sc = sa;
#pragma unroll
          for (int k = 0; k < RPP; ++k) {
            sa += red[tid + Q * k];
            sc += red[NTHR + tid + Q * k];
          }
          *reinterpret_cast<f32x4*>(p.stats + ((size_t)tile_m * 2 + 0) * p.ldy + n4) = sa;
          *reinterpret_cast<f32x4*>(p.stats + ((size_t)tile_m * 2 + 1) * p.ldy + n4) = sc;
        }
      }
    } else if (acc_stats) {
      if (ethread) {
        if (first_tile) {
          pv_acc = *reinterpret_cast<const f32x4*>(tile + 4 * q) + bias4;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (n4 + e >= p.Cout) pv_acc[e] = 0.f;
        }
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps)
          if (er0 + ps * RPP < BM) {
            const f32x4 d = val[ps] - pv_acc;
            d1_acc += d;
            d2_acc += d * d;
          }
      }
    } else if (p.stats != nullptr) {
      // shifted sums around the column's value in the tile's first row (accurate when |mean| >> std)
      const int nvalid = min(BM, p.M - m0);
      f32x4 pv = {0.f, 0.f, 0.f, 0.f}, d1 = pv, d2 = pv;
      if (ethread) {
        pv = *reinterpret_cast<const f32x4*>(tile + 4 * q) + bias4;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (n4 + e >= p.Cout) pv[e] = 0.f;
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps)
          if (er0 + ps * RPP < BM && m0 + er0 + ps * RPP < p.M) {
            const f32x4 d = val[ps] - pv;
            d1 += d;
            d2 += d * d;
          }
      }
      red[tid] = d1;
      red[NTHR + tid] = d2;
      lds_barrier();
      if (tid < Q && n4 < p.ldy) {  // er0 == 0: q == tid
        f32x4 sa = {0.f, 0.f, 0.f, 0.f}, sc = sa;
#pragma unroll
        for (int k = 0; k < RPP; ++k) {
          sa += red[tid + Q * k];
          sc += red[NTHR + tid + Q * k];
        }
        const float inv = 1.f / (float)nvalid;
        *reinterpret_cast<f32x4*>(p.stats + ((size_t)tile_m * 2 + 0) * p.ldy + n4) = pv + sa * inv;
        *reinterpret_cast<f32x4*>(p.stats + ((size_t)tile_m * 2 + 1) * p.ldy + n4) = sc - sa * sa * inv;
      }
    }
    lds_barrier();  // the output tile has been read: the region takes the next A image
  }
  if (acc_stats && blockIdx.x < p.tiles_m) {
    // every thread of a column (q) used the SAME pivot: the one thread er0 == 0 read from the first tile's row 0 is not
    // shared - broadcast it through LDS first
    if (ethread && er0 == 0) red[tid] = pv_acc;
    lds_barrier();
    const f32x4 pv = ethread ? red[q] : (f32x4){0.f, 0.f, 0.f, 0.f};
    lds_barrier();
    // a thread's sums are around ITS pivot only if its pivot equals the column's: all threads of a column loaded the same
    // tile row 0, so pv_acc == pv bit for bit (same LDS value, same bias) - nothing to re-centre
    red[tid] = d1_acc;
    red[NTHR + tid] = d2_acc;
    lds_barrier();
    if (tid < Q && n4 < p.ldy) {
      f32x4 sa = {0.f, 0.f, 0.f, 0.f}, sc = sa;
#pragma unroll
      for (int k = 0; k < RPP; ++k) {
        sa += red[tid + Q * k];
        sc += red[NTHR + tid + Q * k];
      }
      const float inv = 1.f / (float)((p.tiles_m / nprog) * BM);
      if (EZ) {  // plain sums
        *reinterpret_cast<f32x4*>(p.stats + ((size_t)blockIdx.x * 2 + 0) * p.ldy + n4) = sa;
        *reinterpret_cast<f32x4*>(p.stats + ((size_t)blockIdx.x * 2 + 1) * p.ldy + n4) = sc;
      } else {
        *reinterpret_cast<f32x4*>(p.stats + ((size_t)blockIdx.x * 2 + 0) * p.ldy + n4) = pv + sa * inv;
        *reinterpret_cast<f32x4*>(p.stats + ((size_t)blockIdx.x * 2 + 1) * p.ldy + n4) = sc - sa * sa * inv;
      }
    }
  }
  (void)first;
}

// ---------------------------------------------------------------------------------------------- host side
// (TN, KW) for a problem: 64-wide tiles unless N <= 32; split K over the 4 waves when the 128-row tiling leaves the
// chip under-filled and there is K to split
static void pw_pick(int M, int ldy, int Ks, int* tn, int* kw) {
  *tn = ldy <= 32 ? 2 : 4;
  const int bn = 16 * *tn;
  const long long wgs = (long long)cdiv(M, 128) * cdiv(ldy, bn);
  *kw = (wgs < 512 && Ks >= 64) ? 4 : 1;
  static EnvInt force{"VMTL_PW_KW", 0};  // tuning aid
  const int v = env_int(force);
  if (v == 1 || v == 4) *kw = v;
}

// ---- large-M path: configuration id for a problem, or -1 (see pw_big_kernel) ----
//  id  wave tile x waves      BM x BN    KC (K <= 32 KC)
//   0  <2,4> x 2x2            64 x 128   2, 4, 6
//   1  <2,2> x 2x2            64 x 64    4, 8
//   2  <1,1> x 4x2 (8 waves)  64 x 32    4
//   3  <2,3> x 2x2            64 x 96    4
// (each also as an 8-wave workgroup - two waves per SIMD - with half the wave tile: VMTL_PW_BIG_WAVES)
struct BigCfg { int id, bm, bn, kc; };
static bool pw_big_cfg(int M, int ldy, int Ks, BigCfg* out) {
  static EnvInt e_on{"VMTL_PW_BIG", 1};  // tuning aid: 0 = every 1x1 conv on pw_gemm_kernel
  if (!env_int(e_on)) return false;
  // large problems only: below, the fragment-from-global kernel wins on launch latency (and the encoder chains of
  // basic / csnet - K = 16..960 at M <= 262144 - stay on it)
  if (M < 65536 || Ks < 32 || Ks > 256 || ldy < 32 || (long long)M * ldy < (1ll << 23)) return false;
  const int kc = Ks <= 64 ? 2 : Ks <= 128 ? 4 : Ks <= 192 ? 6 : 8;
  BigCfg c;
  if (ldy <= 32) {
    // (4-wave 256 x 32 tiles measured SLOWER than pw_gemm_kernel on the one shape that uses this row - M = 1 M, K = 128,
    // N = 32 with the staging prologue: 338 vs 312 us; now 64 x 32 tiles, 8 waves, two workgroups per CU)
    if (Ks > 128) return false;
    c = {2, 64, 32, 4};
  } else if (ldy <= 64) {
    c = {1, 64, 64, Ks <= 128 ? 4 : 8};
  } else if (Ks > 192) {
    c = {1, 64, 64, 8};  // B [128][256] does not fit next to A: 64-column tiles (A re-read from L2)
  } else if (ldy % 96 == 0 && ldy % 128 != 0 && Ks <= 128) {
    c = {3, 64, 96, 4};
  } else {
    c = {0, 64, 128, kc};
  }
  if (c.id == 1 && c.kc < 4) c.kc = 4;
  *out = c;
  return true;
}

// variant 0: launches that may run on the large-M kernel (vmtl_conv1x1_fwd / _cat_fwd / _bn_fwd / _cat_dgrad / _bnbwd);
// 1: launches that always run on pw_gemm_kernel (residual operand, BatchNorm-backward epilogue with an addend)
static int pw_num_cus();

// programs (persistent workgroups per column tile) of the large-M kernel for this problem: one per CU, two where the LDS
// images of two workgroups fit a CU (their staging / epilogue phases then overlap the other's K loop)
static int pw_big_nprog(int M, int ldy, const BigCfg& c) {
  const int tiles_m = cdiv(M, c.bm), tiles_n = cdiv(ldy, c.bn);
  const int a_floats = c.kc * c.bm * 32, o_floats = c.bm * (c.bn + 4);
  const long long lds = ((long long)c.kc * c.bn * 32 + (a_floats > o_floats ? a_floats : o_floats)) * 4 + 2 * 512 * 16;
  static EnvInt e_occ{"VMTL_PW_BIG_OCC", 2};  // tuning aid
  const int occ = (2 * lds <= 160 * 1024 && env_int(e_occ) >= 2) ? 2 : 1;
  int nprog = occ * pw_num_cus() / tiles_n;
  if (nprog < 1) nprog = 1;
  return nprog > tiles_m ? tiles_m : nprog;
}

extern "C" int vmtl_conv1x1_stats_block(int M, int ldy, int Ks, int variant) {
  BigCfg c;
  if (variant == 0 && pw_big_cfg(M, ldy, Ks, &c)) {
    const int tiles_m = cdiv(M, c.bm), nprog = pw_big_nprog(M, ldy, c);
    return (M % c.bm == 0 && tiles_m % nprog == 0) ? (tiles_m / nprog) * c.bm : c.bm;  // one row per workgroup / per tile
  }
  int tn, kw;
  pw_pick(M, ldy, Ks, &tn, &kw);
  return (4 / kw) * 32;
}

extern "C" int vmtl_conv1x1_stats_rows(int M, int ldy, int Ks, int variant) {
  return cdiv(M, vmtl_conv1x1_stats_block(M, ldy, Ks, variant));
}

static int pw_num_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    n = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            ? prop.multiProcessorCount
            : 256;
  }
  return n;
}

template <int TM, int TN, int WM, int WN, int KC>
static int launch_pw_big(PwP& p, hipStream_t st) {
  constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
  constexpr int A_FLOATS = KC * BM * 32, O_FLOATS = BM * (BN + 4);
  constexpr int NTHR = WM * WN * 64;
  constexpr size_t lds = ((size_t)KC * BN * 32 + (A_FLOATS > O_FLOATS ? A_FLOATS : O_FLOATS)) * 4 + 2 * NTHR * 16;
  static_assert(lds <= 160 * 1024, "LDS budget");
  p.tiles_m = cdiv(p.M, BM);
  p.tiles_n = cdiv(p.ldy, BN);
  const BigCfg cfg = {0, BM, BN, KC};  // (LDS bytes in pw_big_nprog assume the 8-wave reduction scratch: an upper bound for 4 waves)
  const int nprog = pw_big_nprog(p.M, p.ldy, cfg);
  const dim3 grid(nprog, p.tiles_n);
#define VMTL_PW_BIG_LAUNCH(SRC2, PRO, EZ)                                                                               \
  {                                                                                                                     \
    static bool attr_set = false;                                                                                       \
    if (!attr_set) {                                                                                                    \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pw_big_kernel<TM, TN, WM, WN, KC, SRC2, PRO, EZ>),       \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                 \
      attr_set = true;                                                                                                  \
    }                                                                                                                   \
    hipLaunchKernelGGL((pw_big_kernel<TM, TN, WM, WN, KC, SRC2, PRO, EZ>), grid, dim3(NTHR), lds, st, p, nprog);        \
  }
  if (p.ez_x != nullptr) VMTL_PW_BIG_LAUNCH(false, false, true)
  else if (p.x2 != nullptr) VMTL_PW_BIG_LAUNCH(true, false, false)
  else if (p.pa != nullptr) VMTL_PW_BIG_LAUNCH(false, true, false)
  else VMTL_PW_BIG_LAUNCH(false, false, false)
#undef VMTL_PW_BIG_LAUNCH
  return vmtl_check_launch();
}

static int pw_big_dispatch(PwP& p, const BigCfg& c, hipStream_t st) {
  static EnvInt e_w8{"VMTL_PW_BIG_WAVES", 8};  // tuning aid: 4 = the 4-wave workgroups (one wave per SIMD)
  const bool w8 = env_int(e_w8) == 8;
  switch (c.id) {
    case 0:  // 64 x 128
      if (w8)
        return c.kc == 2 ? launch_pw_big<2, 2, 2, 4, 2>(p, st) : c.kc == 4 ? launch_pw_big<2, 2, 2, 4, 4>(p, st)
                                                                          : launch_pw_big<2, 2, 2, 4, 6>(p, st);
      return c.kc == 2 ? launch_pw_big<2, 4, 2, 2, 2>(p, st) : c.kc == 4 ? launch_pw_big<2, 4, 2, 2, 4>(p, st)
                                                                        : launch_pw_big<2, 4, 2, 2, 6>(p, st);
    case 1:  // 64 x 64
      if (w8) return c.kc == 4 ? launch_pw_big<2, 1, 2, 4, 4>(p, st) : launch_pw_big<2, 1, 2, 4, 8>(p, st);
      return c.kc == 4 ? launch_pw_big<2, 2, 2, 2, 4>(p, st) : launch_pw_big<2, 2, 2, 2, 8>(p, st);
    case 2:  // 64 x 32 (8 waves only)
      return launch_pw_big<1, 1, 4, 2, 4>(p, st);
    default:  // 64 x 96
      if (w8) return launch_pw_big<1, 3, 4, 2, 4>(p, st);
      return launch_pw_big<2, 3, 2, 2, 4>(p, st);
  }
}

template <int TN, int KW>
static int launch_pw(PwP& p, hipStream_t st) {
  p.tiles_m = cdiv(p.M, (4 / KW) * 32);
  p.tiles_n = cdiv(p.ldy, 16 * TN);
  if (p.x2 != nullptr)
    hipLaunchKernelGGL((pw_gemm_kernel<TN, KW, true>), dim3(p.tiles_m * p.tiles_n), dim3(256), 0, st, p);
  else if (p.pa != nullptr)
    hipLaunchKernelGGL((pw_gemm_kernel<TN, KW, false, true>), dim3(p.tiles_m * p.tiles_n), dim3(256), 0, st, p);
  else
    hipLaunchKernelGGL((pw_gemm_kernel<TN, KW>), dim3(p.tiles_m * p.tiles_n), dim3(256), 0, st, p);
  return vmtl_check_launch();
}

static void pw_plain(PwP& p) {  // no prologue, ordinary epilogue
  p.pa = p.pc = p.res = nullptr; p.a_out = nullptr; p.act_in = 0;
  p.ez_x = p.ez_mean = p.ez_invstd = p.ez_gamma = p.ez_beta = p.ez_add = nullptr; p.ez_act = 0;
}

static int pw_dispatch(PwP& p, hipStream_t st) {
  BigCfg c;
  if (p.ez_add == nullptr && p.res == nullptr && pw_big_cfg(p.M, p.ldy, p.Ks, &c)) return pw_big_dispatch(p, c, st);
  int tn, kw;
  pw_pick(p.M, p.ldy, p.Ks, &tn, &kw);
  if (tn == 2) return kw == 4 ? launch_pw<2, 4>(p, st) : launch_pw<2, 1>(p, st);
  return kw == 4 ? launch_pw<4, 4>(p, st) : launch_pw<4, 1>(p, st);
}

extern "C" int vmtl_conv1x1_fwd(const float* x, const float* wp, const float* bias, float* y, float* stats, int M, int Ks,
                                int ldy, int Nw, int Cout, void* stream) {
  VMTL_ENTER();
  if (!x || !wp || !y || M <= 0 || Ks <= 0 || (Ks & 3) || ldy <= 0 || (ldy & 3)) return VMTL_ERR_ARG;
  if (Nw <= 0 || Nw > ldy || Cout <= 0 || Cout > Nw) return VMTL_ERR_ARG;
  PwP p;
  p.x = x; p.wp = wp; p.bias = bias; p.y = y; p.stats = stats; p.M = M; p.Ks = Ks; p.ldy = ldy; p.Nw = Nw; p.Cout = Cout;
  p.x2 = nullptr; p.y2 = nullptr; p.K1 = 0; p.N1 = 0;
  pw_plain(p);
  return pw_dispatch(p, (hipStream_t)stream);
}

// conv1x1(cat[x, x2]) without the concat: x is [M][K1] (K1 % 4 == 0), x2 is [M][K2s], the packed weight rows are
// [Nw][K1 + K2s] (the ordinary packing of the (Nw, K1 + C2) weight).  Statistics geometry = vmtl_conv1x1_stats_*(M, ldy,
// K1 + K2s).
extern "C" int vmtl_conv1x1_cat_fwd(const float* x, int K1, const float* x2, int K2s, const float* wp, const float* bias,
                                    float* y, float* stats, int M, int ldy, int Nw, int Cout, void* stream) {
  VMTL_ENTER();
  if (!x || !x2 || !wp || !y || M <= 0 || K1 <= 0 || (K1 & 3) || K2s <= 0 || (K2s & 3) || ldy <= 0 || (ldy & 3))
    return VMTL_ERR_ARG;
  if (Nw <= 0 || Nw > ldy || Cout <= 0 || Cout > Nw) return VMTL_ERR_ARG;
  PwP p;
  p.x = x; p.wp = wp; p.bias = bias; p.y = y; p.stats = stats; p.M = M; p.Ks = K1 + K2s; p.ldy = ldy; p.Nw = Nw;
  p.Cout = Cout; p.x2 = x2; p.y2 = nullptr; p.K1 = K1; p.N1 = 0;
  pw_plain(p);
  return pw_dispatch(p, (hipStream_t)stream);
}

// its data gradient: [dx | dx2] = dy * W without a split pass: output columns [0, N1) (N1 % 4 == 0) land in dx
// ([M][N1]), columns [N1, N1 + N2s) in dx2 ([M][N2s]; columns past the N2 real ones are zero).  wp: [N1 + N2][Ks].
extern "C" int vmtl_conv1x1_cat_dgrad(const float* dy, const float* wp, float* dx, int N1, float* dx2, int N2s, int N2,
                                      int M, int Ks, void* stream) {
  VMTL_ENTER();
  if (!dy || !wp || !dx || !dx2 || M <= 0 || Ks <= 0 || (Ks & 3) || N1 <= 0 || (N1 & 3) || N2s <= 0 || (N2s & 3) ||
      N2 <= 0 || N2 > N2s)
    return VMTL_ERR_ARG;
  PwP p;
  p.x = dy; p.wp = wp; p.bias = nullptr; p.y = dx; p.stats = nullptr; p.M = M; p.Ks = Ks; p.ldy = N1 + N2s;
  p.Nw = N1 + N2; p.Cout = N1 + N2; p.x2 = nullptr; p.y2 = dx2; p.K1 = 0; p.N1 = N1;
  pw_plain(p);
  return pw_dispatch(p, (hipStream_t)stream);
}

// conv1x1(act(coef_a[k] * x + coef_c[k])): the BatchNorm + activation of the layer that produced x (coefficients from
// vmtl_bn_stats_coef; act in {none, relu, hardswish}: act(0) must be 0 on the pad channels) applied to the operand
// fragments; a_out (nullable, [M][Ks]) receives the activated matrix for the weight gradient.
static int conv1x1_bn_fwd_impl(const float* x, const float* coef_a, const float* coef_c, int act_in, const float* res,
                               float* a_out, const float* wp, const float* bias, float* y, float* stats, int M, int Ks,
                               int ldy, int Nw, int Cout, void* stream) {
  if (!x || !coef_a || !coef_c || !wp || !y || M <= 0 || Ks <= 0 || (Ks & 3) || ldy <= 0 || (ldy & 3)) return VMTL_ERR_ARG;
  if (Nw <= 0 || Nw > ldy || Cout <= 0 || Cout > Nw) return VMTL_ERR_ARG;
  if (act_in != VMTL_ACT_NONE && act_in != VMTL_ACT_RELU && act_in != VMTL_ACT_HSWISH) return VMTL_ERR_ARG;
  PwP p;
  p.x = x; p.wp = wp; p.bias = bias; p.y = y; p.stats = stats; p.M = M; p.Ks = Ks; p.ldy = ldy; p.Nw = Nw; p.Cout = Cout;
  p.x2 = nullptr; p.y2 = nullptr; p.K1 = 0; p.N1 = 0;
  pw_plain(p);
  p.pa = coef_a; p.pc = coef_c; p.res = res; p.a_out = a_out; p.act_in = act_in;
  return pw_dispatch(p, (hipStream_t)stream);
}

extern "C" int vmtl_conv1x1_bn_fwd(const float* x, const float* coef_a, const float* coef_c, int act_in, float* a_out,
                                   const float* wp, const float* bias, float* y, float* stats, int M, int Ks, int ldy,
                                   int Nw, int Cout, void* stream) {
  VMTL_ENTER();
  return conv1x1_bn_fwd_impl(x, coef_a, coef_c, act_in, nullptr, a_out, wp, bias, y, stats, M, Ks, ldy, Nw, Cout, stream);
}

// the same with a residual operand: the GEMM's input is a = act(coef_a*x + coef_c) + res (an inverted-residual
// block's bn3 output plus its skip connection, consumed by the next block's expand conv); a_out receives a.
extern "C" int vmtl_conv1x1_bn_res_fwd(const float* x, const float* coef_a, const float* coef_c, int act_in,
                                       const float* res, float* a_out, const float* wp, const float* bias, float* y,
                                       float* stats, int M, int Ks, int ldy, int Nw, int Cout, void* stream) {
  VMTL_ENTER();
  if (!res) return VMTL_ERR_ARG;
  return conv1x1_bn_fwd_impl(x, coef_a, coef_c, act_in, res, a_out, wp, bias, y, stats, M, Ks, ldy, Nw, Cout, stream);
}

// data gradient of a 1x1 conv whose input was act(BN(ez_x)), ending with that activation's and BatchNorm's backward:
// dz [M][ldy] = (dy * W) * act'(gamma * xhat + beta), stats [vmtl_conv1x1_stats_rows(M, ldy, Ks)][2][ldy] =
// per-row-block (sum dz, sum dz * xhat) for vmtl_bn_bwd_finalize / vmtl_bn_bwd_apply.
static int conv1x1_bnbwd_impl(const float* dy, const float* wp, const float* addend, float* dz, float* stats,
                              const float* ez_x, const float* ez_mean, const float* ez_invstd, const float* ez_gamma,
                              const float* ez_beta, int ez_act, int M, int Ks, int ldy, int Nw, int Cout, void* stream) {
  if (!dy || !wp || !dz || !stats || !ez_x || !ez_mean || !ez_invstd || M <= 0 || Ks <= 0 || (Ks & 3) || ldy <= 0 ||
      (ldy & 3))
    return VMTL_ERR_ARG;
  if (Nw <= 0 || Nw > ldy || Cout <= 0 || Cout > Nw) return VMTL_ERR_ARG;
  PwP p;
  p.x = dy; p.wp = wp; p.bias = nullptr; p.y = dz; p.stats = stats; p.M = M; p.Ks = Ks; p.ldy = ldy; p.Nw = Nw;
  p.Cout = Cout; p.x2 = nullptr; p.y2 = nullptr; p.K1 = 0; p.N1 = 0;
  pw_plain(p);
  p.ez_x = ez_x; p.ez_mean = ez_mean; p.ez_invstd = ez_invstd; p.ez_gamma = ez_gamma; p.ez_beta = ez_beta; p.ez_act = ez_act;
  p.ez_add = addend;
  return pw_dispatch(p, (hipStream_t)stream);
}

extern "C" int vmtl_conv1x1_bnbwd(const float* dy, const float* wp, float* dz, float* stats, const float* ez_x,
                                  const float* ez_mean, const float* ez_invstd, const float* ez_gamma,
                                  const float* ez_beta, int ez_act, int M, int Ks, int ldy, int Nw, int Cout,
                                  void* stream) {
  VMTL_ENTER();
  return conv1x1_bnbwd_impl(dy, wp, nullptr, dz, stats, ez_x, ez_mean, ez_invstd, ez_gamma, ez_beta, ez_act, M, Ks, ldy,
                            Nw, Cout, stream);
}

// the same with a second gradient of the differentiated tensor (the block's residual / skip consumers) added to the
// GEMM result before the activation's and BatchNorm's backward: dz = (dy*W + addend) * act'(...)
extern "C" int vmtl_conv1x1_bnbwd_add(const float* dy, const float* wp, const float* addend, float* dz, float* stats,
                                      const float* ez_x, const float* ez_mean, const float* ez_invstd,
                                      const float* ez_gamma, const float* ez_beta, int ez_act, int M, int Ks, int ldy,
                                      int Nw, int Cout, void* stream) {
  VMTL_ENTER();
  if (!addend) return VMTL_ERR_ARG;
  return conv1x1_bnbwd_impl(dy, wp, addend, dz, stats, ez_x, ez_mean, ez_invstd, ez_gamma, ez_beta, ez_act, M, Ks, ldy,
                            Nw, Cout, stream);
}
