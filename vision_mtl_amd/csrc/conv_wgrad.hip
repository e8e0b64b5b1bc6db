// Weight gradient of the dense convs (split out of conv_igemm.hip: its own translation unit builds in parallel).
#include <mutex>
#include <unordered_map>

#include "common.h"

// ---------------------------------------------------------------------------
// weight gradient:  slab[z][n][kk] = sum_{m in pixel slice z} dY[m][n] * Xcol[m][kk]
// A operand = dY rows (i = n), B operand = im2col(X) (j = kk), k = pixel.
// LDS tiles are [pixel][channel] exactly as they sit in HBM; a fragment read is a ds_read_b32 of 16 consecutive
// floats per lane quarter, quarter q reading pixel row 4s+q.  ds_read_b32 is served in two groups of 32 lanes over 32
// banks, so the quarters of a group (consecutive pixel rows) must land on disjoint bank ranges: the row stride is padded to
// == 16 (mod 64) floats (wg_ld below).  Round 2 used stride = width + 4 (== 4 or 20 mod 32): 12 (4) of the 16 banks
// of the two quarters overlapped - 0.38-0.41 of the LDS cycles were bank conflicts (profiles/r02_*_pmc.json).
// Each pixel slice
// writes its own slab with plain stores; vmtl_unpack_weights sums the slabs in a fixed
// order (deterministic, no float atomics) while converting to the torch layout.
// ---------------------------------------------------------------------------
struct WgradP {
  const float* x;   // [B][H][W][Cs]
  const float* dy;  // [B][Ho][Wo][ldy]
  float* slabs;     // [splits][Nw][Ktot]
  int B, H, W, Cs;
  int Ho, Wo, ldy;
  int Nw;
  int KH, KW, stride, pad;
  int Ktot, M;
  int chunk;        // pixels per z-slice (multiple of BP)
  int tiles_kk, tiles_co, splits;
  // virtual concat (1x1 only): channels [0, K1) of a pixel come from x (row stride K1), [K1, Cs) from x2 (row stride
  // Cs - K1); null = single source
  const float* x2;
  int K1;
};

#define BP 32
#define WG_BNK 128  // kk columns per workgroup (4 waves x 2 tiles x 16)

// smallest row stride >= w (floats, multiple of 4) that is == 16 (mod 64): the four pixel rows of one fragment read
// start 16 banks apart whether the hardware serves the 64 lanes over 64 banks at once or as two 32-lane groups over 32
// PM (tuning aid VMTL_WG_PAD, A/B on the GPU box): 0 = round 2's width + 4, 1 = == 16 (mod 64), 2 = == 16 (mod 32)
__host__ __device__ constexpr int wg_ld(int w, int pm = 1) {
  return pm == 0 ? w + 4 : pm == 1 ? ((w + 47) / 64) * 64 + 16 : ((w + 15) / 32) * 32 + 16;
}
static_assert(wg_ld(16) == 16 && wg_ld(20) == 80 && wg_ld(144) == 144 && wg_ld(128) == 144 && wg_ld(68) == 80 &&
              wg_ld(80) == 80 && wg_ld(84) == 144 && wg_ld(36, 2) == 48 && wg_ld(128, 2) == 144, "wg_ld");

// co rows per workgroup = TM * 16 (+ NTR "tail" rows: the 33rd / 17-20th / 65-68th output channel is not
// given an MFMA tile of its own - each lane multiplies its X fragment with the tail dY values on the VALU,
// the same trick as the tail columns of conv_igemm_kernel); waves are laid out 1 x 4 along kk
// PD = pixel chunks in flight in registers ahead of the one being multiplied.  A chunk of a NARROW tile is 16-64 MFMAs
// per wave (0.5-2 k cycles) against ~4-5 k cycles of loaded memory latency: with one chunk in flight (round 2) the 32- and
// 36-row tiles ran at 36 % of the matrix pipe (wgrad M = 1 M, N = 33: 51 TF; MTAN N = 32: 56 TF), latency-bound; the
// tall tiles (>= 80 rows: 160+ MFMAs per chunk) cover it with PD = 1 and have no registers to spare (the 68-row tile
// would drop from 3 to 2 waves per SIMD at PD = 2: 196 VGPRs).
// FAST (host: wgrad_fast_ok): one source, Wo % BP == 0 - the BP pixels of a chunk are consecutive pixels of ONE output
// row, so (image, row, first column) of a chunk are uniform and live in scalar registers: a gather address is
// scalar chunk base + per-thread constant, its bounds test one add + one compare.  The general loader tracks
// (b, ho, wo) per gather row in vector registers and rebuilds every offset with two quarter-rate multiplies: ~10
// non-MFMA instructions per MFMA on the 32-row tile (ISA count), which is what bounded the narrow tiles.
template <int TM, int NTR = 0, int PM = 1, int PD = (TM <= 2 ? 3 : (TM <= 4 && NTR == 0) ? 2 : 1), bool FAST = false>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradP p) {
  constexpr int TN = 2;
  constexpr int BMM = TM * 16;    // rows covered by MFMA tiles
  constexpr int BMC = BMM + NTR;  // + tail rows
  constexpr int LDY = wg_ld(BMC, PM);
  constexpr int LDX = wg_ld(WG_BNK, PM);
  constexpr int YQ = BMC / 4;                 // float4 per dY row
  constexpr int YIT = (BP * YQ + 255) / 256;  // loader iterations for the dY tile
  constexpr int XQ = WG_BNK / 4;              // 32 float4 per X row
  constexpr int XROWS = 256 / XQ;             // 8 rows per pass
  constexpr int XP = BP / XROWS;              // 4 passes

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ys = smem;                 // [2][BP][LDY]
  float* Xs = smem + 2 * BP * LDY;  // [2][BP][LDX]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wn = tid >> 6;
  const int l15 = lane & 15, lq = lane >> 4;

  // 1-D grid, XCD-aware: workgroups with consecutive remapped ids share an XCD (one L2); the kk/co
  // tiles of ONE pixel slice are consecutive, so the slice's x / dy rows are fetched from HBM once per
  // XCD instead of once per tile (measured before: 97 % L2 misses, 2.6 GB fabric traffic on blk4).
  const int tiles = p.tiles_kk * p.tiles_co;
  const int bid = xcd_remap(blockIdx.x, tiles * p.splits);
  const int zsl = bid / tiles, trem = bid - zsl * tiles;
  const int kk0 = (trem % p.tiles_kk) * WG_BNK;
  const int co0 = (trem / p.tiles_kk) * BMC;
  const int p_begin = zsl * p.chunk;
  const int p_end = min(p.M, p_begin + p.chunk);

  // X loader: fixed (tap, ci) column per thread, rows advance with the chunk
  const int xq = tid % XQ, xr = tid / XQ;
  const int kk = kk0 + xq * 4;
  const bool xok = kk < p.Ktot;
  const int tap = xok ? kk / p.Cs : 0;
  const int ci = kk - tap * p.Cs;
  const int dh = tap / p.KW - p.pad;
  const int dw = tap % p.KW - p.pad;
  const int hw = p.Ho * p.Wo;
  // two-source X (1x1): this thread's channel quad lives in one of the two maps; plain global loads (the source differs
  // between the lanes of a wave, a buffer descriptor cannot)
  const bool two = p.x2 != nullptr;
  const float* xsrc = two ? (ci < p.K1 ? p.x + ci : p.x2 + (ci - p.K1)) : nullptr;
  const int xstride = two ? (ci < p.K1 ? p.K1 : p.Cs - p.K1) : 0;

  // per-row pixel coordinates of this thread's XP gather rows, advanced by BP pixels per chunk with
  // adds / compares only (two integer divisions per row and chunk made this kernel VALU-issue bound:
  // 7 VALU instructions per MFMA on the narrow tiles)
  int xb[XP], xho[XP], xwo[XP];
  if (!FAST) {
#pragma unroll
    for (int i = 0; i < XP; ++i) {
      const int m = p_begin + xr + XROWS * i;
      const int b = m / hw;
      const int rem = m - b * hw;
      xb[i] = b;
      xho[i] = rem / p.Wo;
      xwo[i] = rem - xho[i] * p.Wo;
    }
  }
  // FAST: uniform chunk position (scalar registers) + per-thread constants
  int s_b = 0, s_ho = 0, s_wo = 0;
  int tw[XP];          // input column of gather row i relative to the chunk's first input column
  int toff[XP];        // byte offset of gather row i relative to the chunk's first input pixel (may be negative)
  int tyoff[YIT];      // dY: byte offset relative to the chunk's first dY row, -1 = this lane loads nothing
  if (FAST) {
    s_b = p_begin / hw;
    const int rem = p_begin - s_b * hw;
    s_ho = rem / p.Wo;
    s_wo = rem - s_ho * p.Wo;
#pragma unroll
    for (int i = 0; i < XP; ++i) {
      tw[i] = (xr + XROWS * i) * p.stride + dw;
      toff[i] = ((dh * p.W + tw[i]) * p.Cs + ci) * 4;
    }
#pragma unroll
    for (int it = 0; it < YIT; ++it) {
      const int idx = tid + it * 256;
      const int row = idx / YQ, q = idx - row * YQ;
      tyoff[it] = (idx < BP * YQ && co0 + q * 4 < p.ldy) ? (row * p.ldy + co0 + q * 4) * 4 : -1;
    }
  }

  f32x4 ry[PD][YIT], rx[PD][XP];
  // buffer loads: 32-bit offsets, out-of-range (slice end, image border, tile edge) reads return zeros
  const __amdgpu_buffer_rsrc_t rs_dy =
      __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, (int)((unsigned)p.M * p.ldy * 4u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_x =
      __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)((unsigned)p.B * p.H * p.W * p.Cs * 4u), 0x00020000);
  constexpr unsigned OOB = 0xFFFFFFFFu;
  auto bload = [](__amdgpu_buffer_rsrc_t r, unsigned off) -> f32x4 {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
  };
  // slot: compile-time index after unrolling (register arrays must not be indexed at run time)
  auto load_tile = [&](int pp, f32x4* ry_s, f32x4* rx_s) {
    if (FAST) {
      // every chunk is whole (M and the slice length are multiples of BP): no m < p_end test
      const unsigned ybase = (unsigned)pp * (unsigned)p.ldy * 4u;
#pragma unroll
      for (int it = 0; it < YIT; ++it) ry_s[it] = bload(rs_dy, tyoff[it] >= 0 ? ybase + (unsigned)tyoff[it] : OOB);
      const int hin = s_ho * p.stride;  // uniform: first input row / column of the chunk (before the tap shift)
      const int win = s_wo * p.stride;
      const unsigned xbase = (unsigned)((s_b * p.H + hin) * p.W + win) * (unsigned)p.Cs * 4u;
      // `&`, not `&&`: a short-circuit on a per-lane condition became divergent branches around duplicated loads
      const bool hok = xok & ((unsigned)(hin + dh) < (unsigned)p.H);
#pragma unroll
      for (int i = 0; i < XP; ++i) {
        const bool ok = hok & ((unsigned)(win + tw[i]) < (unsigned)p.W);
        rx_s[i] = bload(rs_x, ok ? xbase + (unsigned)toff[i] : OOB);
      }
      s_wo += BP;
      if (s_wo >= p.Wo) {
        s_wo = 0;
        if (++s_ho == p.Ho) {
          s_ho = 0;
          ++s_b;
        }
      }
      return;
    }
#pragma unroll
    for (int it = 0; it < YIT; ++it) {
      const int idx = tid + it * 256;
      const int row = idx / YQ, q = idx - row * YQ;
      const int m = pp + row, co = co0 + q * 4;
      const bool ok = idx < BP * YQ && m < p_end && co < p.ldy;
      ry_s[it] = bload(rs_dy, ok ? ((unsigned)m * (unsigned)p.ldy + (unsigned)co) * 4u : OOB);
    }
#pragma unroll
    for (int i = 0; i < XP; ++i) {
      const int m = pp + xr + XROWS * i;
      const int h = xho[i] * p.stride + dh, w = xwo[i] * p.stride + dw;
      const bool ok = xok && m < p_end && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;
      const unsigned off = ((unsigned)((xb[i] * p.H + h) * p.W + w) * (unsigned)p.Cs + (unsigned)ci) * 4u;
      f32x4 v;
      if (two) {
        v = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (ok) v = *reinterpret_cast<const f32x4*>(xsrc + (size_t)((xb[i] * p.H + h) * p.W + w) * xstride);
      } else {
        v = bload(rs_x, ok ? off : OOB);
      }
      rx_s[i] = v;
      // advance this row by BP pixels
      xwo[i] += BP;
      while (xwo[i] >= p.Wo) {
        xwo[i] -= p.Wo;
        if (++xho[i] == p.Ho) {
          xho[i] = 0;
          ++xb[i];
        }
      }
    }
  };
  auto store_tile = [&](int buf, const f32x4* ry_s, const f32x4* rx_s) {
    float* ys = Ys + buf * BP * LDY;
    float* xs = Xs + buf * BP * LDX;
#pragma unroll
    for (int it = 0; it < YIT; ++it) {
      const int idx = tid + it * 256;
      const int row = idx / YQ, q = idx - row * YQ;
      if (idx < BP * YQ) *reinterpret_cast<f32x4*>(ys + row * LDY + q * 4) = ry_s[it];
    }
#pragma unroll
    for (int i = 0; i < XP; ++i) *reinterpret_cast<f32x4*>(xs + (xr + XROWS * i) * LDX + xq * 4) = rx_s[i];
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float tacc[NTR > 0 ? NTR : 1][TN];  // tail rows: this lane quarter's pixels only
#pragma unroll
  for (int t = 0; t < (NTR > 0 ? NTR : 1); ++t)
#pragma unroll
    for (int j = 0; j < TN; ++j) tacc[t][j] = 0.f;

  const int ntr = NTR > 0 ? max(0, min(NTR, p.Nw - (co0 + BMM))) : 0;
  // chunks pp = p_begin + it * BP; chunk it lives in register slot it % PD until it is stored to LDS buffer it & 1
#pragma unroll
  for (int d = 0; d < PD; ++d)
    if (p_begin + d * BP < p_end) load_tile(p_begin + d * BP, ry[d], rx[d]);
  if (p_begin < p_end) store_tile(0, ry[0], rx[0]);
  __syncthreads();
  int cur = 0;
  // 16-column tiles of this wave that hold real kk columns (the last kk tile of Ktot = 288 has 32 of its 128: three
  // of its four waves would multiply zeros - a quarter of ALL the MFMAs of a 32-channel 3x3 layer); wave-uniform
  const int nj = min(TN, max(0, (p.Ktot - (kk0 + wn * TN * 16) + 15) >> 4));
  auto compute = [&](int cb) {
    if (nj == 0) return;
    const float* ys = Ys + cb * BP * LDY + lq * LDY + l15;
    const float* xs = Xs + cb * BP * LDX + lq * LDX + wn * TN * 16 + l15;
#pragma unroll
    for (int s = 0; s < BP / 4; ++s) {  // 4 pixels per MFMA: lane quarter q supplies pixel 4s+q
      float fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i] = ys[4 * s * LDY + i * 16];
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[j] = xs[4 * s * LDX + j * 16];
#ifdef VMTL_SETPRIO
      __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if (j > 0 && j >= nj) break;
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
      }
#ifdef VMTL_SETPRIO
      __builtin_amdgcn_s_setprio(0);
#endif
      if (NTR > 0) {  // dY[pixel 4s+lq][BMM .. BMM+3]: one 16-byte LDS read, broadcast within the quarter
        const f32x4 ty = *reinterpret_cast<const f32x4*>(Ys + cb * BP * LDY + (4 * s + lq) * LDY + BMM);
#pragma unroll
        for (int t = 0; t < NTR; ++t) {
          if (t >= ntr) break;  // dY columns past Nw are zero pad lanes
#pragma unroll
          for (int j = 0; j < TN; ++j) tacc[t][j] += ty[t] * fb[j];
        }
      }
    }
  };
  for (int base = p_begin; base < p_end; base += PD * BP) {
#pragma unroll
    for (int k = 0; k < PD; ++k) {  // unrolled: slot indices k and (k + 1) % PD are compile-time constants
      const int pp = base + k * BP;
      if (pp < p_end) {  // uniform over the workgroup
        // slot k held chunk pp: it went to LDS one step ago (or in the prologue); refill it PD chunks ahead
        if (pp + PD * BP < p_end) load_tile(pp + PD * BP, ry[k], rx[k]);
        compute(cur);
        if (pp + BP < p_end) store_tile(cur ^ 1, ry[(k + 1) % PD], rx[(k + 1) % PD]);
        __syncthreads();
        cur ^= 1;
      }
    }
  }

  float* slab = p.slabs + (size_t)zsl * p.Nw * p.Ktot;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = kk0 + (wn * TN + j) * 16 + l15;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = co0 + i * 16 + 4 * lq + r;
        if (row < p.Nw && col < p.Ktot) slab[(size_t)row * p.Ktot + col] = acc[i][j][r];
      }
    }
  if (NTR > 0) {
#pragma unroll
    for (int t = 0; t < NTR; ++t)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const float v = quarter_sum(tacc[t][j]);
        const int row = co0 + BMM + t;
        const int col = kk0 + (wn * TN + j) * 16 + l15;
        if (lq == 0 && row < p.Nw && col < p.Ktot) slab[(size_t)row * p.Ktot + col] = v;
      }
  }
}

// ---- weight gradient ----
static int wgrad_rows(int Nw) {
  // co rows per workgroup (TM*16).  Cost = padded rows / relative efficiency of that tile height:
  // a 16- or 32-row tile issues 3-4 LDS reads per 2-4 MFMAs and loses to a taller, slightly more
  // padded one (measured: Nw=270 as 17x16 rows ran at 42 TF, as 2x144 at ~90 TF).
  // 20 / 36 / 68 = 16 / 32 / 64 MFMA rows + 4 tail rows on the VALU
  // 128 rows (round 3): N = 128 / 256 / 512 (MTAN's widths) padded to 144-row tiles wasted 11 % of the MFMAs
  static const int cands[] = {16, 32, 48, 64, 80, 144, 20, 36, 68, 128};
  static const float eff[] = {0.35f, 0.55f, 0.72f, 0.82f, 0.88f, 1.0f, 0.43f, 0.61f, 0.86f, 0.97f};
  constexpr int NC = 10;
  static EnvInt force{"VMTL_FORCE_WG_ROWS", 0};  // tuning aid
  if (const int v = env_int(force)) {
    for (int i = 0; i < NC; ++i)
      if (cands[i] == v) return v;
  }
  int best = 16;
  float bc = -1.f;
  for (int i = 0; i < NC; ++i) {
    const float cost = (float)((long long)cdiv(Nw, cands[i]) * cands[i]) / eff[i];
    if (bc < 0.f || cost < bc) {
      best = cands[i];
      bc = cost;
    }
  }
  return best;
}

// LDS bytes of one weight-gradient workgroup under the stride rule its tile height gets (wg_pm)
static int wgrad_lds_bytes(int rows) {
  auto lds = [&](int pm) { return 2 * BP * (wg_ld(rows, pm) + wg_ld(WG_BNK, pm)) * 4; };
  const int base = 160 * 1024 / lds(0);
  if (160 * 1024 / lds(1) >= base) return lds(1);
  if (160 * 1024 / lds(2) >= base) return lds(2);
  return lds(0);
}

static int wgrad_splits_uncached(int M, int Nw, int Ktot);

// number of pixel slices (= slabs the caller must provide: splits * Nw * Ktot floats); memoised: the search below walks
// up to a few thousand candidates, and this is called twice per weight-gradient launch on the (eager) launch path
extern "C" int vmtl_conv2d_wgrad_splits(int M, int Nw, int Ktot) {
  if (M <= 0 || Nw <= 0 || Ktot <= 0) return 0;
  static std::mutex mu;
  static std::unordered_map<unsigned long long, int> memo;
  static int memo_epoch = 0;
  const unsigned long long key = ((unsigned long long)M << 32) ^ ((unsigned long long)Nw << 20) ^ (unsigned long long)Ktot;
  std::lock_guard<std::mutex> lk(mu);
  if (memo_epoch != vmtl_env_epoch) {  // the tuning overrides may have changed
    memo.clear();
    memo_epoch = vmtl_env_epoch;
  }
  auto it = memo.find(key);
  if (it != memo.end()) return it->second;
  const int v = wgrad_splits_uncached(M, Nw, Ktot);
  memo.emplace(key, v);
  return v;
}

static int wgrad_splits_uncached(int M, int Nw, int Ktot) {
  const long long tiles = (long long)cdiv(Ktot, WG_BNK) * cdiv(Nw, wgrad_rows(Nw));
  // Pixels per slice: at least 16 K-steps (512) - except for SMALL problems (<= 16384 pixels: the deep encoder layers,
  // whose tile grid is a few dozen workgroups): there 4 K-steps per slice, parallelism over the chip beats the longer
  // slab sum (the 1x1 weight gradients at M = 1024 / 4096 ran 45-60 us on 28-72 workgroups; basic bs32 14.4 -> 14.0
  // ms/step).  Extending the rule to 65536 / all sizes measured +0.05..0.1 ms (VMTL_WG_SMALL_M).
  static EnvInt e_steps{"VMTL_WG_MIN_STEPS", 4}, e_small{"VMTL_WG_SMALL_M", 16384}, e_splits{"VMTL_FORCE_WG_SPLITS", 0};  // tuning aids
  const int min_steps = env_int(e_steps) < 1 ? 1 : env_int(e_steps), small_m = env_int(e_small);
  const long long max_by_rows = cdiv(M, (M <= small_m ? min_steps : 16) * BP);
  const long long max_by_mem = (32ll << 20) / ((long long)Nw * Ktot);  // slabs <= 128 MB
  long long smax = max_by_rows < max_by_mem ? max_by_rows : max_by_mem;
  if (smax < 1) smax = 1;
  if (const long long v = env_int(e_splits)) {
    if (v >= 1 && v <= max_by_rows) {
      const int chunk = cdiv(cdiv(M, (int)v), BP) * BP;
      return cdiv(M, chunk);
    }
  }
  // Wave quantisation: the grid is tiles x slices workgroups on 256 CUs x (workgroups per CU) slots and runs in whole
  // ROUNDS - 1280 workgroups on 512 slots take three rounds of 512-pixel slices where 980 workgroups of 672 pixels take
  // two (decoder block 2: 228 us at 2.5 rounds).  Pick the slice count that minimises
  //   rounds * (pixels per slice + fixed cost of a workgroup) * time per pixel  +  slices * slab bytes / unpack rate
  // (round 2 took 1536 / tiles slices whatever the remainder).
  const int rows = wgrad_rows(Nw);
  int occ = 160 * 1024 / wgrad_lds_bytes(rows);
  if (occ > 4) occ = 4;
  if (rows >= 128 && occ > 2) occ = 2;
  const long long slots = 256ll * occ;
  const double t_pix = (rows / 16.0) * occ * 0.0095;             // us per pixel of a round (MFMA-paced, measured ~0.7 eff)
  const double t_slab = (double)Nw * Ktot * 4.0 / 3.0e6;          // us per slab in the slab sum (vmtl_unpack_weights)
  long long best = 1;
  double best_cost = -1.0;
  for (long long sp = 1; sp <= smax; ++sp) {
    const int chunk = cdiv(cdiv(M, (int)sp), BP) * BP;
    const long long nsl = cdiv(M, chunk);
    if (nsl != sp && sp != 1) continue;  // slice counts that collapse onto another one
    const long long rounds = cdivll(tiles * nsl, slots);
    const double cost = rounds * (chunk + 64.0) * t_pix + nsl * t_slab;
    if (best_cost < 0.0 || cost < best_cost) {
      best = nsl;
      best_cost = cost;
    }
  }
  return (int)best;
}

template <int TM, int NTR, int PM, bool FAST>
static int launch_wgrad_pm(WgradP& p, int splits, hipStream_t st);

// Stride rule per tile height.  Measured on MI355X (tools/bench_conv.py, VMTL_WG_PAD A/B, round 3): the conflict-free
// strides remove ALL bank-conflict cycles (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.425 -> 0.000 on the 144-row tile)
// but change the kernel's time by < 1 % - it is not LDS-bound - while every workgroup per CU lost to the larger tiles
// costs 10-20 % (68 rows: 517 -> 589 us, 36 rows: 562 -> 665 us).  So: the conflict-free stride wherever it keeps the
// workgroups per CU (160 KB LDS), round 2's width + 4 elsewhere.
template <int BMC>
constexpr int wg_pm() {
  constexpr int base = 160 * 1024 / (2 * BP * (wg_ld(BMC, 0) + wg_ld(WG_BNK, 0)) * 4);
  return 160 * 1024 / (2 * BP * (wg_ld(BMC, 1) + wg_ld(WG_BNK, 1)) * 4) >= base   ? 1
         : 160 * 1024 / (2 * BP * (wg_ld(BMC, 2) + wg_ld(WG_BNK, 2)) * 4) >= base ? 2
                                                                                    : 0;
}
static_assert(wg_pm<144>() == 1 && wg_pm<80>() == 1 && wg_pm<36>() == 2 && wg_pm<68>() == 0 && wg_pm<20>() == 0 &&
                  wg_pm<128>() == 1,
              "wg_pm");

// scalar-chunk loader (conv_wgrad_kernel FAST): one source, whole chunks of one output row
static bool wgrad_fast_ok(const WgradP& p) {
  static EnvInt e{"VMTL_WG_FAST", 1};  // tuning aid: 0 = the general loader everywhere
  return env_int(e) != 0 && p.x2 == nullptr && p.Wo % BP == 0 && p.chunk % BP == 0;
}

template <int TM, int NTR = 0>
static int launch_wgrad(WgradP& p, int splits, hipStream_t st) {
  constexpr int PMD = wg_pm<TM * 16 + NTR>();
#ifdef VMTL_TUNING
  static EnvInt e_pad{"VMTL_WG_PAD", -1};  // force one LDS row-stride rule (wg_ld) for every tile height
  switch (env_int(e_pad)) {
    case 0: return launch_wgrad_pm<TM, NTR, 0, false>(p, splits, st);
    case 1: return launch_wgrad_pm<TM, NTR, 1, false>(p, splits, st);
    case 2: return launch_wgrad_pm<TM, NTR, 2, false>(p, splits, st);
    default: break;
  }
#endif
  if (p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0 && p.x2 == nullptr && p.M % BP == 0) {
    // a pointwise conv has no borders: one flat row of M pixels (so any image width takes the scalar-chunk loader)
    p.B = 1; p.H = 1; p.W = p.M; p.Ho = 1; p.Wo = p.M;
  }
  if (wgrad_fast_ok(p)) return launch_wgrad_pm<TM, NTR, PMD, true>(p, splits, st);
  return launch_wgrad_pm<TM, NTR, PMD, false>(p, splits, st);
}

template <int TM, int NTR, int PM, bool FAST>
static int launch_wgrad_pm(WgradP& p, int splits, hipStream_t st) {
  constexpr int BMC = TM * 16 + NTR;
  constexpr int PD = (TM <= 2 ? 3 : (TM <= 4 && NTR == 0) ? 2 : 1);
  const size_t lds = (size_t)2 * BP * (wg_ld(BMC, PM) + wg_ld(WG_BNK, PM)) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_kernel<TM, NTR, PM, PD, FAST>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  if ((long long)p.M * p.ldy * 4 >= (1ll << 32) || (long long)p.B * p.H * p.W * p.Cs * 4 >= (1ll << 32))
    return VMTL_ERR_UNSUPPORTED;  // buffer addressing: 32-bit byte offsets
  p.tiles_kk = cdiv(p.Ktot, WG_BNK);
  p.tiles_co = cdiv(p.Nw, BMC);
  p.splits = splits;
  hipLaunchKernelGGL((conv_wgrad_kernel<TM, NTR, PM, PD, FAST>), dim3(p.tiles_kk * p.tiles_co * splits), dim3(256), lds, st,
                     p);
  return vmtl_check_launch();
}

static int wgrad_dispatch(WgradP& p, int splits, void* stream);

extern "C" int vmtl_conv2d_wgrad(const float* x, const float* dy, float* slabs, int splits, int B, int H, int W,
                                 int Cs, int Ho, int Wo, int ldy, int Nw, int KH, int KW, int stride, int pad,
                                 void* stream) {
  VMTL_ENTER();
  if (!x || !dy || !slabs) return VMTL_ERR_ARG;
  if (Cs <= 0 || (Cs & 3) || (ldy & 3) || Nw <= 0 || Nw > ldy) return VMTL_ERR_ARG;
  if ((H + 2 * pad - KH) / stride + 1 != Ho || (W + 2 * pad - KW) / stride + 1 != Wo) return VMTL_ERR_ARG;
  if ((long long)B * Ho * Wo > 0x7fffffffLL || (long long)B * H * W > 0x7fffffffLL) return VMTL_ERR_ARG;
  WgradP p;
  p.x = x; p.dy = dy; p.slabs = slabs; p.B = B; p.H = H; p.W = W; p.Cs = Cs; p.Ho = Ho; p.Wo = Wo; p.ldy = ldy;
  p.Nw = Nw; p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad; p.Ktot = KH * KW * Cs; p.M = B * Ho * Wo;
  p.x2 = nullptr; p.K1 = 0;
  return wgrad_dispatch(p, splits, stream);
}

// weight gradient of conv1x1(cat[x, x2]) (vmtl_conv1x1_cat_fwd) in one launch: slabs [splits][Nw][K1 + K2s] with
// splits = vmtl_conv2d_wgrad_splits(M, Nw, K1 + K2s); x [M][K1] (K1 % 4 == 0), x2 [M][K2s]
extern "C" int vmtl_conv1x1_cat_wgrad(const float* x, int K1, const float* x2, int K2s, const float* dy, float* slabs,
                                      int splits, int M, int ldy, int Nw, void* stream) {
  VMTL_ENTER();
  if (!x || !x2 || !dy || !slabs || M <= 0 || K1 <= 0 || (K1 & 3) || K2s <= 0 || (K2s & 3) || (ldy & 3) || Nw <= 0 ||
      Nw > ldy)
    return VMTL_ERR_ARG;
  WgradP p;
  p.x = x; p.dy = dy; p.slabs = slabs; p.B = 1; p.H = 1; p.W = M; p.Cs = K1 + K2s; p.Ho = 1; p.Wo = M; p.ldy = ldy;
  p.Nw = Nw; p.KH = 1; p.KW = 1; p.stride = 1; p.pad = 0; p.Ktot = p.Cs; p.M = M;
  p.x2 = x2; p.K1 = K1;
  return wgrad_dispatch(p, splits, stream);
}

static int wgrad_dispatch(WgradP& p, int splits, void* stream) {
  const int Nw = p.Nw;
  if (splits != vmtl_conv2d_wgrad_splits(p.M, Nw, p.Ktot)) return VMTL_ERR_ARG;
  p.chunk = cdiv(cdiv(p.M, splits), BP) * BP;
  hipStream_t st = (hipStream_t)stream;
  switch (wgrad_rows(Nw)) {
    case 16: return launch_wgrad<1>(p, splits, st);
    case 32: return launch_wgrad<2>(p, splits, st);
    case 48: return launch_wgrad<3>(p, splits, st);
    case 64: return launch_wgrad<4>(p, splits, st);
    case 80: return launch_wgrad<5>(p, splits, st);
    case 20: return launch_wgrad<1, 4>(p, splits, st);
    case 36: return launch_wgrad<2, 4>(p, splits, st);
    case 68: return launch_wgrad<4, 4>(p, splits, st);
    case 128: return launch_wgrad<8>(p, splits, st);
    default: return launch_wgrad<9>(p, splits, st);
  }
}
