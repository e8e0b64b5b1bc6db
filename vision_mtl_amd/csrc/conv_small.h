// 3x3 / stride 1 / pad 1 convolution for the NARROW full-resolution layers (<= 36 storage channels in, <= 36 out):
// the last decoder block and the two heads of the `basic` model (reference vision_mtl/models/basic_model.py:30-51,
// smp DecoderBlock conv2 reached from utils/model_utils.py:25-34), their data gradients, and the 32-channel
// full-resolution convs of MTAN (reference models/mtan_model.py:44,119,127; utils/model_utils.py:71,74).
//
// Why a second conv kernel: with N <= 36 output columns the implicit-GEMM kernel (conv_igemm.hip) re-stages every
// input row nine times (once per tap) for only 32 MFMA columns - it is bound by the A-operand path, not by the
// matrix pipe (58-66 TF).  Here a workgroup owns a 4 x 32 pixel output tile:
//   * the (4+2) x (32+2) input halo is loaded ONCE, transformed (see prologue) and kept in LDS for all nine taps;
//   * the whole packed weight matrix (<= 36 x 9 x 36 floats) sits in LDS for the lifetime of the (persistent) workgroup;
//   * the K loop has no barrier and no global->LDS traffic: each wave owns one output row (2 x 16 pixels) and issues
//     v_mfma_f32_16x16x4_f32 straight from ds_read_b128 fragments; the next tile's halo is prefetched into
//     registers under the MFMAs.
// LDS layouts are "slot-major": halo[channel quad][pixel][4] and w[k quad][row][4].  The weight rows extent is a
// multiple of 16, so the four 16-lane groups of a ds_read_b128 (MI355X_MICROARCH.md, LDS table) each touch 16
// distinct 16-byte slots: conflict-free without a swizzle.  The halo's pixel extent is 209 (odd): the staging
// ds_write_b128 of 8 consecutive lanes (8 channel quads of one pixel, coalesced in HBM) then land on 8 distinct
// slots (a multiple of 8 would make every store 8-way conflicted, 2 of 13 k cycles of each tile); the price is one
// 2-way conflict per 16-lane group of the A fragment reads (2 of the 8 ds_read_b128 of a k-group; LDS has slack).
// After the K loop the halo region is reused for the output tile so that y (and the mode-2 operand) move as
// coalesced float4 rows instead of 64-byte pieces of the MFMA C layout.
//
// K order: per tap the Cs/4 channel quads are consumed four at a time (one per lane quarter); the Cs/4 % 4 left-over
// quads of the nine taps are gathered into shared k-groups (36 channels: 18 full groups + 3 groups for the nine
// left-over quads = 21 groups of 16 k instead of 27), the same order on the weight side.
//
// Fusions (what the reference runs as separate BatchNorm2d / ReLU kernels, utils/model_utils.py:72-76):
//   prologue  v = act(pa[c] * x + pb[c] * x2 + pc[c]) applied once per halo element: BatchNorm-apply + ReLU of the
//             producer (x2 = null), or the BatchNorm-backward apply dx = A*dz + B*x + C (two operands); the
//             transformed interior can be written back (a_out) for the weight-gradient kernel;
//   epilogue  mode 1: per-tile BatchNorm partials (mean, M2) of the output (as conv_igemm.hip);
//             mode 2: BatchNorm + activation backward of the PRODUCER of the output tensor: dz = acc * act'(z(xz)),
//                     stored instead of acc, plus per-tile (sum dz, sum dz * xhat);
//   store     NHWC [B][H][W][ldy], or the reference's NCHW split into two tensors (the two heads' logits).
#pragma once
#include <stdlib.h>

#include "common.h"

#define CSM_TH 4
#define CSM_TW 32
#define CSM_HX (CSM_TW + 2)
#define CSM_NHALO ((CSM_TH + 2) * CSM_HX)  // 204 halo pixels
#define CSM_NPIX 209                       // slot-major stride (pixels per channel quad), see the layout note
#define CSM_OT4 1152                       // float4 of the output tile image: 128 pixels x (<= 36 floats)
#define CSM_RED4 512                       // float4 of the statistics scratch: [2][4 waves][64 lanes]

// ablation switches of the kernel below (skip stores / staging / MFMAs: WRONG results) exist only in a -DVMTL_TUNING build
#ifdef VMTL_TUNING
#define CSM_DBG(p) ((p).dbg)
#else
#define CSM_DBG(p) 0
#endif

struct SmallP {
  const float* x;    // [B][H][W][CS]
  const float* x2;   // optional second prologue operand, same shape
  const float* pa;   // [CS] per-channel prologue coefficients (null: identity prologue)
  const float* pb;
  const float* pc;
  float* a_out;      // optional: transformed input, same shape as x
  const float* wp;   // [Nw][9*CS] packed ([row][tap*CS + c])
  const float* bias; // [Nw] or null
  float* y;          // NHWC [B][H][W][ldy], or NCHW [B][Ca][H][W] when yb != null
  float* yb;         // NCHW [B][Cout-Ca][H][W] (split store) or null
  float* stats;      // [ntiles][2][ldy] (ep_mode 1, 2)
  const float* ez_x; // ep_mode 2: pre-BatchNorm activation of the producer of y's tensor, [B][H][W][ldy]
  const float* ez_mean;
  const float* ez_invstd;
  const float* ez_gamma;
  const float* ez_beta;
  int act_in, ep_mode, ez_act, Ca;
  int B, H, W, ldy, Nw, Cout;
  int tiles_x, tiles_y, ntiles;
  int grid;      // workgroups (persistent: each walks tiles grid apart)
  int acc_rows;  // statistics: 1 = one row per WORKGROUP (accumulated over its tiles), 0 = one row per tile
  int dbg;  // tuning aid (VMTL_SMALL_DBG): 1 skip the output stores, 2 skip halo staging, 4 skip the MFMA loop, 8 no start skew
};

// Workgroup barrier for LDS hand-overs only.  __syncthreads() is also a global-memory fence: it drains vmcnt, i.e.
// waits for every outstanding global STORE (the previous tile's output rows, the a_out rows) and prefetch load at
// each of the five barriers of a tile - measured as 26 % of the wave cycles parked, matrix pipe 69 % busy at two
// waves per SIMD.  Nothing another wave reads through global memory is produced inside the tile loop, so the
// barrier only has to order LDS traffic: wait for this wave's LDS operations, then s_barrier.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// per-CU arrival tickets (index: XCC id << 8 | SE/SH/CU id), see the start-skew note in the kernel
static __device__ int g_small_cu_ticket[4096];

template <int CS, int TN, int NT>
struct SmallCfg {
  static constexpr int SP = CS / 4;         // channel quads per pixel
  static constexpr int FG = SP / 4;         // full k-groups per tap
  static constexpr int RS = SP % 4;         // left-over quads per tap
  static constexpr int KS = 9 * SP;         // k quads of the weight matrix
  static constexpr int NGF = 9 * FG;
  static constexpr int NR = 9 * RS;
  static constexpr int NGR = (NR + 3) / 4;
  static constexpr int NROWS = 16 * TN;     // weight rows fed to the MFMAs
  static constexpr int NCAP = NROWS + NT;   // + tail rows dotted on the VALU (0, 1 or 4)
  static constexpr int OS = NROWS + 4;      // floats per pixel of the LDS output tile: 36 / 20 = 4 (mod 8)
  static constexpr int NST = CSM_NHALO * SP;        // float4 elements of one halo
  static constexpr int IT = (NST + 255) / 256;      // staging iterations per thread
  // LDS image in float4 units
  static constexpr int HALO_NEED = SP * CSM_NPIX + 1;   // + one zero quad for the lanes of a partial k-group
  static constexpr int HALO4 = HALO_NEED > CSM_OT4 + CSM_RED4 ? HALO_NEED : CSM_OT4 + CSM_RED4;
  static constexpr int WM4 = (KS + 1) * NROWS;      // + one zero k quad
  static constexpr int WT4 = (KS + 1) * 4;
  static constexpr int COEF4 = 3 * SP + 5 * 9;  // prologue coefficients + epilogue per-channel parameters
  static constexpr int LDS_BYTES = (HALO4 + WM4 + WT4 + COEF4) * 16;
};

template <int CS, int TN, int NT, bool X2>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv3x3_small_kernel(SmallP p) {
  using C = SmallCfg<CS, TN, NT>;
  constexpr int SP = C::SP, FG = C::FG, RS = C::RS, KS = C::KS, NGF = C::NGF, NR = C::NR, NGR = C::NGR;
  constexpr int NROWS = C::NROWS, NCAP = C::NCAP, NST = C::NST, IT = C::IT;
  constexpr int TM = 2;
  constexpr int NTT = NT > 0 ? NT : 1;
  constexpr int OS = C::OS;            // LDS output-tile row stride (floats): conflict-free ds_write_b32 from the C layout
  constexpr int SQ = OS / 4;           // channel quads per output pixel handled by the epilogue lanes (9 / 5)
  constexpr int PPI = 64 / SQ;         // pixels per epilogue pass of a wave (7 / 12)
  constexpr int EIT = (CSM_TW + PPI - 1) / PPI;  // passes over the wave's 32 pixels (5 / 3)

  extern __shared__ __attribute__((aligned(16))) f32x4 smem4[];
  f32x4* halo = smem4;                  // [SP][NPIX] (+ zero quad at SP*NPIX); after the K loop: output tile + scratch
  f32x4* wm = halo + C::HALO4;          // [KS+1][NROWS]
  f32x4* wt = wm + C::WM4;              // [KS+1][4]
  f32x4* coef = wt + C::WT4;            // [3][SP]
  float* otile = reinterpret_cast<float*>(halo);  // [128 pixels][OS]
  f32x4* red = halo + CSM_OT4;                    // [2][4][64]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
  const int l15 = lane & 15, lq = lane >> 4;

  // ---- one-time: weights and prologue coefficients into LDS ----
  for (int idx = tid; idx < NCAP * (KS + 1); idx += 256) {
    const int n = idx / (KS + 1), ks = idx - n * (KS + 1);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (ks < KS && n < p.Nw) v = *reinterpret_cast<const f32x4*>(p.wp + ((size_t)n * KS + ks) * 4);
    if (n < NROWS) wm[ks * NROWS + n] = v;
    else wt[ks * 4 + (n - NROWS)] = v;
  }
  if (tid < 3 * SP) {
    const int which = tid / SP, s = tid - which * SP;
    const float* src = which == 0 ? p.pa : (which == 1 ? p.pb : p.pc);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (src != nullptr) v = *reinterpret_cast<const f32x4*>(src + 4 * s);
    coef[tid] = v;
  }
  const bool has_pro = p.pa != nullptr;

  // ---- fragment addresses (float4 units) ----
  const int a_base = lq * CSM_NPIX + wv * CSM_HX + l15;  // + g constants: cb*NPIX + dh*HX + dw
  const int b_base = lq * NROWS + l15;                   // + (tap*SP + cb) * NROWS
  int a_rem[NGR > 0 ? NGR : 1], ks_rem[NGR > 0 ? NGR : 1];
#pragma unroll
  for (int h = 0; h < NGR; ++h) {
    const int r = 4 * h + lq;
    if (r < NR) {
      const int tap = r / (RS > 0 ? RS : 1), cs = FG * 4 + r % (RS > 0 ? RS : 1);
      a_rem[h] = cs * CSM_NPIX + (wv + tap / 3) * CSM_HX + tap % 3 + l15;
      ks_rem[h] = tap * SP + cs;
    } else {
      a_rem[h] = -1;  // zero quad (same address for the 16 lanes: broadcast)
      ks_rem[h] = KS;
    }
  }

  // ---- epilogue geometry: lane <-> (pixel of the wave's output row, channel quad): coalesced float4 I/O ----
  const int eq = lane % SQ, epl = lane / SQ;
  const bool elane = lane < PPI * SQ && 4 * eq < p.ldy;
  // per-channel epilogue parameters (bias; mean, invstd, gamma, beta of mode 2) as float4 per channel quad in LDS
  f32x4* epar = coef + 3 * SP;  // [5][9]
  if (tid < 45) {
    const int which = tid / 9, q = tid - which * 9;
    const float* src = which == 0 ? p.bias : (which == 1 ? p.ez_mean : (which == 2 ? p.ez_invstd : (which == 3 ? p.ez_gamma : p.ez_beta)));
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (src != nullptr && (which == 0 || p.ep_mode == 2)) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (4 * q + e < p.Cout) v[e] = src[4 * q + e];
    }
    epar[tid] = v;
  }

  const int tiles_per_img = p.tiles_x * p.tiles_y;
  f32x4 rx[IT], rx2[X2 ? IT : 1];
  unsigned okmask = 0;

  auto tile_origin = [&](int t, int& b, int& h0, int& w0) {
    b = t / tiles_per_img;
    const int rem = t - b * tiles_per_img;
    const int ty = rem / p.tiles_x;
    h0 = ty * CSM_TH;
    w0 = (rem - ty * p.tiles_x) * CSM_TW;
  };
  auto prefetch = [&](int b, int h0, int w0) {
    okmask = 0;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int f = tid + it * 256;
      const int pp = f / SP, s = f - pp * SP;
      const int hy = pp / CSM_HX, hx = pp - hy * CSM_HX;
      const int gh = h0 - 1 + hy, gw = w0 - 1 + hx;
      const bool ok = f < NST && (unsigned)gh < (unsigned)p.H && (unsigned)gw < (unsigned)p.W;
      rx[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (X2) rx2[X2 ? it : 0] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (ok) {
        const unsigned off = ((unsigned)((b * p.H + gh) * p.W + gw)) * CS + 4 * s;  // host: B*H*W*36 < 2^31
        rx[it] = *reinterpret_cast<const f32x4*>(p.x + off);
        if (X2) rx2[X2 ? it : 0] = *reinterpret_cast<const f32x4*>(p.x2 + off);
        okmask |= 1u << it;
      }
    }
  };
  auto stage_store = [&](int b, int h0, int w0) {
    if (tid == 0) halo[SP * CSM_NPIX] = (f32x4){0.f, 0.f, 0.f, 0.f};  // the region was the previous output tile
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int f = tid + it * 256;
      if (f >= NST) continue;
      const int pp = f / SP, s = f - pp * SP;
      const bool ok = (okmask >> it) & 1u;
      f32x4 v = rx[it];
      if (has_pro) {
        v = v * coef[s] + coef[2 * SP + s];
        if (X2) v += rx2[X2 ? it : 0] * coef[SP + s];
        if (p.act_in == VMTL_ACT_RELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if (!ok) v = (f32x4){0.f, 0.f, 0.f, 0.f};  // conv zero padding applies to the TRANSFORMED input
      }
      halo[s * CSM_NPIX + pp] = v;
      if (p.a_out != nullptr && ok) {
        const int hy = pp / CSM_HX, hx = pp - hy * CSM_HX;
        if (hy >= 1 && hy <= CSM_TH && hx >= 1 && hx <= CSM_TW) {
          const unsigned off = ((unsigned)((b * p.H + h0 - 1 + hy) * p.W + (w0 - 1 + hx))) * CS + 4 * s;
          *reinterpret_cast<f32x4*>(p.a_out + off) = v;
        }
      }
    }
  };

  // Two workgroups share a CU and would run in lockstep (same code, same start): both in the K loop sharing the
  // matrix pipe, then both in staging / epilogue with the pipe idle - measured: MFMA phase + other phases, no
  // overlap.  The second half of the grid (dispatched after every CU has its first workgroup) starts half a tile
  // period late, so one workgroup's staging / epilogue falls into the other's K loop.  Timing only.
  {
    // CU-exact pairing: the hardware id of the CU this workgroup landed on (XCC, SE, SH, CU) indexes a ticket counter;
    // the second workgroup to arrive on a CU (odd ticket) starts half a tile period late.  Tickets are never reset:
    // with two workgroups per CU and launch the parity alternates by itself.
    const int mode = (CSM_DBG(p) >> 3) & 3;  // tuning aid: 0 ticket per CU, 1 nobody, 2 upper half of the grid, 3 odd ids
    int late = 0;
    if (tid == 0) {
      if (mode == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((16 - 1) << 11 | 0 << 6 | 4);   // HW_ID[15:0]: cu_id [11:8], sh_id [12], se_id [15:13]
        const unsigned xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20);  // XCC_ID[3:0]
        const unsigned key = ((xcc & 15u) << 8) | ((hw >> 8) & 255u);
        late = atomicAdd(&g_small_cu_ticket[key], 1) & 1;
      } else {
        late = mode == 2 ? blockIdx.x >= (gridDim.x + 1) / 2 : (mode == 3 ? blockIdx.x & 1 : 0);
      }
      if (late) {
        __builtin_amdgcn_s_sleep(127);
        __builtin_amdgcn_s_sleep(40);
      }
    }
    // the other waves wait for wave 0 at the first barrier of the tile loop
  }
  // statistics accumulated over this workgroup's tiles (threads tid < SQ own one channel quad each)
  f32x4 wg_a = {0.f, 0.f, 0.f, 0.f}, wg_c = wg_a;
  float wg_n = 0.f;
  // acc_rows: every epilogue lane keeps ITS OWN sums over all the tiles it sees (fixed channel quad per lane) and the
  // lanes are folded ONCE, after the last tile - no per-tile barriers / LDS passes for the statistics.
  //   mode 2: la = sum dz, lc = sum dz*xhat;   mode 1: shifted sums around the lane's first value lk (accurate when
  //   |mean| >> std): la = sum (v - lk), lc = sum (v - lk)^2, ln = count.
  f32x4 la = {0.f, 0.f, 0.f, 0.f}, lc = la, lk = la;
  float ln = 0.f;
  int t = blockIdx.x;
  int nb = 0, nh0 = 0, nw0 = 0;
  if (t < p.ntiles) {
    tile_origin(t, nb, nh0, nw0);
    prefetch(nb, nh0, nw0);
  }
  for (; t < p.ntiles; t += gridDim.x) {
    const int b = nb, h0 = nh0, w0 = nw0;
    lds_barrier();  // every wave is done with the previous tile's LDS image (first pass: weights / coef are in LDS)
    if (!(CSM_DBG(p) & 2)) stage_store(b, h0, w0);
    lds_barrier();
    if (t + (int)gridDim.x < p.ntiles && !(CSM_DBG(p) & 2)) {  // global loads stay in flight under the MFMAs
      tile_origin(t + gridDim.x, nb, nh0, nw0);
      prefetch(nb, nh0, nw0);
    }
    // this wave's output row; the mode-2 operand of this tile is fetched under the MFMAs too
    const int h = h0 + wv;
    const bool rowok = h < p.H && !(CSM_DBG(p) & 1);
    const int pixrow = (b * p.H + h) * p.W + w0;  // + pixel of the row
    f32x4 rz[EIT];
    if (p.ep_mode == 2) {
#pragma unroll
      for (int it = 0; it < EIT; ++it) {
        const int px = it * PPI + epl;
        rz[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (elane && px < CSM_TW && rowok && w0 + px < p.W)
          rz[it] = *reinterpret_cast<const f32x4*>(p.ez_x + (unsigned)(pixrow + px) * (unsigned)p.ldy + 4 * eq);
      }
    }

    f32x4 acc[TM][TN];
    f32x2 tacc[TM][NTT];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int tt = 0; tt < NTT; ++tt) tacc[i][tt] = (f32x2){0.f, 0.f};
    }
    // fragments of one k-group: A (TM pixel tiles), B (TN weight-row tiles), tail weight rows
    struct Frag { f32x4 a[TM], b[TN], t[NTT]; };
    auto load_frag = [&](int g, Frag& f) {
      int ai, bi, ti;
      if (g < NGF) {
        const int tap = g / (FG > 0 ? FG : 1), cb = (g % (FG > 0 ? FG : 1)) * 4;
        const int kq = tap * SP + cb;  // + lq
        ai = a_base + cb * CSM_NPIX + (tap / 3) * CSM_HX + tap % 3;
        bi = b_base + kq * NROWS;
        ti = (kq + lq) * 4;
#pragma unroll
        for (int i = 0; i < TM; ++i) f.a[i] = halo[ai + 16 * i];
      } else {
        // left-over quads: lanes without one (a_rem < 0) read the zero quad / the zero k row
        const int h2 = g - NGF;
        const bool dead = a_rem[h2] < 0;
        bi = ks_rem[h2] * NROWS + l15;
        ti = ks_rem[h2] * 4;
#pragma unroll
        for (int i = 0; i < TM; ++i) f.a[i] = halo[dead ? SP * CSM_NPIX : a_rem[h2] + 16 * i];
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) f.b[j] = wm[bi + 16 * j];
      if (NT > 0) {  // tail weight rows (wave-uniform row, one k quad per lane quarter)
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) f.t[tt] = wt[ti + tt];
      }
    };
    auto mma_frag = [&](const Frag& f) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[i][e], f.b[j][e], acc[i][j], 0, 0, 0);
      if (NT > 0) {
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
          const f32x2 lo = __builtin_shufflevector(f.t[tt], f.t[tt], 0, 1), hi = __builtin_shufflevector(f.t[tt], f.t[tt], 2, 3);
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            tacc[i][tt] += __builtin_shufflevector(f.a[i], f.a[i], 0, 1) * lo;
            tacc[i][tt] += __builtin_shufflevector(f.a[i], f.a[i], 2, 3) * hi;
            // pin the accumulation here: left alone, the optimiser sinks the whole chain of tail FMAs below the
            // MFMA loop (their only consumer is the epilogue) and spills every fragment they read
            asm volatile("" : "+v"(tacc[i][tt]));
          }
        }
      }
    };
    // software pipeline: the LDS reads of group g+1 are issued before the MFMAs of group g.  The order inside a group
    // is pinned (all reads of g+1, then the MFMAs of g, then the tail FMAs): left to itself the scheduler put the B
    // reads at the END of the group and the tail read + its FMAs at the start, i.e. two exposed LDS latencies per
    // 16 MFMAs (matrix pipe measured 69 % busy at two waves per SIMD); the barrier at the end keeps it from hoisting ALL
    // groups' reads to the top (which spilled: 21 groups x 12 fragments).
    Frag fr[2];
    load_frag(0, fr[0]);
    if (!(CSM_DBG(p) & 4))
#pragma unroll
    for (int g = 0; g < NGF + NGR; ++g) {
      if (g + 1 < NGF + NGR) load_frag(g + 1, fr[(g + 1) & 1]);
      mma_frag(fr[g & 1]);
      if (g + 1 < NGF + NGR) __builtin_amdgcn_sched_group_barrier(0x100, TM + TN + NT, 0);  // DS reads of g+1
      __builtin_amdgcn_sched_group_barrier(0x008, 4 * TM * TN, 0);                         // MFMAs of g
      __builtin_amdgcn_sched_group_barrier(0x002, 4 * TM * NT + 8, 0);                     // tail FMAs (+ address VALU)
      __builtin_amdgcn_sched_barrier(0);
    }

    // ---------------- epilogue ----------------
    // C layout of 16x16x4: column = lane & 15, row = 4 * (lane >> 4) + reg.  Wave wv owns output row h0 + wv,
    // m tile i covers its pixels 16 i .. 16 i + 15.  Tail columns: fold the four k quarters first.
    float tv[TM][NTT];
    if (NT > 0) {
#pragma unroll
      for (int tt = 0; tt < NT; ++tt)
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          float v = tacc[i][tt][0] + tacc[i][tt][1];
          v += __shfl_xor(v, 16, 64);
          v += __shfl_xor(v, 32, 64);
          tv[i][tt] = v;  // pixel 16 i + l15, column NROWS + tt (identical in the four lane quarters)
        }
    }

    if (p.yb != nullptr) {
      // NCHW split store straight from the C layout: a lane holds 4 consecutive pixels of one channel plane
      const int HWsz = p.H * p.W;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = 16 * j + l15;
        if (n < p.Cout) {
          const float bv = p.bias != nullptr ? p.bias[n] : 0.f;
          float* base = n < p.Ca ? p.y + (unsigned)((b * p.Ca + n) * HWsz) : p.yb + (unsigned)((b * (p.Cout - p.Ca) + (n - p.Ca)) * HWsz);
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            const int wq = w0 + 16 * i + 4 * lq;
            if (rowok && wq < p.W) *reinterpret_cast<f32x4*>(base + (unsigned)(h * p.W + wq)) = acc[i][j] + bv;
          }
        }
      }
      if (NT > 0) {
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
          const int n = NROWS + tt;
          if (n < p.Cout && lq == 0) {
            const float bv = p.bias != nullptr ? p.bias[n] : 0.f;
            float* base = n < p.Ca ? p.y + (unsigned)((b * p.Ca + n) * HWsz) : p.yb + (unsigned)((b * (p.Cout - p.Ca) + (n - p.Ca)) * HWsz);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
              const int w = w0 + 16 * i + l15;
              if (rowok && w < p.W) base[(unsigned)(h * p.W + w)] = tv[i][tt] + bv;
            }
          }
        }
      }
      continue;
    }

    // NHWC: C layout -> LDS output tile (the halo region; every wave must be out of the K loop first) -> each
    // lane moves float4 (pixel, channel quad) rows of the wave's own 32 pixels: fully coalesced global accesses
    lds_barrier();
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) otile[(32 * wv + 16 * i + 4 * lq + r) * OS + 16 * j + l15] = acc[i][j][r];
    if (lq == 0) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) otile[(32 * wv + 16 * i + l15) * OS + NROWS + tt] = tt < NT ? tv[i][tt < NT ? tt : 0] : 0.f;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-local hand-over: a wave only re-reads its own 32 rows

    f32x4 val[EIT];
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = s1;
#pragma unroll
    for (int it = 0; it < EIT; ++it) {
      const int px = it * PPI + epl;
      val[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (elane && px < CSM_TW) {
        f32x4 v = *reinterpret_cast<const f32x4*>(otile + (32 * wv + px) * OS + 4 * eq);
        const bool ok = rowok && w0 + px < p.W;
        if (p.ep_mode == 2) {
          const f32x4 xh = (rz[it] - epar[9 + eq]) * epar[18 + eq];
          const f32x4 z = epar[27 + eq] * xh + epar[36 + eq];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] *= act_grad(z[e], p.ez_act);
          if (!ok) v = (f32x4){0.f, 0.f, 0.f, 0.f};
          s1 += v;
          s2 += v * xh;
        } else {
          v += epar[eq];
          if (ok) s1 += v;
          if (p.ep_mode == 1 && p.acc_rows && ok) {
            if (ln == 0.f) lk = v;
            const f32x4 d = v - lk;
            la += d;
            lc += d * d;
            ln += 1.f;
          }
        }
        val[it] = v;
        if (ok) *reinterpret_cast<f32x4*>(p.y + (unsigned)(pixrow + px) * (unsigned)p.ldy + 4 * eq) = v;
      }
    }

    if (p.ep_mode == 2 && p.acc_rows) {
      la += s1;
      lc += s2;
    } else if (p.ep_mode != 0 && !p.acc_rows) {
      // per-tile column sums: lane partials -> LDS -> the first SQ threads add the 4 x PPI partials of their quad
      red[wv * 64 + lane] = s1;
      red[256 + wv * 64 + lane] = s2;
      lds_barrier();
      if (p.ep_mode == 1) {
        // (mean, M2): the tile mean first (statistics are only enabled for full tiles: 128 pixels), M2 around it
        constexpr float inv_n = 1.f / (float)(CSM_TH * CSM_TW);
        f32x4 m = {0.f, 0.f, 0.f, 0.f};
        if (elane) {
#pragma unroll
          for (int w = 0; w < 4; ++w)
#pragma unroll
            for (int q = 0; q < PPI; ++q) m += red[w * 64 + q * SQ + eq];
        }
        m *= inv_n;
        f32x4 q2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int it = 0; it < EIT; ++it)
          if (elane && it * PPI + epl < CSM_TW) q2 += (val[it] - m) * (val[it] - m);
        red[256 + wv * 64 + lane] = q2;
        lds_barrier();
        if (tid < SQ && 4 * tid < p.ldy) {
          f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int w = 0; w < 4; ++w)
#pragma unroll
            for (int q = 0; q < PPI; ++q) c += red[256 + w * 64 + q * SQ + tid];
          *reinterpret_cast<f32x4*>(p.stats + ((size_t)t * 2 + 0) * p.ldy + 4 * tid) = m;  // tid == eq for these lanes
          *reinterpret_cast<f32x4*>(p.stats + ((size_t)t * 2 + 1) * p.ldy + 4 * tid) = c;
        }
      } else if (tid < SQ && 4 * tid < p.ldy) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f}, c = a;
#pragma unroll
        for (int w = 0; w < 4; ++w)
#pragma unroll
          for (int q = 0; q < PPI; ++q) {
            a += red[w * 64 + q * SQ + tid];
            c += red[256 + w * 64 + q * SQ + tid];
          }
        *reinterpret_cast<f32x4*>(p.stats + ((size_t)t * 2 + 0) * p.ldy + 4 * tid) = a;
        *reinterpret_cast<f32x4*>(p.stats + ((size_t)t * 2 + 1) * p.ldy + 4 * tid) = c;
      }
    }
  }
  if (p.ep_mode != 0 && p.acc_rows) {
    // fold the lanes' sums of this workgroup into its statistics row (once per kernel)
    __syncthreads();  // the last tile's output image / halo is dead
    f32x4* fold = halo;  // [3][256]
    if (p.ep_mode == 1) {  // lane (count, mean, M2) from its shifted sums
      f32x4 mean = lk, m2 = {0.f, 0.f, 0.f, 0.f};
      if (ln > 0.f) {
        mean = lk + la * (1.f / ln);
        m2 = lc - la * la * (1.f / ln);
      }
      fold[tid] = mean;
      fold[256 + tid] = m2;
      fold[512 + tid] = (f32x4){ln, 0.f, 0.f, 0.f};
    } else {
      fold[tid] = la;
      fold[256 + tid] = lc;
    }
    __syncthreads();
    if (tid < SQ && 4 * tid < p.ldy) {
      if (p.ep_mode == 1) {  // Chan's merge over the 4 x PPI lanes of this channel quad
#pragma unroll
        for (int w = 0; w < 4; ++w)
#pragma unroll
          for (int q = 0; q < PPI; ++q) {
            const int j = w * 64 + q * SQ + tid;
            const float nb = fold[512 + j][0];
            if (nb > 0.f) {
              const float nt = wg_n + nb;
              const f32x4 d = fold[j] - wg_a;
              wg_a += d * (nb / nt);
              wg_c += fold[256 + j] + d * d * (wg_n * nb / nt);
              wg_n = nt;
            }
          }
      } else {
#pragma unroll
        for (int w = 0; w < 4; ++w)
#pragma unroll
          for (int q = 0; q < PPI; ++q) {
            wg_a += fold[w * 64 + q * SQ + tid];
            wg_c += fold[256 + w * 64 + q * SQ + tid];
          }
      }
      *reinterpret_cast<f32x4*>(p.stats + ((size_t)blockIdx.x * 2 + 0) * p.ldy + 4 * tid) = wg_a;
      *reinterpret_cast<f32x4*>(p.stats + ((size_t)blockIdx.x * 2 + 1) * p.ldy + 4 * tid) = wg_c;
    }
  }
}

// ---------------------------------------------------------------------------------------------- launch
int small_cus();
int small_grid(int ntiles, int* acc_rows);

template <int CS, int TN, int NT, bool X2>
static int launch_small_x(SmallP& p, hipStream_t st) {
  using C = SmallCfg<CS, TN, NT>;
  // set on every launch: a function attribute is per device, and a cached flag would be neither thread-safe nor
  // right for a second GPU of the process (the call is a cheap driver-side table update)
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_small_kernel<CS, TN, NT, X2>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES) != hipSuccess)
    return VMTL_ERR_LAUNCH;
  static_assert(C::LDS_BYTES * 2 <= 160 * 1024, "two workgroups per CU (small_grid() assumes it)");
  int grid = p.grid;
  if (CSM_DBG(p) & 64) grid = small_cus() < p.ntiles ? small_cus() : p.ntiles;  // tuning aid: one workgroup per CU
  hipLaunchKernelGGL((conv3x3_small_kernel<CS, TN, NT, X2>), dim3(grid), dim3(256), C::LDS_BYTES, st, p);
  return vmtl_check_launch();
}

// weight rows -> (MFMA column tiles, VALU tail columns): 1..16 -> (1,0), 17 -> (1,1), 18..20 -> (1,4),
// 21..32 -> (2,0), 33 -> (2,1), 34..36 -> (2,4)
template <int CS>
static int launch_small_cs(SmallP& p, hipStream_t st) {
  const bool x2 = p.x2 != nullptr;
#define VMTL_SMALL_CASE(TN, NT) return x2 ? launch_small_x<CS, TN, NT, true>(p, st) : launch_small_x<CS, TN, NT, false>(p, st)
  if (p.Nw <= 16) VMTL_SMALL_CASE(1, 0);
  if (p.Nw == 17) VMTL_SMALL_CASE(1, 1);
  if (p.Nw <= 20) VMTL_SMALL_CASE(1, 4);
  if (p.Nw <= 32) VMTL_SMALL_CASE(2, 0);
  if (p.Nw == 33) VMTL_SMALL_CASE(2, 1);
  VMTL_SMALL_CASE(2, 4);
#undef VMTL_SMALL_CASE
}
