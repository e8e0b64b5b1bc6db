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
// LDS layouts are "slot-major": halo[channel quad][pixel][4] and w[k quad][row][4] with the pixel / row extents
// multiples of 16, so the four 16-lane groups of a ds_read_b128 (MI355X_MICROARCH.md, LDS table) each touch 16
// distinct 16-byte slots: conflict-free without a swizzle.
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
#include "common.h"

#define CSM_TH 4
#define CSM_TW 32
#define CSM_HX (CSM_TW + 2)
#define CSM_NHALO ((CSM_TH + 2) * CSM_HX)  // 204 halo pixels
#define CSM_NPIX 208                       // rounded up to a multiple of 16 (slot-major stride)

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
};

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
  static constexpr int NCAP = NROWS + NT;   // + tail rows dotted on the VALU
  static constexpr int NST = CSM_NHALO * SP;        // float4 elements of one halo
  static constexpr int IT = (NST + 255) / 256;      // staging iterations per thread
  // LDS image in float4 units
  static constexpr int HALO4 = SP * CSM_NPIX + 1;   // + one zero quad for the lanes of a partial k-group
  static constexpr int WM4 = (KS + 1) * NROWS;      // + one zero k quad
  static constexpr int WT4 = (KS + 1) * 4;
  static constexpr int COEF4 = 3 * SP;
  static constexpr int RED4 = (2 * 4 * NCAP + 3) / 4;
  static constexpr int LDS_BYTES = (HALO4 + WM4 + WT4 + COEF4 + RED4) * 16;
};

template <int CS, int TN, int NT, bool X2>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv3x3_small_kernel(SmallP p) {
  using C = SmallCfg<CS, TN, NT>;
  constexpr int SP = C::SP, FG = C::FG, RS = C::RS, KS = C::KS, NGF = C::NGF, NR = C::NR, NGR = C::NGR;
  constexpr int NROWS = C::NROWS, NCAP = C::NCAP, NST = C::NST, IT = C::IT;
  constexpr int TM = 2;
  constexpr int NTT = NT > 0 ? NT : 1;

  extern __shared__ __attribute__((aligned(16))) f32x4 smem4[];
  f32x4* halo = smem4;                  // [SP][NPIX] (+ zero quad at SP*NPIX)
  f32x4* wm = halo + C::HALO4;          // [KS+1][NROWS]
  f32x4* wt = wm + C::WM4;              // [KS+1][4]
  f32x4* coef = wt + C::WT4;            // [3][SP]
  float* red = reinterpret_cast<float*>(coef + C::COEF4);  // [2][4][NCAP]

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
  if (tid == 0) halo[SP * CSM_NPIX] = (f32x4){0.f, 0.f, 0.f, 0.f};
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
  auto prefetch = [&](int t) {
    int b, h0, w0;
    tile_origin(t, b, h0, w0);
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
  auto stage_store = [&](int t) {
    int b, h0, w0;
    tile_origin(t, b, h0, w0);
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

  int t = blockIdx.x;
  if (t < p.ntiles) prefetch(t);
  for (; t < p.ntiles; t += gridDim.x) {
    __syncthreads();  // every wave is done with the previous tile's halo (first pass: weights / coef are in LDS)
    stage_store(t);
    __syncthreads();
    if (t + (int)gridDim.x < p.ntiles) prefetch(t + gridDim.x);  // global loads stay in flight under the MFMAs

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
        const int h = g - NGF;
        const bool dead = a_rem[h] < 0;
        bi = ks_rem[h] * NROWS + l15;
        ti = ks_rem[h] * 4;
#pragma unroll
        for (int i = 0; i < TM; ++i) f.a[i] = halo[dead ? SP * CSM_NPIX : a_rem[h] + 16 * i];
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) f.b[j] = wm[bi + 16 * j];
      if (NT > 0) {
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
    // software pipeline: the LDS reads of group g+1 are issued before the MFMAs of group g; the scheduling barrier
    // keeps the compiler from hoisting ALL groups' reads to the top (which spilled: 21 groups x 12 fragments)
    Frag fr[2];
    load_frag(0, fr[0]);
#pragma unroll
    for (int g = 0; g < NGF + NGR; ++g) {
      if (g + 1 < NGF + NGR) load_frag(g + 1, fr[(g + 1) & 1]);
      mma_frag(fr[g & 1]);
      __builtin_amdgcn_sched_barrier(0);
    }

    // ---------------- epilogue ----------------
    // C layout of 16x16x4: column = lane & 15, row = 4 * (lane >> 4) + reg.  Wave wv owns output row h0 + wv,
    // m tile i covers columns w0 + 16 i .. + 15.
    int b, h0, w0;
    tile_origin(t, b, h0, w0);
    const int h = h0 + wv;
    const bool rowok = h < p.H;
    const unsigned pix0 = (unsigned)((b * p.H + h) * p.W);  // + w (32-bit element offsets: host checks the extents)
    const int HWsz = p.H * p.W;
    float s1[TN], s2[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) s1[j] = s2[j] = 0.f;

#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = 16 * j + l15;
      const bool nok = n < p.Cout;
      const float bv = (p.bias != nullptr && n < p.Nw) ? p.bias[n] : 0.f;
      float em = 0.f, ei = 0.f, eg = 0.f, eb = 0.f;
      if (p.ep_mode == 2 && nok) {
        em = p.ez_mean[n]; ei = p.ez_invstd[n]; eg = p.ez_gamma[n]; eb = p.ez_beta[n];
      }
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        f32x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int w = w0 + 16 * i + 4 * lq + r;
          const bool ok = rowok && w < p.W;
          float val = nok ? acc[i][j][r] + bv : 0.f;
          if (p.ep_mode == 2) {
            float xh = 0.f;
            if (ok && nok && n < p.ldy) {
              xh = (p.ez_x[(pix0 + w) * p.ldy + n] - em) * ei;
              val *= act_grad(eg * xh + eb, p.ez_act);
            } else {
              val = 0.f;
            }
            s1[j] += val;
            s2[j] += val * xh;
          } else if (p.ep_mode == 1) {
            if (ok) s1[j] += val;
          }
          v[r] = val;
          acc[i][j][r] = val;  // kept for the second statistics pass
        }
        const int wq = w0 + 16 * i + 4 * lq;
        if (p.yb != nullptr) {  // NCHW split store: 4 consecutive pixels of one channel plane
          if (rowok && wq < p.W && nok) {
            float* base = n < p.Ca ? p.y + (unsigned)((b * p.Ca + n) * HWsz) : p.yb + (unsigned)((b * (p.Cout - p.Ca) + (n - p.Ca)) * HWsz);
            *reinterpret_cast<f32x4*>(base + (unsigned)(h * p.W + wq)) = v;
          }
        } else if (n < p.ldy) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (rowok && wq + r < p.W) p.y[(pix0 + wq + r) * p.ldy + n] = v[r];
        }
      }
    }

    // tail columns: fold the four k quarters; every lane then holds the value of pixel (16 i + l15), column NROWS + tt
    float tv[TM][NTT], ts1[NTT], ts2[NTT];
    if (NT > 0) {
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) {
        ts1[tt] = ts2[tt] = 0.f;
        const int n = NROWS + tt;
        const bool nok = n < p.Cout;
        const float bv = (p.bias != nullptr && n < p.Nw) ? p.bias[n] : 0.f;
        float em = 0.f, ei = 0.f, eg = 0.f, eb = 0.f;
        if (p.ep_mode == 2 && nok) {
          em = p.ez_mean[n]; ei = p.ez_invstd[n]; eg = p.ez_gamma[n]; eb = p.ez_beta[n];
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          float v = tacc[i][tt][0] + tacc[i][tt][1];
          v += __shfl_xor(v, 16, 64);
          v += __shfl_xor(v, 32, 64);
          const int w = w0 + 16 * i + l15;
          const bool ok = rowok && w < p.W;
          v = nok ? v + bv : 0.f;
          if (p.ep_mode == 2) {
            float xh = 0.f;
            if (ok && nok && n < p.ldy) {
              xh = (p.ez_x[(pix0 + w) * p.ldy + n] - em) * ei;
              v *= act_grad(eg * xh + eb, p.ez_act);
            } else {
              v = 0.f;
            }
            ts1[tt] += v;
            ts2[tt] += v * xh;
          } else if (p.ep_mode == 1) {
            if (ok) ts1[tt] += v;
          }
          tv[i][tt] = v;
          if (lq == 0 && ok) {
            if (p.yb != nullptr) {
              if (nok) {
                float* base = n < p.Ca ? p.y + (unsigned)((b * p.Ca + n) * HWsz) : p.yb + (unsigned)((b * (p.Cout - p.Ca) + (n - p.Ca)) * HWsz);
                base[(unsigned)(h * p.W + w)] = v;
              }
            } else if (n < p.ldy) {
              p.y[(pix0 + w) * p.ldy + n] = v;
            }
          }
        }
      }
    }

    if (p.ep_mode != 0) {
      // column sums of this wave's 32 pixels -> LDS -> per-tile rows.  Mode 1 needs a second pass for M2.
      float* red1 = red;               // [4][NCAP]
      float* red2 = red + 4 * NCAP;    // [4][NCAP]
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        float a = s1[j], c = s2[j];
        a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
        c += __shfl_xor(c, 16, 64); c += __shfl_xor(c, 32, 64);
        if (lq == 0) {
          red1[wv * NCAP + 16 * j + l15] = a;
          red2[wv * NCAP + 16 * j + l15] = c;
        }
      }
      if (NT > 0) {
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
          float a = ts1[tt], c = ts2[tt];  // identical in the four lane quarters: fold the 16 pixels of a quarter
#pragma unroll
          for (int o = 1; o < 16; o <<= 1) {
            a += __shfl_xor(a, o, 64);
            c += __shfl_xor(c, o, 64);
          }
          if (lane == 0) {
            red1[wv * NCAP + NROWS + tt] = a;
            red2[wv * NCAP + NROWS + tt] = c;
          }
        }
      }
      __syncthreads();
      if (p.ep_mode == 2) {
        if (tid < NCAP && tid < p.ldy) {
          float a = 0.f, c = 0.f;
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            a += red1[w * NCAP + tid];
            c += red2[w * NCAP + tid];
          }
          p.stats[((size_t)t * 2 + 0) * p.ldy + tid] = a;
          p.stats[((size_t)t * 2 + 1) * p.ldy + tid] = c;
        }
      } else {
        // mean of the tile (the host only enables statistics for full tiles: 128 pixels), then M2 around it
        constexpr float inv_n = 1.f / (float)(CSM_TH * CSM_TW);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          float m = 0.f;
#pragma unroll
          for (int w = 0; w < 4; ++w) m += red1[w * NCAP + 16 * j + l15];
          m *= inv_n;
          float q = 0.f;
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) q += (acc[i][j][r] - m) * (acc[i][j][r] - m);
          q += __shfl_xor(q, 16, 64);
          q += __shfl_xor(q, 32, 64);
          if (lq == 0) red2[wv * NCAP + 16 * j + l15] = q;
        }
        if (NT > 0) {
#pragma unroll
          for (int tt = 0; tt < NT; ++tt) {
            float m = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) m += red1[w * NCAP + NROWS + tt];
            m *= inv_n;
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i) q += (tv[i][tt] - m) * (tv[i][tt] - m);
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) q += __shfl_xor(q, o, 64);
            if (lane == 0) red2[wv * NCAP + NROWS + tt] = q;
          }
        }
        __syncthreads();
        if (tid < NCAP && tid < p.ldy) {
          float a = 0.f, c = 0.f;
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            a += red1[w * NCAP + tid];
            c += red2[w * NCAP + tid];
          }
          p.stats[((size_t)t * 2 + 0) * p.ldy + tid] = a * inv_n;
          p.stats[((size_t)t * 2 + 1) * p.ldy + tid] = c;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------- host side
static int small_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        v <= 0)
      v = 256;
    n = v;
  }
  return n;
}

template <int CS, int TN, int NT, bool X2>
static int launch_small_x(SmallP& p, hipStream_t st) {
  using C = SmallCfg<CS, TN, NT>;
  // set on every launch: a function attribute is per device, and a cached flag would be neither thread-safe nor
  // right for a second GPU of the process (the call is a cheap driver-side table update)
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_small_kernel<CS, TN, NT, X2>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES) != hipSuccess)
    return VMTL_ERR_LAUNCH;
  const int per_cu = C::LDS_BYTES * 2 <= 160 * 1024 ? 2 : 1;
  int grid = small_cus() * per_cu;
  if (grid > p.ntiles) grid = p.ntiles;
  hipLaunchKernelGGL((conv3x3_small_kernel<CS, TN, NT, X2>), dim3(grid), dim3(256), C::LDS_BYTES, st, p);
  return vmtl_check_launch();
}

template <int CS, int TN, int NT>
static int launch_small(SmallP& p, hipStream_t st) {
  return p.x2 != nullptr ? launch_small_x<CS, TN, NT, true>(p, st) : launch_small_x<CS, TN, NT, false>(p, st);
}

// 1 when vmtl_conv3x3_small handles a 3x3/s1/p1 conv with Cs input storage channels and Nw weight rows
extern "C" int vmtl_conv3x3_small_supported(int Cs, int Nw) {
  return (Cs == 20 || Cs == 36 || Cs == 32 || Cs == 16) && Nw >= 1 && Nw <= 36;
}

// statistics rows (= tiles); the statistics epilogues need full tiles: H % 4 == 0 and W % 32 == 0
extern "C" int vmtl_conv3x3_small_tiles(int B, int H, int W) { return B * cdiv(H, CSM_TH) * cdiv(W, CSM_TW); }

extern "C" int vmtl_conv3x3_small(const float* x, const float* x2, const float* pa, const float* pb, const float* pc,
                                  int act_in, float* a_out, const float* wp, const float* bias, float* y, float* yb,
                                  int Ca, float* stats, int ep_mode, const float* ez_x, const float* ez_mean,
                                  const float* ez_invstd, const float* ez_gamma, const float* ez_beta, int ez_act, int B,
                                  int H, int W, int Cs, int ldy, int Nw, int Cout, void* stream) {
  VMTL_ENTER();
  if (!x || !wp || !y || B <= 0 || H <= 0 || W <= 0 || ldy <= 0 || (ldy & 3)) return VMTL_ERR_ARG;
  if (!vmtl_conv3x3_small_supported(Cs, Nw) || Cout <= 0 || Cout > Nw || Nw > ldy || ldy > 36) return VMTL_ERR_ARG;
  if ((pa == nullptr) != (pc == nullptr) || (x2 != nullptr && (pa == nullptr || pb == nullptr))) return VMTL_ERR_ARG;
  if (a_out != nullptr && pa == nullptr) return VMTL_ERR_ARG;
  if (act_in != VMTL_ACT_NONE && act_in != VMTL_ACT_RELU) return VMTL_ERR_ARG;
  if (ep_mode < 0 || ep_mode > 2 || (ep_mode != 0 && (!stats || (H % CSM_TH) || (W % CSM_TW)))) return VMTL_ERR_ARG;
  if (ep_mode == 2 && (!ez_x || !ez_mean || !ez_invstd || !ez_gamma || !ez_beta)) return VMTL_ERR_ARG;
  if (ep_mode != 0 && (bias != nullptr || yb != nullptr)) return VMTL_ERR_ARG;
  if (yb != nullptr && (Ca <= 0 || Ca >= Cout || (W & 3))) return VMTL_ERR_ARG;
  if ((long long)B * H * W * 36 > 0x7fffffffLL) return VMTL_ERR_UNSUPPORTED;  // 32-bit element offsets in the kernel
  SmallP p;
  p.x = x; p.x2 = x2; p.pa = pa; p.pb = pb; p.pc = pc; p.a_out = a_out; p.wp = wp; p.bias = bias; p.y = y; p.yb = yb;
  p.stats = stats; p.ez_x = ez_x; p.ez_mean = ez_mean; p.ez_invstd = ez_invstd; p.ez_gamma = ez_gamma; p.ez_beta = ez_beta;
  p.act_in = act_in; p.ep_mode = ep_mode; p.ez_act = ez_act; p.Ca = Ca;
  p.B = B; p.H = H; p.W = W; p.ldy = ldy; p.Nw = Nw; p.Cout = Cout;
  p.tiles_x = cdiv(W, CSM_TW); p.tiles_y = cdiv(H, CSM_TH); p.ntiles = B * p.tiles_x * p.tiles_y;
  hipStream_t st = (hipStream_t)stream;
  const bool wide = Nw > 20;  // 32 MFMA columns + 4 on the VALU, else 16 + 4
  switch (Cs) {
    case 36: return wide ? launch_small<36, 2, 4>(p, st) : launch_small<36, 1, 4>(p, st);
    case 20: return wide ? launch_small<20, 2, 4>(p, st) : launch_small<20, 1, 4>(p, st);
    case 32: return wide ? launch_small<32, 2, 4>(p, st) : launch_small<32, 1, 4>(p, st);
    default: return wide ? launch_small<16, 2, 4>(p, st) : launch_small<16, 1, 4>(p, st);
  }
}
