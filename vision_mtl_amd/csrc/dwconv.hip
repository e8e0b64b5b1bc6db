// Depthwise KxK convolution (k in {3,5}, stride in {1,2}), NHWC fp32: forward, data
// gradient and weight gradient.  K = 9 / 25 per output, so this is VALU + L1/L2-reuse
// work (not MFMA): one thread owns 4 channels of one pixel and walks the taps; all
// accesses are 16-byte and coalesced along the channel axis.
//
// Replaces the depthwise ATen convs of timm MobileNetV3 DepthwiseSeparable / InvertedResidual
// blocks reached from reference vision_mtl/utils/model_utils.py:25-34 (smp.Unet encoder) [3P].
//
// Weights are consumed in packed [tap][Cs] form (see pack.hip).
#include <stdlib.h>

#include "reduce.h"

#define DW_MAX_TAPS 25

__global__ __launch_bounds__(256) void dwconv_fwd_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                         float* __restrict__ y, int B, int H, int W, int Cs, int Ho,
                                                         int Wo, int K, int stride, int pad, long long total4) {
  const int CQ = Cs >> 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total4;
       i += (long long)gridDim.x * blockDim.x) {
    const int q = (int)(i % CQ);
    const long long pix = i / CQ;
    const int wo = (int)(pix % Wo), ho = (int)((pix / Wo) % Ho), b = (int)(pix / ((long long)Wo * Ho));
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int dh = 0; dh < K; ++dh) {
      const int h = ho * stride - pad + dh;
      if ((unsigned)h >= (unsigned)H) continue;
      for (int dw = 0; dw < K; ++dw) {
        const int w = wo * stride - pad + dw;
        if ((unsigned)w >= (unsigned)W) continue;
        const f32x4 xv = *reinterpret_cast<const f32x4*>(x + ((size_t)(b * H + h) * W + w) * Cs + (size_t)q * 4);
        const f32x4 wv = *reinterpret_cast<const f32x4*>(wp + (size_t)(dh * K + dw) * Cs + (size_t)q * 4);
        acc += xv * wv;
      }
    }
    reinterpret_cast<f32x4*>(y)[i] = acc;
  }
}

// Depthwise conv as a PRE-ACTIVATION node: the input is the RAW output of the preceding pointwise conv, its
// BatchNorm + activation (timm InvertedResidual: conv_pw -> bn1 -> act -> conv_dw -> bn2, reached from reference
// utils/model_utils.py:25-34) is applied while the taps are loaded (v = act(ca[c] * x + cc[c]); zero padding applies
// to v), and the BatchNorm (mean, M2) partial rows of the OUTPUT come from the same pass - the separate apply and
// statistics launches (2 of the ~14 launches of a block's forward) disappear.  a_out (optional) receives the
// activated input for the weight-gradient kernel: input pixel (h, w) is written by the one tap that owns it
// (dh = pad + h % stride, dw = pad + w % stride, i.e. the centre tap for stride 1).
// Threads keep ONE channel quad and sweep output pixels (reduce.h mapping); one partial row per workgroup.
template <int ACT>
__global__ __launch_bounds__(RED_THREADS) void dwconv_bn_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ ca, const float* __restrict__ cc, const float* __restrict__ wp,
    float* __restrict__ y, float* __restrict__ a_out, float* __restrict__ partial, int H, int W, int Cs, int Ho, int Wo, int K,
    int stride, int pad, int M, int rows_per_blk) {
  __shared__ f32x4 red_mean[RED_THREADS];
  __shared__ f32x4 red_m2[RED_THREADS];
  __shared__ float red_n[RED_THREADS];
  const int CQ = Cs >> 2;
  const int r_begin = blockIdx.x * rows_per_blk;
  const int r_end = min(M, r_begin + rows_per_blk);
  for (int q0 = 0; q0 < CQ; q0 += RED_THREADS) {
    const int cq = min(CQ - q0, RED_THREADS);
    const int rpt = RED_THREADS / cq;
    const int T = rpt * cq;
    const int t = threadIdx.x;
    const int q = q0 + t % cq, ro = t / cq;
    f32x4 K0 = {0.f, 0.f, 0.f, 0.f}, s1 = K0, s2 = K0;
    float n = 0.f;
    if (t < T) {
      const f32x4 sc = *reinterpret_cast<const f32x4*>(ca + (size_t)q * 4);
      const f32x4 sh = *reinterpret_cast<const f32x4*>(cc + (size_t)q * 4);
      for (int r = r_begin + ro; r < r_end; r += rpt) {
        const int wo = r % Wo, ho = (r / Wo) % Ho, b = r / (Wo * Ho);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int dh = 0; dh < K; ++dh) {
          const int h = ho * stride - pad + dh;
          if ((unsigned)h >= (unsigned)H) continue;
          for (int dw = 0; dw < K; ++dw) {
            const int w = wo * stride - pad + dw;
            if ((unsigned)w >= (unsigned)W) continue;
            const size_t off = ((size_t)(b * H + h) * W + w) * Cs + (size_t)q * 4;
            f32x4 v = *reinterpret_cast<const f32x4*>(x + off) * sc + sh;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = act_fwd(v[e], ACT);
            acc += v * *reinterpret_cast<const f32x4*>(wp + (size_t)(dh * K + dw) * Cs + (size_t)q * 4);
            if (a_out != nullptr && dh == pad + h % stride && dw == pad + w % stride)
              *reinterpret_cast<f32x4*>(a_out + off) = v;
          }
        }
        *reinterpret_cast<f32x4*>(y + (size_t)r * Cs + (size_t)q * 4) = acc;
        if (n == 0.f) K0 = acc;  // shifted sums around the first value: accurate when |mean| >> std
        const f32x4 d = acc - K0;
        s1 += d;
        s2 += d * d;
        n += 1.f;
      }
    }
    if (partial == nullptr) continue;
    f32x4 mean = K0, m2 = {0.f, 0.f, 0.f, 0.f};
    if (n > 0.f) {
      mean = K0 + s1 * (1.f / n);
      m2 = s2 - s1 * s1 * (1.f / n);
    }
    __syncthreads();
    red_mean[t] = mean;
    red_m2[t] = m2;
    red_n[t] = t < T ? n : 0.f;
    __syncthreads();
    if (t < cq) {
      float nt = red_n[t];
      f32x4 mt = red_mean[t], m2t = red_m2[t];
      for (int j = 1; j < rpt; ++j) chan_merge(nt, mt, m2t, red_n[t + j * cq], red_mean[t + j * cq], red_m2[t + j * cq]);
      *reinterpret_cast<f32x4*>(partial + ((size_t)blockIdx.x * 2 + 0) * Cs + (size_t)(q0 + t) * 4) = mt;
      *reinterpret_cast<f32x4*>(partial + ((size_t)blockIdx.x * 2 + 1) * Cs + (size_t)(q0 + t) * 4) = m2t;
    }
  }
}

// The same node for the shapes the encoder has (K in {3, 5}, stride in {1, 2}, even Wo), with everything the generic
// kernel leaves to run time fixed at compile time: the K*K taps are unrolled, so the K+S loads of a tap row are in
// flight together (the run-time tap loops issued one dependent L2 round trip per tap: 27 us for a 16x32x120 5x5
// layer whose traffic is worth 8); the thread's K*K weight quads live in registers across its pixels; and a thread
// produces TWO horizontally adjacent outputs per pass, which share K-S of their K columns - (K+S)/(2K) of the loads
// and of the BatchNorm + activation evaluations (5x5 stride 1: 6 instead of 10 per tap row).
template <int ACT, int K, int S, bool WLDS>
__global__ __launch_bounds__(RED_THREADS) void dwconv_bn_fwd_pair_kernel(
    const float* __restrict__ x, const float* __restrict__ ca, const float* __restrict__ cc, const float* __restrict__ wp,
    float* __restrict__ y, float* __restrict__ a_out, float* __restrict__ partial, int H, int W, int Cs, int Ho, int Wo,
    int M, int rows_per_blk) {
  constexpr int PAD = (K - 1) / 2, NC = K + S;
  // weights: 3x3 - the thread's 9 quads in registers; 5x5 (25 quads would cost 100 VGPRs) - the whole [K*K][Cs]
  // matrix in LDS when a workgroup has enough pixels to pay for the copy (WLDS), else read through L1 as needed
  constexpr bool WREG = K == 3;
  extern __shared__ __attribute__((aligned(16))) float wsm[];  // WLDS: [K*K][Cs]
  if (WLDS) {
    for (int i = threadIdx.x; i < K * K * (Cs >> 2); i += RED_THREADS)
      reinterpret_cast<f32x4*>(wsm)[i] = reinterpret_cast<const f32x4*>(wp)[i];
    __syncthreads();
  }
  __shared__ f32x4 red_mean[RED_THREADS];
  __shared__ f32x4 red_m2[RED_THREADS];
  __shared__ float red_n[RED_THREADS];
  const int CQ = Cs >> 2;
  const int r_begin = blockIdx.x * rows_per_blk;  // even (host)
  const int r_end = min(M, r_begin + rows_per_blk);
  for (int q0 = 0; q0 < CQ; q0 += RED_THREADS) {
    const int cq = min(CQ - q0, RED_THREADS);
    const int rpt = RED_THREADS / cq;
    const int T = rpt * cq;
    const int t = threadIdx.x;
    const int q = q0 + t % cq, ro = t / cq;
    f32x4 K0 = {0.f, 0.f, 0.f, 0.f}, s1 = K0, s2 = K0;
    float n = 0.f;
    if (t < T) {
      const f32x4 sc = *reinterpret_cast<const f32x4*>(ca + (size_t)q * 4);
      const f32x4 sh = *reinterpret_cast<const f32x4*>(cc + (size_t)q * 4);
      f32x4 wgt[WREG ? K * K : 1];
      if (WREG) {
#pragma unroll
        for (int i = 0; i < K * K; ++i) wgt[i] = *reinterpret_cast<const f32x4*>(wp + (size_t)i * Cs + (size_t)q * 4);
      }
      auto wq = [&](int tap) -> f32x4 {
        if constexpr (WREG) return wgt[tap];
        return *reinterpret_cast<const f32x4*>((WLDS ? wsm : wp) + tap * Cs + q * 4);
      };
      const float* xq = x + (size_t)q * 4;
      for (int r = r_begin + 2 * ro; r < r_end; r += 2 * rpt) {  // outputs r, r + 1: the same row (Wo is even)
        const int wo = r % Wo, ho = (r / Wo) % Ho, b = r / (Wo * Ho);
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
        // 5x5: the tap ROWS stay a loop (unrolled, the compiler hoists all 35 loads of a pixel pair: 256 VGPRs, one
        // wave per SIMD; one row's K+S loads in flight per wave and 4 waves hide the latency better)
#pragma unroll K > 3 ? 1 : K
        for (int dh = 0; dh < K; ++dh) {
          const int h = ho * S - PAD + dh;
          const bool hok = (unsigned)h < (unsigned)H;
          const int rowoff = (b * H + (hok ? h : 0)) * W;
          f32x4 v[NC];
          bool ok[NC];
#pragma unroll
          for (int j = 0; j < NC; ++j) {
            const int w = wo * S - PAD + j;
            ok[j] = hok && (unsigned)w < (unsigned)W;
            v[j] = *reinterpret_cast<const f32x4*>(xq + (size_t)(rowoff + (ok[j] ? w : 0)) * Cs);
          }
#pragma unroll
          for (int j = 0; j < NC; ++j) {
            f32x4 a = v[j] * sc + sh;
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] = ok[j] ? act_fwd(a[e], ACT) : 0.f;  // the conv pads the ACTIVATED map
            if (j < K) acc0 += a * wq(dh * K + j);
            if (j >= S) acc1 += a * wq(dh * K + j - S);
            // input pixel (h, w) belongs to output (h / S, w / S): rows dh - PAD in [0, S), columns j - PAD in [0, 2S)
            if (dh >= PAD && dh < PAD + S && j >= PAD && j < PAD + 2 * S) {
              if (a_out != nullptr && ok[j])
                *reinterpret_cast<f32x4*>(a_out + (size_t)(rowoff + wo * S - PAD + j) * Cs + (size_t)q * 4) = a;
            }
          }
        }
        *reinterpret_cast<f32x4*>(y + (size_t)r * Cs + (size_t)q * 4) = acc0;
        *reinterpret_cast<f32x4*>(y + (size_t)(r + 1) * Cs + (size_t)q * 4) = acc1;
        if (n == 0.f) K0 = acc0;  // shifted sums around the first value: accurate when |mean| >> std
        const f32x4 d0 = acc0 - K0, d1 = acc1 - K0;
        s1 += d0 + d1;
        s2 += d0 * d0 + d1 * d1;
        n += 2.f;
      }
    }
    if (partial == nullptr) continue;
    f32x4 mean = K0, m2 = {0.f, 0.f, 0.f, 0.f};
    if (n > 0.f) {
      mean = K0 + s1 * (1.f / n);
      m2 = s2 - s1 * s1 * (1.f / n);
    }
    __syncthreads();
    red_mean[t] = mean;
    red_m2[t] = m2;
    red_n[t] = t < T ? n : 0.f;
    __syncthreads();
    if (t < cq) {
      float nt = red_n[t];
      f32x4 mt = red_mean[t], m2t = red_m2[t];
      for (int j = 1; j < rpt; ++j) chan_merge(nt, mt, m2t, red_n[t + j * cq], red_mean[t + j * cq], red_m2[t + j * cq]);
      *reinterpret_cast<f32x4*>(partial + ((size_t)blockIdx.x * 2 + 0) * Cs + (size_t)(q0 + t) * 4) = mt;
      *reinterpret_cast<f32x4*>(partial + ((size_t)blockIdx.x * 2 + 1) * Cs + (size_t)(q0 + t) * 4) = m2t;
    }
  }
}

__global__ __launch_bounds__(256) void dwconv_bwd_data_kernel(const float* __restrict__ dy,
                                                              const float* __restrict__ wp,
                                                              const float* __restrict__ addend, float* __restrict__ dx,
                                                              int B, int H, int W, int Cs, int Ho, int Wo, int K,
                                                              int stride, int pad, long long total4) {
  const int CQ = Cs >> 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total4;
       i += (long long)gridDim.x * blockDim.x) {
    const int q = (int)(i % CQ);
    const long long pix = i / CQ;
    const int w = (int)(pix % W), h = (int)((pix / W) % H), b = (int)(pix / ((long long)W * H));
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int dh = 0; dh < K; ++dh) {
      const int hn = h + pad - dh;
      if (hn < 0 || hn % stride) continue;
      const int ho = hn / stride;
      if (ho >= Ho) continue;
      for (int dw = 0; dw < K; ++dw) {
        const int wn = w + pad - dw;
        if (wn < 0 || wn % stride) continue;
        const int wo = wn / stride;
        if (wo >= Wo) continue;
        const f32x4 g = *reinterpret_cast<const f32x4*>(dy + ((size_t)(b * Ho + ho) * Wo + wo) * Cs + (size_t)q * 4);
        const f32x4 wv = *reinterpret_cast<const f32x4*>(wp + (size_t)(dh * K + dw) * Cs + (size_t)q * 4);
        acc += g * wv;
      }
    }
    if (addend != nullptr) acc += reinterpret_cast<const f32x4*>(addend)[i];
    reinterpret_cast<f32x4*>(dx)[i] = acc;
  }
}

// The data gradient for the encoder's shapes (K in {3, 5}, stride in {1, 2}, even H and W) with the tap loops
// resolved at compile time.  Stride 1: a thread produces two horizontally adjacent pixels from K rows of K+1 shared
// dy columns.  Stride 2: a thread produces a 2x2 block of dx; which taps reach a pixel depends only on its parity, so
// the block needs the (K+1)/2 x (K+1)/2 dy pixels around it once (4 loads for 3x3, 9 for 5x5, instead of a
// run-time loop over K*K taps with a divisibility test per tap and pixel).
template <int K, int S>
__global__ __launch_bounds__(256) void dwconv_bwd_data_tpl_kernel(const float* __restrict__ dy,
                                                                  const float* __restrict__ wp,
                                                                  const float* __restrict__ addend,
                                                                  float* __restrict__ dx, int B, int H, int W, int Cs,
                                                                  int Ho, int Wo, long long total) {
  constexpr int PAD = (K - 1) / 2;
  const int CQ = Cs >> 2;
  const int H2 = S == 2 ? H >> 1 : H, W2 = W >> 1;  // thread grid: stride 1 -> (H, W/2) pairs, stride 2 -> 2x2 blocks
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int q = (int)(idx % CQ);
    const long long pix = idx / CQ;
    const int j = (int)(pix % W2), i = (int)((pix / W2) % H2), b = (int)(pix / ((long long)W2 * H2));
    const float* dyq = dy + (size_t)q * 4;
    const float* wq = wp + (size_t)q * 4;
    if constexpr (S == 1) {
      const int h = i, w = 2 * j;
      f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
#pragma unroll
      for (int dh = 0; dh < K; ++dh) {
        const int hn = h + PAD - dh;
        const bool hok = (unsigned)hn < (unsigned)Ho;
        const int rowoff = (b * Ho + (hok ? hn : 0)) * Wo;
        f32x4 g[K + 1];
#pragma unroll
        for (int c = 0; c <= K; ++c) {
          const int wn = w - PAD + c;
          const bool ok = hok && (unsigned)wn < (unsigned)Wo;
          g[c] = *reinterpret_cast<const f32x4*>(dyq + (size_t)(rowoff + (ok ? wn : 0)) * Cs);
          if (!ok) g[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int c = 0; c <= K; ++c) {
          // pixel w reads column wn = w + PAD - dw: dw = K - 1 - c; pixel w + 1: dw = K - c
          if (c < K) a0 += g[c] * *reinterpret_cast<const f32x4*>(wq + (size_t)(dh * K + K - 1 - c) * Cs);
          if (c >= 1) a1 += g[c] * *reinterpret_cast<const f32x4*>(wq + (size_t)(dh * K + K - c) * Cs);
        }
      }
      const size_t oo = ((size_t)(b * H + h) * W + w) * Cs + (size_t)q * 4;
      if (addend != nullptr) {
        a0 += *reinterpret_cast<const f32x4*>(addend + oo);
        a1 += *reinterpret_cast<const f32x4*>(addend + oo + Cs);
      }
      *reinterpret_cast<f32x4*>(dx + oo) = a0;
      *reinterpret_cast<f32x4*>(dx + oo + Cs) = a1;
    } else {
      // pixel row h = 2i + ph takes tap dh iff ph + PAD - dh is even, from dy row i + (ph + PAD - dh) / 2
      constexpr int OMIN = -((K - 1 - PAD) / 2), OMAX = (1 + PAD) / 2, NO = OMAX - OMIN + 1;
      f32x4 g[NO][NO];
#pragma unroll
      for (int r = 0; r < NO; ++r) {
        const int ho = i + OMIN + r;
        const bool hok = (unsigned)ho < (unsigned)Ho;
#pragma unroll
        for (int c = 0; c < NO; ++c) {
          const int wo = j + OMIN + c;
          const bool ok = hok && (unsigned)wo < (unsigned)Wo;
          g[r][c] = *reinterpret_cast<const f32x4*>(dyq + (size_t)((b * Ho + (hok ? ho : 0)) * Wo + (ok ? wo : 0)) * Cs);
          if (!ok) g[r][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
      }
#pragma unroll
      for (int ph = 0; ph < 2; ++ph) {
#pragma unroll
        for (int pw = 0; pw < 2; ++pw) {
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int dh = 0; dh < K; ++dh) {
            if ((ph + PAD - dh) & 1) continue;
            const int r = (ph + PAD - dh) / 2 - OMIN;  // (ph + PAD - dh) is even here: exact for negatives too
#pragma unroll
            for (int dw = 0; dw < K; ++dw) {
              if ((pw + PAD - dw) & 1) continue;
              const int c = (pw + PAD - dw) / 2 - OMIN;
              acc += g[r][c] * *reinterpret_cast<const f32x4*>(wq + (size_t)(dh * K + dw) * Cs);
            }
          }
          const size_t oo = ((size_t)(b * H + 2 * i + ph) * W + 2 * j + pw) * Cs + (size_t)q * 4;
          if (addend != nullptr) acc += *reinterpret_cast<const f32x4*>(addend + oo);
          *reinterpret_cast<f32x4*>(dx + oo) = acc;
        }
      }
    }
  }
}

// weight gradient: dw[c][tap] = sum over output pixels of dy * x(shifted).  Grid = (channel panels of
// QL float4 quads, row blocks); a workgroup's 256 threads are QL quad lanes x (256/QL) row lanes with
// QL = min(16, pow2 >= quads) so narrow maps (C = 16..72) still use every lane.  Each thread keeps its
// K*K tap sums in registers over its rows; row lanes are folded with wave shuffles + one LDS pass and
// every workgroup writes one partial row [row block][tap][Cs]; the finalize sums row blocks in fp64.
#define DWW_MAX_RB 512

template <int KK>
__global__ __launch_bounds__(256) void dwconv_bwd_w_kernel(const float* __restrict__ x,
                                                           const float* __restrict__ dy, int H, int W, int Cs,
                                                           int Ho, int Wo, int K, int stride, int pad, int M, int QL,
                                                           float* partial) {
  __shared__ f32x4 red[4][KK][16];  // [wave][tap][quad lane]: ONE barrier for all taps (it was two per tap)
  const int CQ = Cs >> 2;
  const int ql = threadIdx.x & (QL - 1);  // quad lane
  const int rl = threadIdx.x / QL;        // row lane 0 .. 256/QL-1
  const int RL = 256 / QL;
  const int q = blockIdx.x * QL + ql;
  const int nrb = gridDim.y;
  const int rows_per_blk = (M + nrb - 1) / nrb;
  const int r_begin = blockIdx.y * rows_per_blk;
  const int r_end = min(M, r_begin + rows_per_blk);
  f32x4 acc[KK];
#pragma unroll
  for (int t = 0; t < KK; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (q < CQ)
    for (int r = r_begin + rl; r < r_end; r += RL) {
      const int wo = r % Wo, ho = (r / Wo) % Ho, b = r / (Wo * Ho);
      const f32x4 g = *reinterpret_cast<const f32x4*>(dy + (size_t)r * Cs + (size_t)q * 4);
#pragma unroll
      for (int t = 0; t < KK; ++t) {
        const int dh = t / K, dw = t - dh * K;
        const int h = ho * stride - pad + dh, w = wo * stride - pad + dw;
        if ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W)
          acc[t] += g * *reinterpret_cast<const f32x4*>(x + ((size_t)(b * H + h) * W + w) * Cs + (size_t)q * 4);
      }
    }
  const int wave = threadIdx.x >> 6;
#pragma unroll
  for (int t = 0; t < KK; ++t) {
    f32x4 v = acc[t];
    for (int o = QL; o < 64; o <<= 1) {  // fold the row lanes of this wave (lane bits >= log2(QL))
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += __shfl_xor(v[e], o, 64);
    }
    if ((int)(threadIdx.x & 63) < QL) red[wave][t][ql] = v;
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < KK * QL; idx += 256) {
    const int t = idx / QL, l = idx - t * QL;
    const int qq = blockIdx.x * QL + l;
    if (qq < CQ) {
      const f32x4 s = ((red[0][t][l] + red[1][t][l]) + red[2][t][l]) + red[3][t][l];
      *reinterpret_cast<f32x4*>(partial + ((size_t)blockIdx.y * KK + t) * Cs + (size_t)qq * 4) = s;
    }
  }
}

// The same partial sums for the encoder's shapes (K in {3, 5}, stride in {1, 2}, even Wo): taps unrolled at compile
// time, and a thread takes TWO horizontally adjacent output pixels per pass, whose K+S input columns per tap row are
// loaded once for both (2 dy + K*(K+S) x loads instead of 2 + 2*K*K).
template <int K, int S>
__global__ __launch_bounds__(256) void dwconv_bwd_w_pair_kernel(const float* __restrict__ x,
                                                                const float* __restrict__ dy, int H, int W, int Cs,
                                                                int Ho, int Wo, int M, int QL, int rows_per_blk,
                                                                float* partial) {
  constexpr int KK = K * K, PAD = (K - 1) / 2, NC = K + S;
  __shared__ f32x4 red[4][KK][16];
  const int CQ = Cs >> 2;
  const int ql = threadIdx.x & (QL - 1);  // quad lane
  const int rl = threadIdx.x / QL;        // row lane 0 .. 256/QL-1
  const int RL = 256 / QL;
  const int q = blockIdx.x * QL + ql;
  const int r_begin = blockIdx.y * rows_per_blk;  // even (host)
  const int r_end = min(M, r_begin + rows_per_blk);
  f32x4 acc[KK];
#pragma unroll
  for (int t = 0; t < KK; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (q < CQ) {
    const float* xq = x + (size_t)q * 4;
    for (int r = r_begin + 2 * rl; r < r_end; r += 2 * RL) {  // outputs r, r + 1: the same row (Wo is even)
      const int wo = r % Wo, ho = (r / Wo) % Ho, b = r / (Wo * Ho);
      const f32x4 g0 = *reinterpret_cast<const f32x4*>(dy + (size_t)r * Cs + (size_t)q * 4);
      const f32x4 g1 = *reinterpret_cast<const f32x4*>(dy + (size_t)(r + 1) * Cs + (size_t)q * 4);
#pragma unroll
      for (int dh = 0; dh < K; ++dh) {
        const int h = ho * S - PAD + dh;
        const bool hok = (unsigned)h < (unsigned)H;
        const int rowoff = (b * H + (hok ? h : 0)) * W;
        f32x4 v[NC];
#pragma unroll
        for (int j = 0; j < NC; ++j) {
          const int w = wo * S - PAD + j;
          const bool ok = hok && (unsigned)w < (unsigned)W;
          v[j] = *reinterpret_cast<const f32x4*>(xq + (size_t)(rowoff + (ok ? w : 0)) * Cs);
          if (!ok) v[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int j = 0; j < NC; ++j) {
          if (j < K) acc[dh * K + j] += g0 * v[j];
          if (j >= S) acc[dh * K + j - S] += g1 * v[j];
        }
      }
    }
  }
  const int wave = threadIdx.x >> 6;
#pragma unroll
  for (int t = 0; t < KK; ++t) {
    f32x4 v = acc[t];
    for (int o = QL; o < 64; o <<= 1) {  // fold the row lanes of this wave (lane bits >= log2(QL))
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += __shfl_xor(v[e], o, 64);
    }
    if ((int)(threadIdx.x & 63) < QL) red[wave][t][ql] = v;
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < KK * QL; idx += 256) {
    const int t = idx / QL, l = idx - t * QL;
    const int qq = blockIdx.x * QL + l;
    if (qq < CQ) {
      const f32x4 s = ((red[0][t][l] + red[1][t][l]) + red[2][t][l]) + red[3][t][l];
      *reinterpret_cast<f32x4*>(partial + ((size_t)blockIdx.y * KK + t) * Cs + (size_t)qq * 4) = s;
    }
  }
}

// dw[c][tap] (torch layout (C,1,K,K)) = sum over partial rows, fp64, fixed order.  A workgroup owns 32 consecutive
// (tap, channel) elements x 8 row groups: consecutive threads read consecutive channels of a partial row, each sums
// every 8th row, LDS folds the 8 groups.  (One workgroup per element launched C*K*K = 24,000 workgroups of two
// barriers for the 960-channel 5x5 layers; one thread per element left a 256-load serial chain: 32 us.)
template <int EG>  // (tap, channel) elements per workgroup; 256 / EG row groups
__global__ __launch_bounds__(256) void dwconv_bwd_w_finalize_kernel(const float* __restrict__ partial, int nblk,
                                                                    int KK, int C, int Cs, float* dw) {
  constexpr int RG = 256 / EG;
  __shared__ double sh[RG][EG];
  const int e = threadIdx.x % EG, rg = threadIdx.x / EG;
  const int i = blockIdx.x * EG + e;
  double s = 0.0;
  int t = 0, c = 0;
  if (i < KK * C) {
    t = i / C;
    c = i - t * C;
#pragma unroll 4
    for (int b = rg; b < nblk; b += RG) s += (double)partial[((size_t)b * KK + t) * Cs + c];
  }
  sh[rg][e] = s;
  __syncthreads();
  if (rg == 0 && i < KK * C) {
    double tot = 0.0;
#pragma unroll
    for (int k = 0; k < RG; ++k) tot += sh[k][e];
    dw[c * KK + t] = (float)tot;
  }
}

static inline int dw_grid(long long total4) {
  long long nb = cdivll(total4, 256);
  if (nb > 8192) nb = 8192;
  if (nb < 1) nb = 1;
  return (int)nb;
}

static int dw_check(int B, int H, int W, int Cs, int Ho, int Wo, int K, int stride, int pad) {
  if (B <= 0 || H <= 0 || W <= 0 || (Cs & 3) || Cs <= 0) return VMTL_ERR_ARG;
  if ((K != 3 && K != 5) || (stride != 1 && stride != 2) || pad < 0) return VMTL_ERR_ARG;
  if ((H + 2 * pad - K) / stride + 1 != Ho || (W + 2 * pad - K) / stride + 1 != Wo) return VMTL_ERR_ARG;
  if ((long long)B * H * W > 0x7fffffffLL) return VMTL_ERR_ARG;
  return VMTL_OK;
}

extern "C" int vmtl_dwconv_fwd(const float* x, const float* wp, float* y, int B, int H, int W, int Cs, int Ho, int Wo,
                               int K, int stride, int pad, void* stream) {
  VMTL_ENTER();
  if (!x || !wp || !y) return VMTL_ERR_ARG;
  if (int e = dw_check(B, H, W, Cs, Ho, Wo, K, stride, pad)) return e;
  const long long total4 = (long long)B * Ho * Wo * (Cs >> 2);
  hipLaunchKernelGGL(dwconv_fwd_kernel, dim3(dw_grid(total4)), dim3(256), 0, (hipStream_t)stream, x, wp, y, B, H, W,
                     Cs, Ho, Wo, K, stride, pad, total4);
  return vmtl_check_launch();
}

// rows of the statistics tensor the fused forward writes ([rows][2][Cs]); every row covers vmtl_dwconv_bn_stats_block
// output pixels (the last one fewer)
// Workgroups (= statistics rows) of the fused forward.  A thread keeps one channel quad and walks
// rows_per_blk / (256 / quads) output pixels x K*K taps serially, so - unlike the pure reductions, which use
// red_blocks() - the row blocks must stay SHORT for the deep layers (M = 1024 pixels x 240 quads left 64 workgroups
// of 16 serial pixels each: slower than the three launches it replaces): two pixels per thread, at most 1024 rows.
static int dwbn_rows(int M, int Cs) {
  const int cq = Cs >> 2;
  const int rpt = cq >= RED_THREADS ? 1 : RED_THREADS / cq;
  static int per_thread = -1;
  if (per_thread < 0) {
    const char* e = getenv("VMTL_DWBN_ROWS");  // tuning aid: output pixels per thread
    per_thread = e ? atoi(e) : 2;
    if (per_thread < 1) per_thread = 1;
  }
  int rows = per_thread * rpt;
  if (rows < cdiv(M, 1024)) rows = cdiv(M, 1024);
  return (rows + 1) & ~1;  // even: the pair kernel's blocks start on an even output pixel
}
extern "C" int vmtl_dwconv_bn_stats_rows(int M, int Cs) { return cdiv(M, dwbn_rows(M, Cs)); }
extern "C" int vmtl_dwconv_bn_stats_block(int M, int Cs) { return dwbn_rows(M, Cs); }

extern "C" int vmtl_dwconv_bn_fwd(const float* x, const float* coef_a, const float* coef_c, int act, const float* wp,
                                  float* y, float* a_out, float* partial, int B, int H, int W, int Cs, int Ho, int Wo, int K,
                                  int stride, int pad, void* stream) {
  VMTL_ENTER();
  if (!x || !coef_a || !coef_c || !wp || !y) return VMTL_ERR_ARG;
  if (int e = dw_check(B, H, W, Cs, Ho, Wo, K, stride, pad)) return e;
  if (pad != (K - 1) / 2) return VMTL_ERR_ARG;  // the a_out ownership rule assumes "same"-style padding
  const int M = B * Ho * Wo;
  const int rows = dwbn_rows(M, Cs);
  const int nblk = cdiv(M, rows);
  static const bool generic_only = getenv("VMTL_DWBN_GENERIC") != nullptr;  // tuning aid: the run-time-K kernel
  if (!generic_only && (Wo & 1) == 0 && (K == 3 || K == 5) && (stride == 1 || stride == 2) &&
      (act == VMTL_ACT_NONE || act == VMTL_ACT_RELU || act == VMTL_ACT_HSWISH)) {
    // 5x5 weights through LDS when the workgroup's pixels (rows x its share of the channel quads) outnumber the
    // K*K*Cs floats it would copy
    const bool wlds = K == 5 && rows >= 12;
#define CALLP(A, KV, SV, WL)                                                                                           \
  hipLaunchKernelGGL((dwconv_bn_fwd_pair_kernel<A, KV, SV, WL>), dim3(nblk), dim3(RED_THREADS),                        \
                     WL ? (size_t)KV * KV * Cs * sizeof(float) : 0, (hipStream_t)stream, x, coef_a, coef_c, wp, y,     \
                     a_out, partial, H, W, Cs, Ho, Wo, M, rows)
#define CALLKS(A)                                              \
  do {                                                         \
    if (K == 3 && stride == 1) CALLP(A, 3, 1, false);          \
    else if (K == 3) CALLP(A, 3, 2, false);                    \
    else if (stride == 1 && wlds) CALLP(A, 5, 1, true);        \
    else if (stride == 1) CALLP(A, 5, 1, false);               \
    else if (wlds) CALLP(A, 5, 2, true);                       \
    else CALLP(A, 5, 2, false);                                \
  } while (0)
    if (act == VMTL_ACT_NONE) CALLKS(VMTL_ACT_NONE);
    else if (act == VMTL_ACT_RELU) CALLKS(VMTL_ACT_RELU);
    else CALLKS(VMTL_ACT_HSWISH);
#undef CALLKS
#undef CALLP
    return vmtl_check_launch();
  }
#define CALL(A)                                                                                                       \
  hipLaunchKernelGGL((dwconv_bn_fwd_kernel<A>), dim3(nblk), dim3(RED_THREADS), 0, (hipStream_t)stream, x, coef_a, coef_c, \
                     wp, y, a_out, partial, H, W, Cs, Ho, Wo, K, stride, pad, M, rows)
  switch (act) {
    case VMTL_ACT_NONE: CALL(VMTL_ACT_NONE); break;
    case VMTL_ACT_RELU: CALL(VMTL_ACT_RELU); break;
    case VMTL_ACT_HSWISH: CALL(VMTL_ACT_HSWISH); break;
    default: return VMTL_ERR_ARG;
  }
#undef CALL
  return vmtl_check_launch();
}

// dx = dwconv^T(dy) [+ addend]: addend (nullable, dx's shape) is a second gradient of the same tensor - the residual
// branch of a block whose input also feeds the depthwise conv - added on the way out instead of in its own pass
static int dw_bwd_data_impl(const float* dy, const float* wp, const float* addend, float* dx, int B, int H, int W, int Cs,
                            int Ho, int Wo, int K, int stride, int pad, void* stream) {
  if (!dy || !wp || !dx) return VMTL_ERR_ARG;
  if (int e = dw_check(B, H, W, Cs, Ho, Wo, K, stride, pad)) return e;
  const long long total4 = (long long)B * H * W * (Cs >> 2);
  static const bool generic_only = getenv("VMTL_DWBN_GENERIC") != nullptr;  // tuning aid: the run-time-K kernel
  if (!generic_only && pad == (K - 1) / 2 && (K == 3 || K == 5) && (W & 1) == 0 &&
      (stride == 1 || (stride == 2 && (H & 1) == 0 && Ho == H / 2 && Wo == W / 2))) {
    const long long total = stride == 1 ? total4 / 2 : total4 / 4;
#define CALLD(KV, SV)                                                                                              \
  hipLaunchKernelGGL((dwconv_bwd_data_tpl_kernel<KV, SV>), dim3(dw_grid(total)), dim3(256), 0, (hipStream_t)stream, dy, \
                     wp, addend, dx, B, H, W, Cs, Ho, Wo, total)
    if (K == 3 && stride == 1) CALLD(3, 1);
    else if (K == 3) CALLD(3, 2);
    else if (stride == 1) CALLD(5, 1);
    else CALLD(5, 2);
#undef CALLD
    return vmtl_check_launch();
  }
  hipLaunchKernelGGL(dwconv_bwd_data_kernel, dim3(dw_grid(total4)), dim3(256), 0, (hipStream_t)stream, dy, wp, addend, dx,
                     B, H, W, Cs, Ho, Wo, K, stride, pad, total4);
  return vmtl_check_launch();
}

extern "C" int vmtl_dwconv_bwd_data(const float* dy, const float* wp, float* dx, int B, int H, int W, int Cs, int Ho,
                                    int Wo, int K, int stride, int pad, void* stream) {
  VMTL_ENTER();
  return dw_bwd_data_impl(dy, wp, nullptr, dx, B, H, W, Cs, Ho, Wo, K, stride, pad, stream);
}

extern "C" int vmtl_dwconv_bwd_data_add(const float* dy, const float* wp, const float* addend, float* dx, int B, int H,
                                        int W, int Cs, int Ho, int Wo, int K, int stride, int pad, void* stream) {
  VMTL_ENTER();
  if (!addend) return VMTL_ERR_ARG;
  return dw_bwd_data_impl(dy, wp, addend, dx, B, H, W, Cs, Ho, Wo, K, stride, pad, stream);
}

// row blocks of the weight gradient's first stage: ~2048 workgroups in total (the 64x128 maps ran 256 workgroups = one
// wave per SIMD at 0.5 TB/s), at least two passes per row lane, at most DWW_MAX_RB partial rows for the finalize
static void dww_geometry(int M, int Cs, int* QLo, int* panels_o, int* rows_o, int* nblk_o) {
  const int CQ = Cs >> 2;
  int QL = 16;
  while (QL > 1 && (QL >> 1) >= CQ) QL >>= 1;  // smallest power of two >= CQ, at most 16
  const int panels = cdiv(CQ, QL);
  const int rows_per_pass = 2 * (256 / QL);  // two output pixels per row lane and pass
  int nblk = cdiv(2048, panels);
  if (nblk > cdiv(M, 2 * rows_per_pass)) nblk = cdiv(M, 2 * rows_per_pass);
  if (nblk > DWW_MAX_RB) nblk = DWW_MAX_RB;
  if (nblk < 1) nblk = 1;
  int rows = cdiv(M, nblk);
  rows = (rows + 1) & ~1;  // even: the pair kernel's blocks start on an even output pixel
  *QLo = QL; *panels_o = panels; *rows_o = rows; *nblk_o = cdiv(M, rows);
}

extern "C" int vmtl_dwconv_bwd_weight_rows(int M, int Cs) {
  int QL, panels, rows, nblk;
  dww_geometry(M, Cs, &QL, &panels, &rows, &nblk);
  return nblk;
}

// partial: vmtl_dwconv_bwd_weight_rows(B*Ho*Wo, Cs) * K*K * Cs floats.  dw: torch (C,1,K,K) layout.
extern "C" int vmtl_dwconv_bwd_weight(const float* x, const float* dy, float* partial, float* dw, int B, int H, int W,
                                      int C, int Cs, int Ho, int Wo, int K, int stride, int pad, void* stream) {
  VMTL_ENTER();
  if (!x || !dy || !partial || !dw || C <= 0 || C > Cs) return VMTL_ERR_ARG;
  if (int e = dw_check(B, H, W, Cs, Ho, Wo, K, stride, pad)) return e;
  hipStream_t st = (hipStream_t)stream;
  const int M = B * Ho * Wo;
  int QL, panels, rows, nblk;
  dww_geometry(M, Cs, &QL, &panels, &rows, &nblk);
  static const bool generic_only = getenv("VMTL_DWBN_GENERIC") != nullptr;  // tuning aid: the run-time-K kernel
  if (!generic_only && (Wo & 1) == 0 && pad == (K - 1) / 2 && (K == 3 || K == 5) && (stride == 1 || stride == 2)) {
#define CALLW(KV, SV)                                                                                              \
  hipLaunchKernelGGL((dwconv_bwd_w_pair_kernel<KV, SV>), dim3(panels, nblk), dim3(256), 0, st, x, dy, H, W, Cs, Ho, Wo, \
                     M, QL, rows, partial)
    if (K == 3 && stride == 1) CALLW(3, 1);
    else if (K == 3) CALLW(3, 2);
    else if (stride == 1) CALLW(5, 1);
    else CALLW(5, 2);
#undef CALLW
  } else if (K == 3) {
    hipLaunchKernelGGL((dwconv_bwd_w_kernel<9>), dim3(panels, nblk), dim3(256), 0, st, x, dy, H, W, Cs, Ho, Wo, K,
                       stride, pad, M, QL, partial);
  } else {
    hipLaunchKernelGGL((dwconv_bwd_w_kernel<25>), dim3(panels, nblk), dim3(256), 0, st, x, dy, H, W, Cs, Ho, Wo, K,
                       stride, pad, M, QL, partial);
  }
  // few elements (narrow early layers, which also have the most partial rows): more row groups per element
  if (C * K * K >= 4096)
    hipLaunchKernelGGL(dwconv_bwd_w_finalize_kernel<32>, dim3(cdiv(C * K * K, 32)), dim3(256), 0, st, partial, nblk,
                       K * K, C, Cs, dw);
  else
    hipLaunchKernelGGL(dwconv_bwd_w_finalize_kernel<8>, dim3(cdiv(C * K * K, 8)), dim3(256), 0, st, partial, nblk,
                       K * K, C, Cs, dw);
  return vmtl_check_launch();
}
