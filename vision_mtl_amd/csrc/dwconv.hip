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
    int stride, int pad, int M) {
  __shared__ f32x4 red_mean[RED_THREADS];
  __shared__ f32x4 red_m2[RED_THREADS];
  __shared__ float red_n[RED_THREADS];
  const int CQ = Cs >> 2;
  const int nblk = gridDim.x;
  const int rows_per_blk = (M + nblk - 1) / nblk;
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

__global__ __launch_bounds__(256) void dwconv_bwd_data_kernel(const float* __restrict__ dy,
                                                              const float* __restrict__ wp, float* __restrict__ dx,
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
    reinterpret_cast<f32x4*>(dx)[i] = acc;
  }
}

// weight gradient: dw[c][tap] = sum over output pixels of dy * x(shifted).  Grid = (channel panels of
// QL float4 quads, row blocks); a workgroup's 256 threads are QL quad lanes x (256/QL) row lanes with
// QL = min(16, pow2 >= quads) so narrow maps (C = 16..72) still use every lane.  Each thread keeps its
// K*K tap sums in registers over its rows; row lanes are folded with wave shuffles + one LDS pass and
// every workgroup writes one partial row [row block][tap][Cs]; the finalize sums row blocks in fp64.
#define DWW_MAX_RB 256

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

// dw[c][tap] (torch layout (C,1,K,K)) = sum over partial rows, fp64, fixed order.  A workgroup owns 32 consecutive
// (tap, channel) elements x 8 row groups: consecutive threads read consecutive channels of a partial row, each sums
// every 8th row, LDS folds the 8 groups.  (One workgroup per element launched C*K*K = 24,000 workgroups of two
// barriers for the 960-channel 5x5 layers; one thread per element left a 256-load serial chain: 32 us.)
__global__ __launch_bounds__(256) void dwconv_bwd_w_finalize_kernel(const float* __restrict__ partial, int nblk,
                                                                    int KK, int C, int Cs, float* dw) {
  __shared__ double sh[8][32];
  const int e = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + e;
  double s = 0.0;
  int t = 0, c = 0;
  if (i < KK * C) {
    t = i / C;
    c = i - t * C;
    for (int b = rg; b < nblk; b += 8) s += (double)partial[((size_t)b * KK + t) * Cs + c];
  }
  sh[rg][e] = s;
  __syncthreads();
  if (rg == 0 && i < KK * C) {
    double tot = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) tot += sh[k][e];
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
static int dwbn_blocks(int M, int Cs) {
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
  return cdiv(M, rows);
}
extern "C" int vmtl_dwconv_bn_stats_rows(int M, int Cs) { return dwbn_blocks(M, Cs); }
extern "C" int vmtl_dwconv_bn_stats_block(int M, int Cs) { return cdiv(M, dwbn_blocks(M, Cs)); }

extern "C" int vmtl_dwconv_bn_fwd(const float* x, const float* coef_a, const float* coef_c, int act, const float* wp,
                                  float* y, float* a_out, float* partial, int B, int H, int W, int Cs, int Ho, int Wo, int K,
                                  int stride, int pad, void* stream) {
  VMTL_ENTER();
  if (!x || !coef_a || !coef_c || !wp || !y) return VMTL_ERR_ARG;
  if (int e = dw_check(B, H, W, Cs, Ho, Wo, K, stride, pad)) return e;
  if (pad != (K - 1) / 2) return VMTL_ERR_ARG;  // the a_out ownership rule assumes "same"-style padding
  const int M = B * Ho * Wo;
  const int nblk = dwbn_blocks(M, Cs);
#define CALL(A)                                                                                                       \
  hipLaunchKernelGGL((dwconv_bn_fwd_kernel<A>), dim3(nblk), dim3(RED_THREADS), 0, (hipStream_t)stream, x, coef_a, coef_c, \
                     wp, y, a_out, partial, H, W, Cs, Ho, Wo, K, stride, pad, M)
  switch (act) {
    case VMTL_ACT_NONE: CALL(VMTL_ACT_NONE); break;
    case VMTL_ACT_RELU: CALL(VMTL_ACT_RELU); break;
    case VMTL_ACT_HSWISH: CALL(VMTL_ACT_HSWISH); break;
    default: return VMTL_ERR_ARG;
  }
#undef CALL
  return vmtl_check_launch();
}

extern "C" int vmtl_dwconv_bwd_data(const float* dy, const float* wp, float* dx, int B, int H, int W, int Cs, int Ho,
                                    int Wo, int K, int stride, int pad, void* stream) {
  VMTL_ENTER();
  if (!dy || !wp || !dx) return VMTL_ERR_ARG;
  if (int e = dw_check(B, H, W, Cs, Ho, Wo, K, stride, pad)) return e;
  const long long total4 = (long long)B * H * W * (Cs >> 2);
  hipLaunchKernelGGL(dwconv_bwd_data_kernel, dim3(dw_grid(total4)), dim3(256), 0, (hipStream_t)stream, dy, wp, dx, B,
                     H, W, Cs, Ho, Wo, K, stride, pad, total4);
  return vmtl_check_launch();
}

// partial: at least 256 * K*K * Cs floats (vmtl_reduce_rows(M) * K*K * Cs always suffices).  dw: torch (C,1,K,K) layout.
extern "C" int vmtl_dwconv_bwd_weight(const float* x, const float* dy, float* partial, float* dw, int B, int H, int W,
                                      int C, int Cs, int Ho, int Wo, int K, int stride, int pad, void* stream) {
  VMTL_ENTER();
  if (!x || !dy || !partial || !dw || C <= 0 || C > Cs) return VMTL_ERR_ARG;
  if (int e = dw_check(B, H, W, Cs, Ho, Wo, K, stride, pad)) return e;
  hipStream_t st = (hipStream_t)stream;
  const int M = B * Ho * Wo;
  const int CQ = Cs >> 2;
  int QL = 16;
  while (QL > 1 && (QL >> 1) >= CQ) QL >>= 1;  // smallest power of two >= CQ, at most 16
  const int panels = cdiv(CQ, QL);
  const int rows_per_pass = 256 / QL;
  int nblk = cdiv(1024, panels);                                           // ~1024 workgroups in total
  if (nblk > cdiv(M, 4 * rows_per_pass)) nblk = cdiv(M, 4 * rows_per_pass);  // >= 4 rows per row lane
  if (nblk > DWW_MAX_RB) nblk = DWW_MAX_RB;
  if (nblk < 1) nblk = 1;
  if (K == 3)
    hipLaunchKernelGGL((dwconv_bwd_w_kernel<9>), dim3(panels, nblk), dim3(256), 0, st, x, dy, H, W, Cs, Ho, Wo, K,
                       stride, pad, M, QL, partial);
  else
    hipLaunchKernelGGL((dwconv_bwd_w_kernel<25>), dim3(panels, nblk), dim3(256), 0, st, x, dy, H, W, Cs, Ho, Wo, K,
                       stride, pad, M, QL, partial);
  hipLaunchKernelGGL(dwconv_bwd_w_finalize_kernel, dim3(cdiv(C * K * K, 32)), dim3(256), 0, st, partial, nblk, K * K,
                     C, Cs, dw);
  return vmtl_check_launch();
}
