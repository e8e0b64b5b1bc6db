// BatchNorm2d (train: batch statistics; eval: running statistics) fused with the
// activation that follows it, an optional elementwise gate operand (MTAN's
// `conv2_shared * sigmoid(bn2(..))`) and an optional residual add (MobileNetV3
// inverted-residual skip).  NHWC fp32, HBM-bound, 16-byte accesses.
//
// Replaces torch.nn.BatchNorm2d + ReLU/Hardswish/Sigmoid + mul/add as invoked at
//   reference vision_mtl/utils/model_utils.py:72-76 (DoubleConv),
//   reference vision_mtl/models/mtan_model.py:67-81,139-167 (attention modules),
//   smp Conv2dReLU / timm BatchNormAct2d (reference vision_mtl/utils/model_utils.py:25-34).
//
// Per-channel reductions are two-stage and deterministic: stage 1 writes one
// partial row per workgroup ([nblk][K][Cs]); the finalize kernels sum the rows in
// fp64 in a fixed order.
#include "common.h"

#include "reduce.h"

extern "C" int vmtl_reduce_rows(int M) { return red_blocks(M); }

// ---------------------------------------------------------------- statistics
// Per-block partial rows are (mean_b, M2_b) over the block's rows, merged by Chan's formula in
// fp64 (never E[x^2] - E[x]^2).  Each thread accumulates SHIFTED sums around the first value it
// sees, which keeps fp32 accurate even when |mean| >> std.
__global__ __launch_bounds__(RED_THREADS) void bn_stats_kernel(const float* __restrict__ x, int M, int Cs,
                                                               float* partial) {
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
    f32x4 K = {0.f, 0.f, 0.f, 0.f}, s1 = K, s2 = K;
    float n = 0.f;
    if (t < T)
      for (int r = r_begin + ro; r < r_end; r += rpt) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + (size_t)r * Cs + (size_t)q * 4);
        if (n == 0.f) K = v;
        const f32x4 d = v - K;
        s1 += d;
        s2 += d * d;
        n += 1.f;
      }
    f32x4 mean = K, m2 = {0.f, 0.f, 0.f, 0.f};
    if (n > 0.f) {
      mean = K + s1 * (1.f / n);
      m2 = s2 - s1 * s1 * (1.f / n);
    }
    __syncthreads();
    red_mean[t] = mean;
    red_m2[t] = m2;
    red_n[t] = n;
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

// fp64 sums of N values per thread over a 256-thread workgroup: out[i] (i < N) valid on every thread after the
// call.  One barrier pair; fixed association order (run-to-run reproducible).
template <int N>
__device__ __forceinline__ void block_sum_n(double (&s)[N], double (*sh)[N] /* [4][N] */) {
#pragma unroll
  for (int i = 0; i < N; ++i) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s[i] += __shfl_xor(s[i], o, 64);
  }
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int i = 0; i < N; ++i) sh[threadIdx.x >> 6][i] = s[i];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < N; ++i) s[i] = (sh[0][i] + sh[1][i]) + (sh[2][i] + sh[3][i]);
}

// mean / invstd from the (mean_b, M2_b) rows; block b covers rows [b*rows_per_blk, min(M, (b+1)*rows_per_blk)).
// Optionally updates running stats the way torch does (momentum, unbiased variance).
// This launch sits on the dependent chain of EVERY BatchNorm of the forward pass (56 per `basic` step), where its
// cost is latency, not throughput: one workgroup per channel QUAD reads each row's (mean, M2) as two float4 (a
// workgroup per channel fetched a 128-byte line for 4 useful bytes: 31 us for the 8192 rows of the full-resolution
// convs), ONE pass (sums of n*(m - p) and M2 + n*(m - p)^2 about the pivot p = first row's mean, in fp64: the
// variance is then s2/M - (s1/M)^2 with nothing to cancel), one barrier pair, and the per-channel parameters of
// the four finishing lanes are requested before the reduction so their latency hides behind it.
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ partial, int nblk, int rows_per_blk,
                                                          int M, int C, int Cs, float eps, float momentum,
                                                          float* running_mean, float* running_var,
                                                          long long* num_batches_tracked, float* save_mean,
                                                          float* save_invstd, const float* gamma, const float* beta,
                                                          float* coef_a, float* coef_c) {
  __shared__ double sh[4][8];
  const int c0 = blockIdx.x * 4, t = threadIdx.x;
  const int c = c0 + t;
  const bool fin = t < 4 && c < C;
  float g = 1.f, bt = 0.f, orm = 0.f, orv = 0.f;
  if (fin) {
    if (gamma != nullptr && coef_a != nullptr) g = gamma[c];
    if (beta != nullptr && coef_a != nullptr) bt = beta[c];
    if (running_mean != nullptr) {
      orm = running_mean[c];
      orv = running_var[c];
    }
  }
  if (blockIdx.x == 0 && t == 0 && num_batches_tracked != nullptr) num_batches_tracked[0] += 1;
  const f32x4 pv = *reinterpret_cast<const f32x4*>(partial + c0);
  double s[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
  for (int b = t; b < nblk; b += 256) {
    const int nb = max(0, min(rows_per_blk, M - b * rows_per_blk));
    const float* row = partial + (size_t)b * 2 * Cs + c0;
    const f32x4 m4 = *reinterpret_cast<const f32x4*>(row);
    const f32x4 q4 = *reinterpret_cast<const f32x4*>(row + Cs);
    const double n = (double)nb;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const double d = nb > 0 ? (double)m4[e] - (double)pv[e] : 0.0;  // rows past M may hold anything
      s[e] += n * d;
      s[4 + e] += (nb > 0 ? (double)q4[e] : 0.0) + n * d * d;
    }
  }
  block_sum_n<8>(s, sh);
  if (t >= 4 || c >= Cs) return;
  if (!fin) {  // pad channel
    save_mean[c] = 0.f;
    save_invstd[c] = 0.f;
    if (coef_a != nullptr) coef_a[c] = coef_c[c] = 0.f;
    return;
  }
  const double dm = s[t] / M;
  const double mean = (double)pv[t] + dm;
  double var = s[4 + t] / M - dm * dm;
  if (var < 0.0) var = 0.0;
  const float fmean = (float)mean, fis = (float)(1.0 / sqrt(var + (double)eps));
  save_mean[c] = fmean;
  save_invstd[c] = fis;
  if (coef_a != nullptr) {  // y = act(coef_a * x + coef_c): the normalisation as the consumer conv's prologue
    const float sc = g * fis;
    coef_a[c] = sc;
    coef_c[c] = bt - fmean * sc;
  }
  if (running_mean != nullptr) {
    const double unb = M > 1 ? var * ((double)M / (double)(M - 1)) : var;
    running_mean[c] = (float)((1.0 - momentum) * orm + momentum * mean);
    running_var[c] = (float)((1.0 - momentum) * orv + momentum * unb);
  }
}

// eval mode: derive mean / invstd from the running buffers
__global__ void bn_eval_stats_kernel(const float* __restrict__ running_mean, const float* __restrict__ running_var,
                                     int C, int Cs, float eps, float* save_mean, float* save_invstd,
                                     const float* gamma, const float* beta, float* coef_a, float* coef_c) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= Cs) return;
  const float m = c < C ? running_mean[c] : 0.f;
  const float is = c < C ? 1.f / sqrtf(running_var[c] + eps) : 0.f;
  save_mean[c] = m;
  save_invstd[c] = is;
  if (coef_a != nullptr) {
    const float sc = c < C ? (gamma ? gamma[c] : 1.f) * is : 0.f;
    coef_a[c] = sc;
    coef_c[c] = c < C ? (beta ? beta[c] : 0.f) - m * sc : 0.f;
  }
}

static int bn_stats_impl(const float* x, int M, int C, int Cs, float* partial, int nblk_from_conv,
                         int rows_per_blk_from_conv, float eps, float momentum, float* running_mean, float* running_var,
                         long long* num_batches_tracked, float* save_mean, float* save_invstd, const float* gamma,
                         const float* beta, float* coef_a, float* coef_c, void* stream) {
  VMTL_ENTER();
  if ((coef_a == nullptr) != (coef_c == nullptr)) return VMTL_ERR_ARG;
  if (!partial || !save_mean || !save_invstd || M <= 0 || C <= 0 || C > Cs || (Cs & 3)) return VMTL_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  int nblk = nblk_from_conv, rows_per_blk = rows_per_blk_from_conv;
  if (nblk <= 0) {  // partial rows not produced by the conv epilogue: sweep x here
    if (!x) return VMTL_ERR_ARG;
    nblk = red_blocks(M);
    rows_per_blk = cdiv(M, nblk);
    hipLaunchKernelGGL(bn_stats_kernel, dim3(nblk), dim3(RED_THREADS), 0, st, x, M, Cs, partial);
  } else if (rows_per_blk <= 0 || (long long)nblk * rows_per_blk < M) {
    return VMTL_ERR_ARG;
  }
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(Cs >> 2), dim3(256), 0, st, partial, nblk, rows_per_blk, M, C, Cs, eps,
                     momentum, running_mean, running_var, num_batches_tracked, save_mean, save_invstd, gamma, beta,
                     coef_a, coef_c);
  return vmtl_check_launch();
}

extern "C" int vmtl_bn_stats(const float* x, int M, int C, int Cs, float* partial, int nblk_from_conv,
                             int rows_per_blk_from_conv, float eps, float momentum, float* running_mean, float* running_var,
                             long long* num_batches_tracked, float* save_mean, float* save_invstd, void* stream) {
  return bn_stats_impl(x, M, C, Cs, partial, nblk_from_conv, rows_per_blk_from_conv, eps, momentum, running_mean,
                       running_var, num_batches_tracked, save_mean, save_invstd, nullptr, nullptr, nullptr, nullptr, stream);
}

// vmtl_bn_stats that also emits the normalisation as prologue coefficients of the consumer conv
// (vmtl_conv3x3_small: act(coef_a[c] * x + coef_c[c]); zero on the pad channels)
extern "C" int vmtl_bn_stats_coef(const float* x, int M, int C, int Cs, float* partial, int nblk_from_conv,
                                  int rows_per_blk_from_conv, float eps, float momentum, float* running_mean,
                                  float* running_var, long long* num_batches_tracked, float* save_mean, float* save_invstd,
                                  const float* gamma, const float* beta, float* coef_a, float* coef_c, void* stream) {
  if (!coef_a || !coef_c) return VMTL_ERR_ARG;
  return bn_stats_impl(x, M, C, Cs, partial, nblk_from_conv, rows_per_blk_from_conv, eps, momentum, running_mean,
                       running_var, num_batches_tracked, save_mean, save_invstd, gamma, beta, coef_a, coef_c, stream);
}

extern "C" int vmtl_bn_eval_stats(const float* running_mean, const float* running_var, int C, int Cs, float eps,
                                  float* save_mean, float* save_invstd, void* stream) {
  VMTL_ENTER();
  if (!running_mean || !running_var || !save_mean || !save_invstd || C <= 0 || C > Cs) return VMTL_ERR_ARG;
  hipLaunchKernelGGL(bn_eval_stats_kernel, dim3(cdiv(Cs, 128)), dim3(128), 0, (hipStream_t)stream, running_mean,
                     running_var, C, Cs, eps, save_mean, save_invstd, nullptr, nullptr, nullptr, nullptr);
  return vmtl_check_launch();
}

extern "C" int vmtl_bn_eval_stats_coef(const float* running_mean, const float* running_var, int C, int Cs, float eps,
                                       float* save_mean, float* save_invstd, const float* gamma, const float* beta,
                                       float* coef_a, float* coef_c, void* stream) {
  VMTL_ENTER();
  if (!running_mean || !running_var || !save_mean || !save_invstd || !coef_a || !coef_c || C <= 0 || C > Cs)
    return VMTL_ERR_ARG;
  hipLaunchKernelGGL(bn_eval_stats_kernel, dim3(cdiv(Cs, 128)), dim3(128), 0, (hipStream_t)stream, running_mean,
                     running_var, C, Cs, eps, save_mean, save_invstd, gamma, beta, coef_a, coef_c);
  return vmtl_check_launch();
}

// ---------------------------------------------------------------- apply
// y = act(gamma * (x - mean) * invstd + beta) [* mul] [+ res];  pad channels -> 0.
// gamma/beta may be null (plain activation of x when mean/invstd are null too).
struct BnCoef {
  f32x4 sc, sh;  // z = x * sc + sh  (sc = 0 on pad channels)
  f32x4 valid;   // 1 for c < C else 0
};

__device__ __forceinline__ BnCoef bn_coef(int q, int C, const float* mean, const float* invstd, const float* gamma,
                                          const float* beta) {
  BnCoef k;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int c = q * 4 + e;
    float sc = 0.f, sh = 0.f, v = 0.f;
    if (c < C) {
      v = 1.f;
      if (mean != nullptr) {
        sc = (gamma ? gamma[c] : 1.f) * invstd[c];
        sh = (beta ? beta[c] : 0.f) - mean[c] * sc;
      } else {
        sc = 1.f;
      }
    }
    k.sc[e] = sc;
    k.sh[e] = sh;
    k.valid[e] = v;
  }
  return k;
}

template <int ACT>
__global__ __launch_bounds__(RED_THREADS) void bn_apply_kernel(const float* __restrict__ x,
                                                               const float* __restrict__ mean,
                                                               const float* __restrict__ invstd,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ beta,
                                                               const float* __restrict__ mul,
                                                               const float* __restrict__ res, float* __restrict__ y,
                                                               int M, int C, int Cs) {
  struct Row { f32x4 v, m, r; };
  column_sweep2(
      M, Cs >> 2, [&](int q) { return bn_coef(q, C, mean, invstd, gamma, beta); },
      [&](int r, int q) {
        const size_t off = (size_t)r * Cs + (size_t)q * 4;
        Row l;
        l.v = *reinterpret_cast<const f32x4*>(x + off);
        l.m = (f32x4){1.f, 1.f, 1.f, 1.f};
        l.r = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (mul != nullptr) l.m = *reinterpret_cast<const f32x4*>(mul + off);
        if (res != nullptr) l.r = *reinterpret_cast<const f32x4*>(res + off);
        return l;
      },
      [&](const Row& l, int r, int q, const BnCoef& k) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = k.valid[e] != 0.f ? act_fwd(l.v[e] * k.sc[e] + k.sh[e], ACT) : 0.f;
        if (mul != nullptr) o *= l.m;
        if (res != nullptr) o += l.r;
        *reinterpret_cast<f32x4*>(y + (size_t)r * Cs + (size_t)q * 4) = o;
      });
}

// Training-mode BatchNorm with FEW statistics rows (small feature maps: the deep half of the MobileNetV3
// encoder): finalize + apply in ONE launch.  Every thread merges the nblk (mean_b, M2_b) rows of its own
// channel quad with Chan's formula in fp64 (nblk <= VMTL_BN_FUSE_MAX_ROWS float4 pairs out of L2) instead
// of waiting for a separate one-workgroup-per-channel finalize launch; workgroup 0 also publishes
// mean / invstd for the backward pass and updates the running buffers.
#define VMTL_BN_FUSE_MAX_ROWS 64

template <int ACT>
__global__ __launch_bounds__(RED_THREADS) void bn_apply_fused_kernel(
    const float* __restrict__ x, const float* __restrict__ partial, int nblk, int rows_per_blk, float eps, float momentum,
    float* running_mean, float* running_var, long long* num_batches_tracked, float* save_mean, float* save_invstd,
    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mul,
    const float* __restrict__ res, float* __restrict__ y, int M, int C, int Cs) {
  if (blockIdx.x == 0 && threadIdx.x == 0 && num_batches_tracked != nullptr) num_batches_tracked[0] += 1;
  column_sweep(
      M, Cs >> 2,
      [&](int q) {
        auto rows_of = [&](int b) { return max(0, min(rows_per_blk, M - b * rows_per_blk)); };
        double s[4] = {0.0, 0.0, 0.0, 0.0};
        for (int b = 0; b < nblk; ++b) {
          const f32x4 mb = *reinterpret_cast<const f32x4*>(partial + ((size_t)b * 2 + 0) * Cs + (size_t)q * 4);
          const double nb = rows_of(b);
#pragma unroll
          for (int e = 0; e < 4; ++e) s[e] += nb * (double)mb[e];
        }
        double m2[4] = {0.0, 0.0, 0.0, 0.0};
        for (int b = 0; b < nblk; ++b) {
          const double nb = rows_of(b);
          if (nb > 0.0) {
            const f32x4 mb = *reinterpret_cast<const f32x4*>(partial + ((size_t)b * 2 + 0) * Cs + (size_t)q * 4);
            const f32x4 vb = *reinterpret_cast<const f32x4*>(partial + ((size_t)b * 2 + 1) * Cs + (size_t)q * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const double d = (double)mb[e] - s[e] / M;
              m2[e] += (double)vb[e] + nb * d * d;
            }
          }
        }
        // the first row lane of workgroup 0 owns the channel's published statistics
        const bool owner = blockIdx.x == 0 && (int)threadIdx.x == (q % RED_THREADS);  // row lane 0 of workgroup 0
        BnCoef k;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int c = q * 4 + e;
          float sc = 0.f, sh = 0.f, v = 0.f, mean_f = 0.f, istd_f = 0.f;
          if (c < C) {
            const double mean = s[e] / M;
            double var = m2[e] / M;
            if (var < 0.0) var = 0.0;
            mean_f = (float)mean;
            istd_f = (float)(1.0 / sqrt(var + (double)eps));
            v = 1.f;
            sc = (gamma ? gamma[c] : 1.f) * istd_f;
            sh = (beta ? beta[c] : 0.f) - mean_f * sc;
            if (owner && running_mean != nullptr) {
              const double unb = M > 1 ? var * ((double)M / (double)(M - 1)) : var;
              running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
              running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unb);
            }
          }
          if (owner) {
            save_mean[c] = mean_f;
            save_invstd[c] = istd_f;
          }
          k.sc[e] = sc;
          k.sh[e] = sh;
          k.valid[e] = v;
        }
        return k;
      },
      [&](int r, int q, const BnCoef& k) {
        const size_t off = (size_t)r * Cs + (size_t)q * 4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + off);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = k.valid[e] != 0.f ? act_fwd(v[e] * k.sc[e] + k.sh[e], ACT) : 0.f;
        if (mul != nullptr) o *= *reinterpret_cast<const f32x4*>(mul + off);
        if (res != nullptr) o += *reinterpret_cast<const f32x4*>(res + off);
        *reinterpret_cast<f32x4*>(y + off) = o;
      });
}

#define VMTL_ACT_SWITCH(act, CALL)                   \
  switch (act) {                                     \
    case VMTL_ACT_NONE: CALL(VMTL_ACT_NONE); break;   \
    case VMTL_ACT_RELU: CALL(VMTL_ACT_RELU); break;   \
    case VMTL_ACT_HSWISH: CALL(VMTL_ACT_HSWISH); break;     \
    case VMTL_ACT_HSIGMOID: CALL(VMTL_ACT_HSIGMOID); break; \
    case VMTL_ACT_SIGMOID: CALL(VMTL_ACT_SIGMOID); break;   \
    default: return VMTL_ERR_ARG;                    \
  }

extern "C" int vmtl_bn_apply(const float* x, const float* mean, const float* invstd, const float* gamma,
                             const float* beta, const float* mul, const float* res, float* y, long long M, int C,
                             int Cs, int act, void* stream) {
  VMTL_ENTER();
  if (!x || !y || M <= 0 || M > 0x7fffffffLL || C <= 0 || C > Cs || (Cs & 3)) return VMTL_ERR_ARG;
  if ((mean == nullptr) != (invstd == nullptr)) return VMTL_ERR_ARG;
  const int nb = sweep_blocks(M, Cs);
#define CALL(A)                                                                                                   \
  hipLaunchKernelGGL((bn_apply_kernel<A>), dim3(nb), dim3(RED_THREADS), 0, (hipStream_t)stream, x, mean, invstd, \
                     gamma, beta, mul, res, y, (int)M, C, Cs)
  VMTL_ACT_SWITCH(act, CALL)
#undef CALL
  return vmtl_check_launch();
}

extern "C" int vmtl_bn_fuse_max_rows() { return VMTL_BN_FUSE_MAX_ROWS; }

extern "C" int vmtl_bn_apply_fused(const float* x, const float* partial, int nblk, int rows_per_blk, float eps,
                                   float momentum, float* running_mean, float* running_var,
                                   long long* num_batches_tracked, float* save_mean, float* save_invstd,
                                   const float* gamma, const float* beta, const float* mul, const float* res, float* y,
                                   long long M, int C, int Cs, int act, void* stream) {
  VMTL_ENTER();
  if (!x || !y || !partial || !save_mean || !save_invstd || M <= 0 || M > 0x7fffffffLL || C <= 0 || C > Cs || (Cs & 3))
    return VMTL_ERR_ARG;
  if (nblk <= 0 || nblk > VMTL_BN_FUSE_MAX_ROWS || rows_per_blk <= 0 || (long long)nblk * rows_per_blk < M)
    return VMTL_ERR_ARG;
  const int nb = sweep_blocks(M, Cs);
#define CALL(A)                                                                                                    \
  hipLaunchKernelGGL((bn_apply_fused_kernel<A>), dim3(nb), dim3(RED_THREADS), 0, (hipStream_t)stream, x, partial, \
                     nblk, rows_per_blk, eps, momentum, running_mean, running_var, num_batches_tracked, save_mean, \
                     save_invstd, gamma, beta, mul, res, y, (int)M, C, Cs)
  VMTL_ACT_SWITCH(act, CALL)
#undef CALL
  return vmtl_check_launch();
}

// ---------------------------------------------------------------- backward
struct BnBwdCoef {
  f32x4 mean, invstd, gamma, beta, valid;
  f32x4 c1, c2;  // sum_dz / M, sum_dzx / M (stage 2 only)
};

__device__ __forceinline__ BnBwdCoef bn_bwd_coef(int q, int C, const float* mean, const float* invstd,
                                                 const float* gamma, const float* beta, const float* sum_dz,
                                                 const float* sum_dzx, float invM) {
  BnBwdCoef k;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int c = q * 4 + e;
    const bool ok = c < C;
    k.valid[e] = ok ? 1.f : 0.f;
    k.mean[e] = (ok && mean) ? mean[c] : 0.f;
    k.invstd[e] = (ok && mean) ? invstd[c] : 1.f;
    k.gamma[e] = (ok && gamma) ? gamma[c] : 1.f;
    k.beta[e] = (ok && beta) ? beta[c] : 0.f;
    k.c1[e] = (ok && sum_dz) ? sum_dz[c] * invM : 0.f;
    k.c2[e] = (ok && sum_dzx) ? sum_dzx[c] * invM : 0.f;
  }
  return k;
}

// stage 1: partial sums of dz and dz*xhat per channel (dz = dL/d(pre-activation));
// also emits dmul = dy * act(z) when a gate operand was used.
template <int ACT>
__global__ __launch_bounds__(RED_THREADS) void bn_bwd_reduce_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ mean,
    const float* __restrict__ invstd, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ mul, float* __restrict__ dmul, int M, int C, int Cs, float* partial) {
  const int CQ = Cs >> 2;
  struct Row { f32x4 xv, g, mv; };
  column_reduce_init2<2>(
      M, CQ, Cs, partial, [&](int q) { return bn_bwd_coef(q, C, mean, invstd, gamma, beta, nullptr, nullptr, 0.f); },
      [&](int r, int q) {
        const size_t off = (size_t)r * Cs + (size_t)q * 4;
        Row l;
        l.xv = *reinterpret_cast<const f32x4*>(x + off);
        l.g = *reinterpret_cast<const f32x4*>(dy + off);
        l.mv = (f32x4){1.f, 1.f, 1.f, 1.f};
        if (mul != nullptr) l.mv = *reinterpret_cast<const f32x4*>(mul + off);
        return l;
      },
      [&](const Row& l, int r, int q, const BnBwdCoef& k, f32x4* acc) {
        const f32x4 xh = (l.xv - k.mean) * k.invstd;
        const f32x4 z = k.gamma * xh + k.beta;
        f32x4 dm, dz;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          dm[e] = k.valid[e] * l.g[e] * act_fwd(z[e], ACT);
          dz[e] = k.valid[e] * l.g[e] * l.mv[e] * act_grad(z[e], ACT);
        }
        acc[0] += dz;
        acc[1] += dz * xh;
        if (dmul != nullptr) *reinterpret_cast<f32x4*>(dmul + (size_t)r * Cs + (size_t)q * 4) = dm;
      });
}

// sums the partial rows: dbeta[c] = sum dz, dgamma[c] = sum dz*xhat  (fp64, fixed order)
// Optionally (coef_a != null, grid = Cs workgroups) also emits the backward-apply as an affine map of (dz, x):
//   dx = gamma*invstd*(dz - sum_dz/M - xhat*sum_dzx/M) = coef_a*dz + coef_b*x + coef_c   (train)
//   dx = gamma*invstd*dz                                                                  (eval)
// for a consumer that applies it while loading (vmtl_conv3x3_small prologue); zeros on the pad channels.
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ partial, int nblk, int C,
                                                              int Cs, float* sum_dz, float* sum_dzx,
                                                              const float* __restrict__ mean,
                                                              const float* __restrict__ invstd,
                                                              const float* __restrict__ gamma, float invM, int training,
                                                              float* coef_a, float* coef_b, float* coef_c) {
  // one workgroup per channel quad, one pass, one barrier pair (see bn_finalize_kernel); sum outputs: exactly C
  // entries (they may be slots of a flat gradient arena)
  __shared__ double sh[4][8];
  const int c0 = blockIdx.x * 4, t = threadIdx.x;
  const int c = c0 + t;
  const bool fin = t < 4 && c < C;
  float g = 1.f, is = 0.f, mu = 0.f;
  if (fin && coef_a != nullptr) {
    if (gamma != nullptr) g = gamma[c];
    is = invstd[c];
    mu = mean[c];
  }
  double s[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
  for (int b = t; b < nblk; b += 256) {
    const float* row = partial + (size_t)b * 2 * Cs + c0;
    const f32x4 a4 = *reinterpret_cast<const f32x4*>(row);
    const f32x4 b4 = *reinterpret_cast<const f32x4*>(row + Cs);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      s[e] += (double)a4[e];
      s[4 + e] += (double)b4[e];
    }
  }
  block_sum_n<8>(s, sh);
  if (t >= 4 || c >= Cs) return;
  if (!fin) {
    if (coef_a != nullptr) coef_a[c] = coef_b[c] = coef_c[c] = 0.f;
    return;
  }
  const double s1 = s[t], s2 = s[4 + t];
  sum_dz[c] = (float)s1;
  sum_dzx[c] = (float)s2;
  if (coef_a != nullptr) {
    const float gi = g * is;
    const float c1 = training ? (float)s1 * invM : 0.f, c2 = training ? (float)s2 * invM : 0.f;
    coef_a[c] = gi;
    coef_b[c] = -gi * is * c2;
    coef_c[c] = gi * (mu * is * c2 - c1);
  }
}

// stage 2: dx.  train: gamma*invstd*(dz - sum_dz/M - xhat*sum_dzx/M); eval (or no BN): scale*dz.
template <int ACT>
__global__ __launch_bounds__(RED_THREADS) void bn_bwd_apply_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ mean,
    const float* __restrict__ invstd, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ mul, const float* __restrict__ sum_dz, const float* __restrict__ sum_dzx,
    float* __restrict__ dx, int M, int C, int Cs, int training) {
  const float invM = 1.f / (float)M;
  const bool use_sums = training && mean != nullptr;
  struct Row { f32x4 xv, g, mv; };
  column_sweep2(
      M, Cs >> 2,
      [&](int q) {
        return bn_bwd_coef(q, C, mean, invstd, gamma, beta, use_sums ? sum_dz : nullptr, use_sums ? sum_dzx : nullptr,
                           invM);
      },
      [&](int r, int q) {
        const size_t off = (size_t)r * Cs + (size_t)q * 4;
        Row l;
        l.xv = *reinterpret_cast<const f32x4*>(x + off);
        l.g = *reinterpret_cast<const f32x4*>(dy + off);
        l.mv = (f32x4){1.f, 1.f, 1.f, 1.f};
        if (mul != nullptr) l.mv = *reinterpret_cast<const f32x4*>(mul + off);
        return l;
      },
      [&](const Row& l, int r, int q, const BnBwdCoef& k) {
        const f32x4 xh = (l.xv - k.mean) * k.invstd;
        const f32x4 z = k.gamma * xh + k.beta;
        f32x4 dz;
#pragma unroll
        for (int e = 0; e < 4; ++e) dz[e] = k.valid[e] * l.g[e] * l.mv[e] * act_grad(z[e], ACT);
        // without BatchNorm (mean == nullptr) gamma = invstd = 1 and c1 = c2 = 0: dx = dz
        *reinterpret_cast<f32x4*>(dx + (size_t)r * Cs + (size_t)q * 4) = k.gamma * k.invstd * (dz - k.c1 - xh * k.c2);
      });
}

// One call = reduce + finalize + apply.  `partial` needs vmtl_reduce_rows(M)*2*Cs floats.
// sum_dz / sum_dzx ([C]) are the bias / weight gradients of the BatchNorm.
extern "C" int vmtl_bn_bwd(const float* x, const float* dy, const float* mean, const float* invstd,
                           const float* gamma, const float* beta, const float* mul, float* dmul, float* partial,
                           float* sum_dz, float* sum_dzx, float* dx, int M, int C, int Cs, int act, int training,
                           void* stream) {
  VMTL_ENTER();
  if (!x || !dy || !dx || M <= 0 || C <= 0 || C > Cs || (Cs & 3)) return VMTL_ERR_ARG;
  if ((mean == nullptr) != (invstd == nullptr)) return VMTL_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const bool need_sums = mean != nullptr;  // a bare activation has no per-channel sums
  if (need_sums || dmul != nullptr) {
    if (!partial || (need_sums && (!sum_dz || !sum_dzx))) return VMTL_ERR_ARG;
    const int nblk = red_blocks(M);
#define CALL(A)                                                                                                 \
  hipLaunchKernelGGL((bn_bwd_reduce_kernel<A>), dim3(nblk), dim3(RED_THREADS), 0, st, x, dy, mean, invstd, gamma, \
                     beta, mul, dmul, M, C, Cs, partial)
    VMTL_ACT_SWITCH(act, CALL)
#undef CALL
    if (need_sums)
      hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 3) >> 2), dim3(256), 0, st, partial, nblk, C, Cs, sum_dz, sum_dzx,
                         nullptr, nullptr, nullptr, 0.f, 0, nullptr, nullptr, nullptr);
  }
  const int nb = sweep_blocks(M, Cs);
#define CALL(A)                                                                                              \
  hipLaunchKernelGGL((bn_bwd_apply_kernel<A>), dim3(nb), dim3(RED_THREADS), 0, st, x, dy, mean, invstd, gamma, \
                     beta, mul, sum_dz, sum_dzx, dx, M, C, Cs, training)
  VMTL_ACT_SWITCH(act, CALL)
#undef CALL
  return vmtl_check_launch();
}

// The two halves of vmtl_bn_bwd for producers that already emitted dz = dy * act'(z) and its per-block column sums
// from their own epilogue (vmtl_conv3x3_small ep_mode 2, vmtl_conv2d_fwd with a BatchNorm-backward epilogue):
// finalize sums the [nblk][2][Cs] rows (sum dz, sum dz*xhat) into the BatchNorm parameter gradients (and the
// affine coefficients, see the kernel); apply turns (x, dz) into dx.
extern "C" int vmtl_bn_bwd_finalize(const float* partial, int nblk, int M, int C, int Cs, float* sum_dz, float* sum_dzx,
                                    const float* mean, const float* invstd, const float* gamma, int training,
                                    float* coef_a, float* coef_b, float* coef_c, void* stream) {
  VMTL_ENTER();
  if (!partial || nblk <= 0 || M <= 0 || C <= 0 || C > Cs || (Cs & 3) || !sum_dz || !sum_dzx) return VMTL_ERR_ARG;
  const bool want = coef_a != nullptr;
  if (want && (!coef_b || !coef_c || !mean || !invstd)) return VMTL_ERR_ARG;
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(want ? Cs >> 2 : (C + 3) >> 2), dim3(256), 0, (hipStream_t)stream, partial,
                     nblk, C, Cs,
                     sum_dz, sum_dzx, mean, invstd, gamma, 1.f / (float)M, training, coef_a, coef_b, coef_c);
  return vmtl_check_launch();
}

extern "C" int vmtl_bn_bwd_apply(const float* x, const float* dz, const float* mean, const float* invstd,
                                 const float* gamma, const float* sum_dz, const float* sum_dzx, float* dx, int M, int C,
                                 int Cs, int training, void* stream) {
  VMTL_ENTER();
  if (!x || !dz || !dx || !mean || !invstd || M <= 0 || C <= 0 || C > Cs || (Cs & 3)) return VMTL_ERR_ARG;
  if (training && (!sum_dz || !sum_dzx)) return VMTL_ERR_ARG;
  const int nb = sweep_blocks(M, Cs);
  hipLaunchKernelGGL((bn_bwd_apply_kernel<VMTL_ACT_NONE>), dim3(nb), dim3(RED_THREADS), 0, (hipStream_t)stream, x, dz, mean,
                     invstd, gamma, nullptr, nullptr, sum_dz, sum_dzx, dx, M, C, Cs, training);
  return vmtl_check_launch();
}

// ---------------------------------------------------------------- BatchNorm + activation + 2x2 max-pool in one pass
// MTAN's encoder attention modules end in conv3 -> BatchNorm -> ReLU -> MaxPool2d(2) (reference models/mtan_model.py:67-83) and
// nobody else reads the full-resolution activation.  Forward: one sweep of x writes only the pooled map (the separate passes
// wrote and re-read the full-resolution activation).  Backward: the pooled gradient is scattered to the window's arg-max ON THE
// FLY (first maximum in window order, NaN wins - the rule of maxpool2_bwd_kernel) inside the BatchNorm reduce and apply
// sweeps; the full-resolution gradient of the pool never exists.  A "row" of the sweep skeletons is one pooled pixel.
struct PoolRow {
  f32x4 v[4], g;
  size_t o00;
};

__device__ __forceinline__ size_t pool_window(int r, int q, int H, int W, int Cs) {
  const int Wo = W >> 1, Ho = H >> 1;
  const int wo = r % Wo, t = r / Wo;
  const int ho = t % Ho, b = t / Ho;
  return ((size_t)(b * H + 2 * ho) * W + 2 * wo) * Cs + (size_t)q * 4;
}

__device__ __forceinline__ void pool_load(PoolRow& l, const float* x, int r, int q, int H, int W, int Cs) {
  l.o00 = pool_window(r, q, H, W, Cs);
  l.v[0] = *reinterpret_cast<const f32x4*>(x + l.o00);
  l.v[1] = *reinterpret_cast<const f32x4*>(x + l.o00 + Cs);
  l.v[2] = *reinterpret_cast<const f32x4*>(x + l.o00 + (size_t)W * Cs);
  l.v[3] = *reinterpret_cast<const f32x4*>(x + l.o00 + (size_t)W * Cs + Cs);
}

// activation values, normalised inputs and arg-max of one window element e of the quad
template <int ACT>
__device__ __forceinline__ int pool_argmax(const PoolRow& l, const BnBwdCoef& k, int e, float* xh, float* z, float& amax) {
  int am = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    xh[w] = (l.v[w][e] - k.mean[e]) * k.invstd[e];
    z[w] = k.gamma[e] * xh[w] + k.beta[e];
    const float a = act_fwd(z[w], ACT);
    if (w == 0) amax = a;
    else if (a > amax || a != a) { amax = a; am = w; }
  }
  return am;
}

template <int ACT>
__global__ __launch_bounds__(RED_THREADS) void bn_act_pool2_fwd_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                                       const float* __restrict__ invstd,
                                                                       const float* __restrict__ gamma,
                                                                       const float* __restrict__ beta, float* __restrict__ y,
                                                                       int Mp, int H, int W, int C, int Cs) {
  column_sweep2(
      Mp, Cs >> 2, [&](int q) { return bn_bwd_coef(q, C, mean, invstd, gamma, beta, nullptr, nullptr, 0.f); },
      [&](int r, int q) {
        PoolRow l;
        pool_load(l, x, r, q, H, W, Cs);
        return l;
      },
      [&](const PoolRow& l, int r, int q, const BnBwdCoef& k) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float xh[4], z[4], amax;
          pool_argmax<ACT>(l, k, e, xh, z, amax);
          o[e] = k.valid[e] != 0.f ? amax : 0.f;
        }
        *reinterpret_cast<f32x4*>(y + (size_t)r * Cs + (size_t)q * 4) = o;
      });
}

template <int ACT>
__global__ __launch_bounds__(RED_THREADS) void bn_pool2_bwd_reduce_kernel(
    const float* __restrict__ x, const float* __restrict__ dyp, const float* __restrict__ mean,
    const float* __restrict__ invstd, const float* __restrict__ gamma, const float* __restrict__ beta, int Mp, int H, int W,
    int C, int Cs, float* partial) {
  column_reduce_init2<2>(
      Mp, Cs >> 2, Cs, partial, [&](int q) { return bn_bwd_coef(q, C, mean, invstd, gamma, beta, nullptr, nullptr, 0.f); },
      [&](int r, int q) {
        PoolRow l;
        pool_load(l, x, r, q, H, W, Cs);
        l.g = *reinterpret_cast<const f32x4*>(dyp + (size_t)r * Cs + (size_t)q * 4);
        return l;
      },
      [&](const PoolRow& l, int r, int q, const BnBwdCoef& k, f32x4* acc) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float xh[4], z[4], amax;
          const int am = pool_argmax<ACT>(l, k, e, xh, z, amax);
          const float dz = k.valid[e] * l.g[e] * act_grad(z[am], ACT);
          acc[0][e] += dz;
          acc[1][e] += dz * xh[am];
        }
      });
}

template <int ACT>
__global__ __launch_bounds__(RED_THREADS) void bn_pool2_bwd_apply_kernel(
    const float* __restrict__ x, const float* __restrict__ dyp, const float* __restrict__ mean,
    const float* __restrict__ invstd, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ sum_dz, const float* __restrict__ sum_dzx, float* __restrict__ dx, int Mp, int H, int W, int C,
    int Cs, int training) {
  const float invM = 1.f / (4.f * (float)Mp);  // the BatchNorm saw all four pixels of every window
  const bool use_sums = training != 0;
  column_sweep2(
      Mp, Cs >> 2,
      [&](int q) {
        return bn_bwd_coef(q, C, mean, invstd, gamma, beta, use_sums ? sum_dz : nullptr, use_sums ? sum_dzx : nullptr, invM);
      },
      [&](int r, int q) {
        PoolRow l;
        pool_load(l, x, r, q, H, W, Cs);
        l.g = *reinterpret_cast<const f32x4*>(dyp + (size_t)r * Cs + (size_t)q * 4);
        return l;
      },
      [&](const PoolRow& l, int r, int q, const BnBwdCoef& k) {
        f32x4 d[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float xh[4], z[4], amax;
          const int am = pool_argmax<ACT>(l, k, e, xh, z, amax);
          const float dz = k.valid[e] * l.g[e] * act_grad(z[am], ACT);
          const float gi = k.gamma[e] * k.invstd[e];
#pragma unroll
          for (int w = 0; w < 4; ++w) d[w][e] = k.valid[e] * gi * ((w == am ? dz : 0.f) - k.c1[e] - xh[w] * k.c2[e]);
        }
        *reinterpret_cast<f32x4*>(dx + l.o00) = d[0];
        *reinterpret_cast<f32x4*>(dx + l.o00 + Cs) = d[1];
        *reinterpret_cast<f32x4*>(dx + l.o00 + (size_t)W * Cs) = d[2];
        *reinterpret_cast<f32x4*>(dx + l.o00 + (size_t)W * Cs + Cs) = d[3];
      });
}

// y [B][H/2][W/2][Cs] = maxpool2(act(BN(x))), x [B][H][W][Cs] (H, W even); mean / invstd from vmtl_bn_stats
extern "C" int vmtl_bn_act_pool2_fwd(const float* x, const float* mean, const float* invstd, const float* gamma,
                                     const float* beta, float* y, int B, int H, int W, int C, int Cs, int act, void* stream) {
  VMTL_ENTER();
  if (!x || !y || !mean || !invstd || B <= 0 || H < 2 || W < 2 || (H & 1) || (W & 1) || C <= 0 || C > Cs || (Cs & 3))
    return VMTL_ERR_ARG;
  if ((long long)B * H * W > 0x7fffffffLL) return VMTL_ERR_ARG;
  const int Mp = B * (H >> 1) * (W >> 1);
  const int nb = sweep_blocks(Mp, Cs);
  hipStream_t st = (hipStream_t)stream;
#define CALL(A)                                                                                                        \
  hipLaunchKernelGGL((bn_act_pool2_fwd_kernel<A>), dim3(nb), dim3(RED_THREADS), 0, st, x, mean, invstd, gamma, beta, y, Mp, H, \
                     W, C, Cs)
  VMTL_ACT_SWITCH(act, CALL)
#undef CALL
  return vmtl_check_launch();
}

// backward of the above: dyp [B][H/2][W/2][Cs] -> dx [B][H][W][Cs], sum_dz / sum_dzx [C] = the BatchNorm's bias / weight
// gradients; partial: vmtl_reduce_rows(B*(H/2)*(W/2)) * 2 * Cs floats
extern "C" int vmtl_bn_act_pool2_bwd(const float* x, const float* dyp, const float* mean, const float* invstd,
                                     const float* gamma, const float* beta, float* partial, float* sum_dz, float* sum_dzx,
                                     float* dx, int B, int H, int W, int C, int Cs, int act, int training, void* stream) {
  VMTL_ENTER();
  if (!x || !dyp || !dx || !mean || !invstd || !partial || !sum_dz || !sum_dzx || B <= 0 || H < 2 || W < 2 || (H & 1) ||
      (W & 1) || C <= 0 || C > Cs || (Cs & 3))
    return VMTL_ERR_ARG;
  if ((long long)B * H * W > 0x7fffffffLL) return VMTL_ERR_ARG;
  const int Mp = B * (H >> 1) * (W >> 1);
  hipStream_t st = (hipStream_t)stream;
  const int nblk = red_blocks(Mp);
#define CALL(A)                                                                                                            \
  hipLaunchKernelGGL((bn_pool2_bwd_reduce_kernel<A>), dim3(nblk), dim3(RED_THREADS), 0, st, x, dyp, mean, invstd, gamma, beta, \
                     Mp, H, W, C, Cs, partial)
  VMTL_ACT_SWITCH(act, CALL)
#undef CALL
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 3) >> 2), dim3(256), 0, st, partial, nblk, C, Cs, sum_dz, sum_dzx,
                     nullptr, nullptr, nullptr, 0.f, 0, nullptr, nullptr, nullptr);
  const int nb = sweep_blocks(Mp, Cs);
#define CALL(A)                                                                                                           \
  hipLaunchKernelGGL((bn_pool2_bwd_apply_kernel<A>), dim3(nb), dim3(RED_THREADS), 0, st, x, dyp, mean, invstd, gamma, beta, \
                     sum_dz, sum_dzx, dx, Mp, H, W, C, Cs, training)
  VMTL_ACT_SWITCH(act, CALL)
#undef CALL
  return vmtl_check_launch();
}

// ---------------------------------------------------------------- generic column sums
// out[k][c] = sum_m f_k(a[m][c], b[m][c]);  mode 0: (a) -> bias gradient, mode 1: (a*b) -> stitch/scale gradient
__global__ __launch_bounds__(RED_THREADS) void colsum_kernel(const float* __restrict__ a,
                                                             const float* __restrict__ b, int M, int Cs, int mode,
                                                             float* partial) {
  const int CQ = Cs >> 2;
  column_reduce<1>(M, CQ, Cs, partial, [&](int r, int q, f32x4* acc) {
    const size_t off = (size_t)r * Cs + (size_t)q * 4;
    f32x4 v = *reinterpret_cast<const f32x4*>(a + off);
    if (mode == 1) v *= *reinterpret_cast<const f32x4*>(b + off);
    acc[0] += v;
  });
}

__global__ __launch_bounds__(256) void colsum_finalize_kernel(const float* __restrict__ partial, int nblk, int C,
                                                              int Cs, float* out) {
  __shared__ double sh[4];
  const int c = blockIdx.x;
  const double s = block_rows_sum(partial, nblk, 1, 0, Cs, c, sh);
  if (threadIdx.x == 0) out[c] = (float)s;
}

// a single scalar: the sum over channels too (layer-wise stitch weight gradient).  The per-channel sums come from the
// C-workgroup finalize above (round 2 had ONE workgroup walk all C columns serially: up to 960 dependent block
// reductions on the critical path - csnet ran 1.6 ms/step slower layer-wise than channel-wise)
__global__ __launch_bounds__(256) void sum_channels_kernel(const float* __restrict__ v, int C, float* out) {
  __shared__ double sh[4];
  double a = 0.0;
  for (int c = threadIdx.x; c < C; c += 256) a += (double)v[c];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (float)(sh[0] + sh[1] + sh[2] + sh[3]);
}

// out[c] = column sums of the partial rows; reduce_all: out[0] = their sum over channels, through the scratch row
// partial[nblk][*] (the caller's partial buffer holds vmtl_reduce_rows(M) + 1 rows)
static void colsum_finish(float* partial, int nblk, int C, int Cs, int reduce_all, float* out, hipStream_t st) {
  float* per_channel = reduce_all ? partial + (size_t)nblk * Cs : out;
  hipLaunchKernelGGL(colsum_finalize_kernel, dim3(C), dim3(256), 0, st, partial, nblk, C, Cs, per_channel);
  if (reduce_all) hipLaunchKernelGGL(sum_channels_kernel, dim3(1), dim3(256), 0, st, per_channel, C, out);
}

// cross-stitch backward in one sweep (reference models/cross_stitch_model.py:21-37): dx = w * dy and the partial
// column sums of x * dy (the stitch weight's gradient) - dy is read once instead of by a scale pass and a reduction
__global__ __launch_bounds__(RED_THREADS) void stitch_bwd_kernel(const float* __restrict__ x,
                                                                 const float* __restrict__ dy,
                                                                 const float* __restrict__ w, float* __restrict__ dx,
                                                                 int M, int C, int Cs, int wstride, float* partial) {
  const int CQ = Cs >> 2;
  column_reduce<1>(M, CQ, Cs, partial, [&](int r, int q, f32x4* acc) {
    const size_t off = (size_t)r * Cs + (size_t)q * 4;
    const f32x4 g = *reinterpret_cast<const f32x4*>(dy + off);
    acc[0] += g * *reinterpret_cast<const f32x4*>(x + off);
    if (dx != nullptr) {
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = q * 4 + e;
        v[e] = c < C ? g[e] * w[(size_t)c * wstride] : 0.f;
      }
      *reinterpret_cast<f32x4*>(dx + off) = v;
    }
  });
}

// dx (nullable) = w[c * wstride] * dy;  dw[c] = sum_m x*dy  (reduce_all: one scalar, the layer-wise stitch weight).
// partial: (vmtl_reduce_rows(M) + 1) * Cs floats (the last row is scratch for reduce_all).
extern "C" int vmtl_stitch_bwd(const float* x, const float* dy, const float* w, float* dx, float* partial, float* dw,
                               int M, int C, int Cs, int wstride, int reduce_all, void* stream) {
  VMTL_ENTER();
  if (!x || !dy || !w || !partial || !dw || M <= 0 || C <= 0 || C > Cs || (Cs & 3)) return VMTL_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int nblk = red_blocks(M);
  hipLaunchKernelGGL(stitch_bwd_kernel, dim3(nblk), dim3(RED_THREADS), 0, st, x, dy, w, dx, M, C, Cs, wstride, partial);
  colsum_finish(partial, nblk, C, Cs, reduce_all, dw, st);
  return vmtl_check_launch();
}

extern "C" int vmtl_colsum(const float* a, const float* b, int M, int C, int Cs, int mode, int reduce_all,
                           float* partial, float* out, void* stream) {
  VMTL_ENTER();
  if (!a || !partial || !out || M <= 0 || C <= 0 || C > Cs || (Cs & 3) || (mode == 1 && !b)) return VMTL_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int nblk = red_blocks(M);
  hipLaunchKernelGGL(colsum_kernel, dim3(nblk), dim3(RED_THREADS), 0, st, a, b, M, Cs, mode, partial);
  colsum_finish(partial, nblk, C, Cs, reduce_all, out, st);
  return vmtl_check_launch();
}
