// Per-pixel losses of the multi-task step, forward and backward.
//
//   cross entropy : torch.nn.CrossEntropyLoss() as used at reference lit_module.py:31,123
//                   (mean over B*H*W, no ignore_index / class weights / smoothing)
//   SILog         : reference vision_mtl/losses.py:14-36 on already-sigmoided predictions
//                   (mask = target > min_depth, unbiased variance, 10*sqrt(var + 0.15*mean^2))
//   L1 / MAE      : the depth metric of reference lit_module.py:68,112 (mean |p - t|)
//
// Reductions are two-stage: per-workgroup partials in fp64, summed in a fixed order by a
// single finalize workgroup, so results are run-to-run reproducible.
#include "common.h"

#define CE_THREADS 256

__device__ __forceinline__ double block_sum_d(double v, double* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = 0.0;
  const int nw = blockDim.x >> 6;
  for (int i = 0; i < nw; ++i) s += sh[i];
  return s;
}

// Element (b, c, hw) of the logits sits at z[b*sb + c*sc + hw*sp]:
//   NCHW (the reference's layout at the boundary): sb = C*HW, sc = HW, sp = 1  -> every channel
//   read of a wave is 256 contiguous bytes;  NHWC: sb = HW*ld, sc = 1, sp = ld.
// One thread per pixel; the (few) channels are reduced in registers.
// Numerics: log-softmax is evaluated as (z - max) - log(sum exp(z - max)), subtracting the max FIRST
// (exact for nearby floats).  Forming lse = max + log(sum) and then z - lse would lose ~ulp(max)
// per pixel when the logits are large, a systematic error the backward pass of a deep net amplifies.
__global__ __launch_bounds__(CE_THREADS) void ce_fwd_kernel(const float* __restrict__ z,
                                                            const long long* __restrict__ tgt,
                                                            double* __restrict__ partial,
                                                            long long* __restrict__ amax, long long P, int HW,
                                                            int C, long long sb, long long sc, long long sp) {
  __shared__ double shd[4];
  double loss = 0.0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (long long)gridDim.x * blockDim.x) {
    const long long b = i / HW, hw = i - b * HW;
    const float* r = z + b * sb + hw * sp;
    float m = r[0];
    int am = 0;
    for (int c = 1; c < C; ++c) {
      const float v = r[c * sc];
      if (v > m) { m = v; am = c; }  // first maximum wins, like torch.argmax on ties
    }
    if (amax != nullptr) amax[i] = am;
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += expf(r[c * sc] - m);
    const long long t = tgt[i];
    // torch raises on an out-of-range class index; raising here would need a host sync on the step path, so the
    // pixel contributes NaN: the loss (and, in ce_bwd, the gradient) fails the step VISIBLY instead of silently
    // adding 0 to a loss still divided by P
    if (t < 0 || t >= C) loss += (double)__int_as_float(0x7fc00000);
    else loss += (double)(logf(s) - (r[t * sc] - m));
  }
  const double bs = block_sum_d(loss, shd);
  if (threadIdx.x == 0) partial[blockIdx.x] = bs;
}

// The same pass with the pixel's logits held in registers (C <= 32: one load per logit instead of three sweeps over
// the channel planes for max / sum-exp / target): 44 -> ~30 us for 32x19x128x256.
template <int Q>
__global__ __launch_bounds__(CE_THREADS) void ce_fwd_regs_kernel(const float* __restrict__ z,
                                                                 const long long* __restrict__ tgt,
                                                                 double* __restrict__ partial,
                                                                 long long* __restrict__ amax, long long P, int HW,
                                                                 int C, long long sb, long long sc, long long sp) {
  __shared__ double shd[4];
  double loss = 0.0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (long long)gridDim.x * blockDim.x) {
    const long long b = i / HW, hw = i - b * HW;
    const float* r = z + b * sb + hw * sp;
    float v[4 * Q];
#pragma unroll
    for (int c = 0; c < 4 * Q; ++c) v[c] = c < C ? r[c * sc] : 0.f;
    const long long t = tgt[i];
    float m = v[0];
    int am = 0;
#pragma unroll
    for (int c = 1; c < 4 * Q; ++c)
      if (c < C && v[c] > m) { m = v[c]; am = c; }  // first maximum wins, like torch.argmax on ties
    if (amax != nullptr) amax[i] = am;
    float s = 0.f, zt = 0.f;
#pragma unroll
    for (int c = 0; c < 4 * Q; ++c) {
      if (c < C) s += expf(v[c] - m);
      if (c == t) zt = v[c];
    }
    if (t < 0 || t >= C) loss += (double)__int_as_float(0x7fc00000);  // see ce_fwd_kernel
    else loss += (double)(logf(s) - (zt - m));
  }
  const double bs = block_sum_d(loss, shd);
  if (threadIdx.x == 0) partial[blockIdx.x] = bs;
}

// err (may be null): the result becomes NaN when the flag is set.
__global__ void sum_finalize_kernel(const double* __restrict__ partial, int n, double scale, float* out,
                                    const int* __restrict__ err) {
  __shared__ double shd[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += partial[i];
  const double t = block_sum_d(s, shd);
  if (threadIdx.x == 0) out[0] = (err != nullptr && err[0] != 0) ? __int_as_float(0x7fc00000) : (float)(t * scale);
}

// dz = (softmax(z) - onehot(y)) * gout / P, written with the same strides as z.  The softmax is
// recomputed from the logits (max-subtracted, see above); nothing is saved by the forward pass.
__global__ __launch_bounds__(CE_THREADS) void ce_bwd_kernel(const float* __restrict__ z,
                                                            const long long* __restrict__ tgt,
                                                            const float* __restrict__ gout, float* __restrict__ dz,
                                                            long long P, int HW, int C, long long sb, long long sc,
                                                            long long sp, long long dsb, long long dsc, long long dsp) {
  const float g = gout[0] / (float)P;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (long long)gridDim.x * blockDim.x) {
    const long long b = i / HW, hw = i - b * HW;
    const long long off = b * sb + hw * sp, doff = b * dsb + hw * dsp;
    float m = z[off];
    for (int c = 1; c < C; ++c) m = fmaxf(m, z[off + c * sc]);
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += expf(z[off + c * sc] - m);
    const float inv = 1.f / s;
    const long long t = tgt[i];
    const float bad = (t < 0 || t >= C) ? __int_as_float(0x7fc00000) : 0.f;  // as the forward: NaN, never a silent 0
    for (int c = 0; c < C; ++c) dz[doff + c * dsc] = (expf(z[off + c * sc] - m) * inv - (c == t ? 1.f : 0.f)) * g + bad;
  }
}

// vmtl_ce_bwd_strided with a channel-contiguous gradient (NHWC rows of ld >= ceil4(C) floats): one thread per
// (pixel, channel quad), so a wave writes 1 KB of CONTIGUOUS gradient (one thread per pixel left every store
// instruction touching 64 lanes x 16 bytes at an 80-byte stride: 2-4x slower than the NCHW form).  The softmax
// statistics of a pixel are recomputed by its Q = ceil(C/4) threads (reads hit L1; ~2*C expf each).
__global__ __launch_bounds__(CE_THREADS) void ce_bwd_nhwc_kernel(const float* __restrict__ z,
                                                                 const long long* __restrict__ tgt,
                                                                 const float* __restrict__ gout, float* __restrict__ dz,
                                                                 long long P, int HW, int C, long long sb, long long sc,
                                                                 long long sp, int ld) {
  const float g = gout[0] / (float)P;
  const int Q = (C + 3) >> 2;
  const long long total = P * Q;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const long long i = idx / Q;
    const int q = (int)(idx - i * Q);
    const long long b = i / HW, hw = i - b * HW;
    const long long off = b * sb + hw * sp;
    float m = z[off];
    for (int c = 1; c < C; ++c) m = fmaxf(m, z[off + c * sc]);
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += expf(z[off + c * sc] - m);
    const float inv = 1.f / s;
    const long long t = tgt[i];
    const float bad = (t < 0 || t >= C) ? __int_as_float(0x7fc00000) : 0.f;  // as the forward: NaN, never a silent 0
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = 4 * q + e;
      v[e] = c < C ? (expf(z[off + c * sc] - m) * inv - (c == t ? 1.f : 0.f)) * g + bad : 0.f;
    }
    *reinterpret_cast<f32x4*>(dz + i * ld + 4 * q) = v;
  }
}

// The same gradient when the NHWC rows are exactly the Q = ceil(C/4) quads (ld == 4*Q, the layout
// ops._CrossEntropy.backward allocates): one thread per PIXEL - logits read once, coalesced along the pixel axis of
// each channel plane, softmax evaluated once - and the wave's 64 finished rows (64*ld contiguous floats) turned
// through LDS so every store instruction writes 1 KB of consecutive gradient.  138 -> ~60 us for 32x19x128x256 on
// MI355X (the quad-per-thread form above re-reads the pixel's logits and re-evaluates 2*C expf in each of its Q threads).
template <int Q>
__global__ __launch_bounds__(256) void ce_bwd_nhwc_rows_kernel(const float* __restrict__ z,
                                                               const long long* __restrict__ tgt,
                                                               const float* __restrict__ gout, float* __restrict__ dz,
                                                               long long P, int HW, int C, long long sb, long long sc,
                                                               long long sp) {
  __shared__ f32x4 sm[4][64 * Q];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float g = gout[0] / (float)P;
  for (long long wg0 = (long long)blockIdx.x * 256; wg0 < P; wg0 += (long long)gridDim.x * 256) {  // uniform trip count
    const long long base = wg0 + wave * 64, i = base + lane;
    if (i < P) {
      const long long b = i / HW, hw = i - b * HW;
      const float* r = z + b * sb + hw * sp;
      float v[4 * Q];
#pragma unroll
      for (int c = 0; c < 4 * Q; ++c) v[c] = c < C ? r[c * sc] : 0.f;
      float m = v[0];
#pragma unroll
      for (int c = 1; c < 4 * Q; ++c) m = c < C ? fmaxf(m, v[c]) : m;
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < 4 * Q; ++c) {
        v[c] = c < C ? expf(v[c] - m) : 0.f;
        s += v[c];
      }
      const float inv = 1.f / s;
      const long long t = tgt[i];
      const float bad = (t < 0 || t >= C) ? __int_as_float(0x7fc00000) : 0.f;  // as the forward: NaN, never a silent 0
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int c = 4 * q + e;
          o[e] = c < C ? (v[c] * inv - (c == t ? 1.f : 0.f)) * g + bad : 0.f;
        }
        sm[wave][lane * Q + q] = o;
      }
    }
    __syncthreads();
    if (base < P) {
      const int nf4 = (int)(P - base < 64 ? P - base : 64) * Q;
      f32x4* out = reinterpret_cast<f32x4*>(dz + base * (4 * Q));
#pragma unroll
      for (int j = 0; j < Q; ++j) {
        const int idx = j * 64 + lane;
        if (idx < nf4) out[idx] = sm[wave][idx];
      }
    }
    __syncthreads();
  }
}

static inline int ce_blocks(long long P) {
  long long nb = cdivll(P, CE_THREADS);
  if (nb > 2048) nb = 2048;
  if (nb < 1) nb = 1;
  return (int)nb;
}

extern "C" long long vmtl_ce_workspace_bytes(long long P) { return ((long long)ce_blocks(P) + 1) * (long long)sizeof(double); }

// workspace: vmtl_ce_workspace_bytes(P) bytes, 8-byte aligned (per-block partial sums).  A target outside [0, C)
// makes the loss NaN.
static int ce_fwd_impl(const float* logits, const long long* target, float* loss, void* workspace, long long* amax,
                       int B, int HW, int C, long long sb, long long sc, long long sp, void* stream) {
  if (!logits || !target || !loss || !workspace || B <= 0 || HW <= 0 || C <= 0) return VMTL_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const long long P = (long long)B * HW;
  const int nblk = ce_blocks(P);
  double* partial = (double*)workspace;
#define CEF(QV)                                                                                              \
  hipLaunchKernelGGL((ce_fwd_regs_kernel<QV>), dim3(nblk), dim3(CE_THREADS), 0, st, logits, target, partial, amax, P, \
                     HW, C, sb, sc, sp)
  switch (C <= 32 ? (C + 3) >> 2 : 0) {
    case 1: CEF(1); break;
    case 2: CEF(2); break;
    case 3: CEF(3); break;
    case 4: CEF(4); break;
    case 5: CEF(5); break;
    case 6: CEF(6); break;
    case 7: CEF(7); break;
    case 8: CEF(8); break;
    default:
      hipLaunchKernelGGL(ce_fwd_kernel, dim3(nblk), dim3(CE_THREADS), 0, st, logits, target, partial, amax, P, HW, C, sb,
                         sc, sp);
  }
#undef CEF
  hipLaunchKernelGGL(sum_finalize_kernel, dim3(1), dim3(256), 0, st, partial, nblk, 1.0 / (double)P, loss,
                     (const int*)nullptr);
  return vmtl_check_launch();
}

extern "C" int vmtl_ce_fwd(const float* logits, const long long* target, float* loss, void* workspace, int B, int HW,
                           int C, long long sb, long long sc, long long sp, void* stream) {
  VMTL_ENTER();
  return ce_fwd_impl(logits, target, loss, workspace, nullptr, B, HW, C, sb, sc, sp, stream);
}

// vmtl_ce_fwd that also emits argmax_c logits (= the segmentation prediction of lit_module.py:137-138): the loss
// pass finds every pixel's maximum anyway, so the prediction costs one 8-byte store instead of a second sweep
// over the logits in its own launch
extern "C" int vmtl_ce_fwd_argmax(const float* logits, const long long* target, float* loss, void* workspace,
                                  long long* argmax, int B, int HW, int C, long long sb, long long sc, long long sp,
                                  void* stream) {
  VMTL_ENTER();
  if (!argmax) return VMTL_ERR_ARG;
  return ce_fwd_impl(logits, target, loss, workspace, argmax, B, HW, C, sb, sc, sp, stream);
}

extern "C" int vmtl_ce_bwd(const float* logits, const long long* target, const float* grad_out, float* dlogits, int B,
                           int HW, int C, long long sb, long long sc, long long sp, void* stream) {
  VMTL_ENTER();
  if (!logits || !target || !grad_out || !dlogits || B <= 0 || HW <= 0 || C <= 0) return VMTL_ERR_ARG;
  const long long P = (long long)B * HW;
  hipLaunchKernelGGL(ce_bwd_kernel, dim3(ce_blocks(P)), dim3(CE_THREADS), 0, (hipStream_t)stream, logits, target,
                     grad_out, dlogits, P, HW, C, sb, sc, sp, sb, sc, sp);
  return vmtl_check_launch();
}

// vmtl_ce_bwd with its own strides for dlogits: element (b, c, hw) of the gradient goes to dlogits[b*dsb + c*dsc +
// hw*dsp].  With (HW*ld, 1, ld) the gradient lands in the NHWC storage the head's data-gradient conv reads (one
// 76-byte run per pixel) and the NCHW -> NHWC relayout of an 80 MB tensor disappears from the step.
extern "C" int vmtl_ce_bwd_strided(const float* logits, const long long* target, const float* grad_out, float* dlogits,
                                   int B, int HW, int C, long long sb, long long sc, long long sp, long long dsb,
                                   long long dsc, long long dsp, void* stream) {
  VMTL_ENTER();
  if (!logits || !target || !grad_out || !dlogits || B <= 0 || HW <= 0 || C <= 0) return VMTL_ERR_ARG;
  const long long P = (long long)B * HW;
  if (dsc == 1 && dsp == ((C + 3) & ~3) && dsp <= 32 && dsb == dsp * HW) {
    long long nb = cdivll(P, 256);
    if (nb > 8192) nb = 8192;
#define CALL(QV)                                                                                                  \
  hipLaunchKernelGGL((ce_bwd_nhwc_rows_kernel<QV>), dim3((int)nb), dim3(256), 0, (hipStream_t)stream, logits, target, \
                     grad_out, dlogits, P, HW, C, sb, sc, sp)
    switch ((int)dsp >> 2) {
      case 1: CALL(1); break;
      case 2: CALL(2); break;
      case 3: CALL(3); break;
      case 4: CALL(4); break;
      case 5: CALL(5); break;
      case 6: CALL(6); break;
      case 7: CALL(7); break;
      default: CALL(8); break;
    }
#undef CALL
    return vmtl_check_launch();
  }
  if (dsc == 1 && (dsp & 3) == 0 && dsp >= ((C + 3) & ~3) && dsb == dsp * HW) {
    long long nb = cdivll(P * ((C + 3) >> 2), CE_THREADS);
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(ce_bwd_nhwc_kernel, dim3((int)nb), dim3(CE_THREADS), 0, (hipStream_t)stream, logits, target,
                       grad_out, dlogits, P, HW, C, sb, sc, sp, (int)dsp);
    return vmtl_check_launch();
  }
  hipLaunchKernelGGL(ce_bwd_kernel, dim3(ce_blocks(P)), dim3(CE_THREADS), 0, (hipStream_t)stream, logits, target,
                     grad_out, dlogits, P, HW, C, sb, sc, sp, dsb, dsc, dsp);
  return vmtl_check_launch();
}

// ------------------------------------------------------------------ SILog
#define SL_BLOCKS_MAX 1024

__global__ __launch_bounds__(256) void silog_fwd_kernel(const float* __restrict__ pred,
                                                        const float* __restrict__ tgt, float min_depth, long long P,
                                                        double* __restrict__ partial) {
  __shared__ double shd[4];
  double n = 0.0, s1 = 0.0, s2 = 0.0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (long long)gridDim.x * blockDim.x) {
    const float t = tgt[i];
    if (t > min_depth) {
      const float g = logf(pred[i]) - logf(t);
      n += 1.0;
      s1 += (double)g;
      s2 += (double)g * (double)g;
    }
  }
  const double bn = block_sum_d(n, shd);
  const double b1 = block_sum_d(s1, shd);
  const double b2 = block_sum_d(s2, shd);
  if (threadIdx.x == 0) {
    partial[blockIdx.x * 3 + 0] = bn;
    partial[blockIdx.x * 3 + 1] = b1;
    partial[blockIdx.x * 3 + 2] = b2;
  }
}

// stats[0] = N, stats[1] = mean(g), stats[2] = Dg ; loss = 10*sqrt(Dg)
__global__ void silog_finalize_kernel(const double* __restrict__ partial, int nblk, float* loss, float* stats) {
  __shared__ double shd[4];
  double n = 0.0, s1 = 0.0, s2 = 0.0;
  for (int i = threadIdx.x; i < nblk; i += blockDim.x) {
    n += partial[i * 3 + 0];
    s1 += partial[i * 3 + 1];
    s2 += partial[i * 3 + 2];
  }
  n = block_sum_d(n, shd);
  s1 = block_sum_d(s1, shd);
  s2 = block_sum_d(s2, shd);
  if (threadIdx.x == 0) {
    const double mu = s1 / n;                          // n == 0 -> NaN, as torch.mean of an empty tensor
    const double var = (s2 - n * mu * mu) / (n - 1.0); // n == 1 -> NaN (unbiased variance), as torch.var
    const double dg = var + 0.15 * mu * mu;
    loss[0] = (float)(10.0 * sqrt(dg));
    stats[0] = (float)n;
    stats[1] = (float)mu;
    stats[2] = (float)dg;
  }
}

// dL/dpred_i = gout * (5/sqrt(Dg)) * (2 (g_i - mu)/(N-1) + 0.3 mu / N) / pred_i   on valid pixels, else 0
__global__ __launch_bounds__(256) void silog_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ tgt,
                                                        const float* __restrict__ stats,
                                                        const float* __restrict__ gout, float min_depth, long long P,
                                                        float* __restrict__ dpred) {
  const float n = stats[0], mu = stats[1], dg = stats[2];
  const float k = gout[0] * 5.f / sqrtf(dg);
  const float a = 2.f / (n - 1.f), b = 0.3f * mu / n;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (long long)gridDim.x * blockDim.x) {
    const float t = tgt[i];
    float r = 0.f;
    if (t > min_depth) {
      const float p = pred[i];
      const float g = logf(p) - logf(t);
      r = k * (a * (g - mu) + b) / p;
    }
    dpred[i] = r;
  }
}

static inline int sl_blocks(long long P) {
  long long nb = cdivll(P, 1024);
  if (nb > SL_BLOCKS_MAX) nb = SL_BLOCKS_MAX;
  if (nb < 1) nb = 1;
  return (int)nb;
}

extern "C" long long vmtl_silog_workspace_bytes(long long P) { return (long long)sl_blocks(P) * 3 * sizeof(double); }

extern "C" int vmtl_silog_fwd(const float* pred, const float* target, float min_depth, float* loss, float* stats,
                              void* workspace, long long P, void* stream) {
  VMTL_ENTER();
  if (!pred || !target || !loss || !stats || !workspace || P <= 0) return VMTL_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int nblk = sl_blocks(P);
  hipLaunchKernelGGL(silog_fwd_kernel, dim3(nblk), dim3(256), 0, st, pred, target, min_depth, P, (double*)workspace);
  hipLaunchKernelGGL(silog_finalize_kernel, dim3(1), dim3(256), 0, st, (const double*)workspace, nblk, loss, stats);
  return vmtl_check_launch();
}

extern "C" int vmtl_silog_bwd(const float* pred, const float* target, const float* stats, const float* grad_out,
                              float min_depth, float* dpred, long long P, void* stream) {
  VMTL_ENTER();
  if (!pred || !target || !stats || !grad_out || !dpred || P <= 0) return VMTL_ERR_ARG;
  hipLaunchKernelGGL(silog_bwd_kernel, dim3(sl_blocks(P)), dim3(256), 0, (hipStream_t)stream, pred, target, stats,
                     grad_out, min_depth, P, dpred);
  return vmtl_check_launch();
}

// ------------------------------------------------------------------ L1 (mean absolute error)
__global__ __launch_bounds__(256) void l1_fwd_kernel(const float* __restrict__ pred, const float* __restrict__ tgt,
                                                     long long P, double* __restrict__ partial) {
  __shared__ double shd[4];
  double s = 0.0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (long long)gridDim.x * blockDim.x)
    s += (double)fabsf(pred[i] - tgt[i]);
  const double b = block_sum_d(s, shd);
  if (threadIdx.x == 0) partial[blockIdx.x] = b;
}

__global__ __launch_bounds__(256) void l1_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ tgt,
                                                     const float* __restrict__ gout, long long P,
                                                     float* __restrict__ dpred) {
  const float g = gout[0] / (float)P;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (long long)gridDim.x * blockDim.x) {
    const float d = pred[i] - tgt[i];
    dpred[i] = d > 0.f ? g : (d < 0.f ? -g : 0.f);
  }
}

extern "C" int vmtl_l1_fwd(const float* pred, const float* target, float* loss, void* workspace, long long P,
                           void* stream) {
  VMTL_ENTER();
  if (!pred || !target || !loss || !workspace || P <= 0) return VMTL_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int nblk = sl_blocks(P);
  hipLaunchKernelGGL(l1_fwd_kernel, dim3(nblk), dim3(256), 0, st, pred, target, P, (double*)workspace);
  hipLaunchKernelGGL(sum_finalize_kernel, dim3(1), dim3(256), 0, st, (const double*)workspace, nblk, 1.0 / (double)P,
                     loss, (const int*)nullptr);
  return vmtl_check_launch();
}

extern "C" int vmtl_l1_bwd(const float* pred, const float* target, const float* grad_out, float* dpred, long long P,
                           void* stream) {
  VMTL_ENTER();
  if (!pred || !target || !grad_out || !dpred || P <= 0) return VMTL_ERR_ARG;
  hipLaunchKernelGGL(l1_bwd_kernel, dim3(sl_blocks(P)), dim3(256), 0, (hipStream_t)stream, pred, target, grad_out, P,
                     dpred);
  return vmtl_check_launch();
}
