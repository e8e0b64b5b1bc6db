// Per-image fully connected layers (batch-sized GEMMs, M = B <= 64 rows) and the per-image spatial
// reductions around them: the squeeze-excite gate of the MobileNetV3 encoder (timm SqueezeExcite, the
// `basic` model's encoder: reference vision_mtl/models/basic_model.py:17-28 via smp/timm).
//
// A 32 x 960 x 240 product has two 64x144 tiles: run through the implicit-GEMM conv kernel it occupies two
// CUs for 30 dependent K steps.  Here a workgroup owns 16 output columns and ALL rows; its four waves split
// K, load both MFMA operands straight from global memory (no LDS staging: nothing is shared between waves)
// and combine through LDS once at the end.  N/16 workgroups, ~K/64 loads deep.
//
// The A operand can be handed over as `a_parts` partial sums ([parts][M][lda], summed and scaled on load):
// that is how the spatial mean / the gate gradient reach the GEMM without a finalize launch.  With `a_z`
// the A operand is multiplied by act'(a_z) on load (activation backward fused into the data gradient).
#include "common.h"

struct FcP {
  const float* a;     // [a_parts][M][lda]
  const float* a_z;   // nullable [M][lda]: pre-activations whose act_grad masks the A operand
  const float* w;     // [N][ldw], K contiguous
  const float* bias;  // nullable [N]
  float* a_out;       // nullable [M][lda]: the finished A operand (parts summed and scaled, before the a_z mask)
  float* z;           // nullable [M][ldy] pre-activation output
  float* y;           // [M][ldy] = act(z); pad columns zero
  long long a_part_stride;
  float a_scale;
  int a_parts, a_act;
  int M, K, N, lda, ldw, ldy, act;
};

#define FC_WAVES 8   // waves per workgroup, all splitting K
#define FC_UNROLL 4  // K chunks whose loads are issued together (the loop is load-latency bound)

template <int MT>
__global__ __launch_bounds__(FC_WAVES * 64) void fc_kernel(FcP p) {
  __shared__ f32x4 red[FC_WAVES][MT][64];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int l15 = lane & 15, lq = lane >> 4;
  const int n0 = blockIdx.x * 16;
  const int n = n0 + l15;
  f32x4 acc[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int nchunks = (p.K + 15) >> 4;
  const bool wvec = (p.ldw & 3) == 0;
  for (int c0 = wv; c0 < nchunks; c0 += FC_WAVES * FC_UNROLL) {
    f32x4 fb[FC_UNROLL], fa[FC_UNROLL][MT];
#pragma unroll
    for (int u = 0; u < FC_UNROLL; ++u) {
      const int c = c0 + u * FC_WAVES;
      const int k4 = c * 16 + lq * 4;
      fb[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (c < nchunks && n < p.N && k4 < p.K) {
        const float* wp = p.w + (size_t)n * p.ldw + k4;
        if (wvec && k4 + 3 < p.K) {
          fb[u] = *reinterpret_cast<const f32x4*>(wp);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (k4 + e < p.K) fb[u][e] = wp[e];
        }
      }
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        const int m = t * 16 + l15;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (c < nchunks && m < p.M && k4 < p.lda) {
          const float* ap = p.a + (size_t)m * p.lda + k4;
          v = *reinterpret_cast<const f32x4*>(ap);
          for (int s = 1; s < p.a_parts; ++s) v += *reinterpret_cast<const f32x4*>(ap + s * p.a_part_stride);
          v *= p.a_scale;
          // every workgroup computes the same finished operand: workgroup (c mod grid) keeps chunk c
          if (p.a_out != nullptr && (c % (int)gridDim.x) == (int)blockIdx.x)
            *reinterpret_cast<f32x4*>(p.a_out + (size_t)m * p.lda + k4) = v;
          if (p.a_z != nullptr) {
            const f32x4 zv = *reinterpret_cast<const f32x4*>(p.a_z + (size_t)m * p.lda + k4);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= act_grad(zv[e], p.a_act);
          }
        }
        fa[u][t] = v;
      }
    }
#pragma unroll
    for (int u = 0; u < FC_UNROLL; ++u)
#pragma unroll
      for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[u][t][e], fb[u][e], acc[t], 0, 0, 0);
  }
#pragma unroll
  for (int t = 0; t < MT; ++t) red[wv][t][lane] = acc[t];
  __syncthreads();
  // C layout of 16x16x4: col = lane & 15, row = 4 * (lane >> 4) + reg.  Thread (ln, rg) finishes element rg of
  // lane ln's accumulator, tiles rt, rt + 2, ... (8 waves -> 2 tiles in flight)
  if (tid >= 512) return;
  const int ln = tid & 63, rg = (tid >> 6) & 3, rt = tid >> 8;
  const int col = ln & 15, row = 4 * (ln >> 4) + rg;
  const int nn = n0 + col;
  if (nn >= p.ldy) return;
  const float bv = (p.bias != nullptr && nn < p.N) ? p.bias[nn] : 0.f;
  for (int t = rt; t < MT; t += 2) {
    const int m = t * 16 + row;
    if (m >= p.M) continue;
    const float* r0 = reinterpret_cast<const float*>(&red[0][t][ln]) + rg;
    constexpr int WS = MT * 64 * 4;  // floats between the per-wave copies
    float v = bv;
#pragma unroll
    for (int w = 0; w < FC_WAVES; ++w) v += r0[w * WS];
    if (nn >= p.N) v = 0.f;
    if (p.z != nullptr) p.z[(size_t)m * p.ldy + nn] = v;
    p.y[(size_t)m * p.ldy + nn] = nn < p.N ? act_fwd(v, p.act) : 0.f;
  }
}

extern "C" int vmtl_fc_max_rows() { return 64; }

extern "C" int vmtl_fc_fwd(const float* a, int a_parts, long long a_part_stride, float a_scale, const float* a_z,
                           int a_act, float* a_out, const float* w, const float* bias, float* z, float* y, int M,
                           int K, int N, int lda, int ldw, int ldy, int act, void* stream) {
  VMTL_ENTER();
  if (!a || !w || !y || M <= 0 || M > 64 || K <= 0 || N <= 0 || (lda & 3) || lda < K || ldw < K || ldy < N ||
      a_parts < 1 || (a_parts > 1 && (a_part_stride & 3)))
    return VMTL_ERR_ARG;
  FcP p{a, a_z, w, bias, a_out, z, y, a_part_stride, a_scale, a_parts, a_act, M, K, N, lda, ldw, ldy, act};
  const dim3 grid(cdiv(ldy, 16));
  hipStream_t st = (hipStream_t)stream;
  switch (cdiv(M, 16)) {
    case 1: hipLaunchKernelGGL(fc_kernel<1>, grid, dim3(FC_WAVES * 64), 0, st, p); break;
    case 2: hipLaunchKernelGGL(fc_kernel<2>, grid, dim3(FC_WAVES * 64), 0, st, p); break;
    case 3: hipLaunchKernelGGL(fc_kernel<3>, grid, dim3(FC_WAVES * 64), 0, st, p); break;
    default: hipLaunchKernelGGL(fc_kernel<4>, grid, dim3(FC_WAVES * 64), 0, st, p); break;
  }
  return vmtl_check_launch();
}

// dw[n][k] = sum_m dz[m][n] * x[m][k],  db[n] = sum_m dz[m][n],  dz = dyo * act'(zo).  Written in the torch
// parameter layout ((N, K, 1, 1) contiguous): no unpack pass.  8 weight rows per workgroup.
struct FcWgP {
  const float* x;  // [x_parts][M][lda]
  const float* dyo;  // [dy_parts][M][ldn]
  const float* zo;   // nullable [M][ldn]
  float* dw;
  float* db;  // nullable
  long long x_part_stride, dy_part_stride;
  float x_scale;
  int x_parts, dy_parts, M, K, N, lda, ldn, act;
};

__global__ __launch_bounds__(256) void fc_wgrad_kernel(FcWgP p) {
  __shared__ float dz[64][8];
  const int tid = threadIdx.x;
  const int n0 = blockIdx.x * 8;
  for (int idx = tid; idx < p.M * 8; idx += 256) {
    const int m = idx >> 3, i = idx & 7, n = n0 + i;
    float v = 0.f;
    if (n < p.N) {
      const size_t off = (size_t)m * p.ldn + n;
      v = p.dyo[off];
      for (int s = 1; s < p.dy_parts; ++s) v += p.dyo[off + s * p.dy_part_stride];
      if (p.zo != nullptr) v *= act_grad(p.zo[off], p.act);
    }
    dz[m][i] = v;
  }
  __syncthreads();
  for (int k = tid; k < p.K; k += 256) {
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    for (int mb = 0; mb < p.M; mb += 8) {  // 8 independent loads in flight per thread
      float xv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        xv[u] = 0.f;
        if (mb + u < p.M) {
          const size_t off = (size_t)(mb + u) * p.lda + k;
          xv[u] = p.x[off];
          for (int s = 1; s < p.x_parts; ++s) xv[u] += p.x[off + s * p.x_part_stride];
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (mb + u < p.M) {
          const float x1 = xv[u] * p.x_scale;
#pragma unroll
          for (int i = 0; i < 8; ++i) acc[i] += dz[mb + u][i] * x1;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (n0 + i < p.N) p.dw[(size_t)(n0 + i) * p.K + k] = acc[i];
  }
  if (p.db != nullptr && tid < 8 && n0 + tid < p.N) {
    float s = 0.f;
    for (int m = 0; m < p.M; ++m) s += dz[m][tid];
    p.db[n0 + tid] = s;
  }
}

extern "C" int vmtl_fc_wgrad(const float* x, int x_parts, long long x_part_stride, float x_scale, const float* dyo,
                             int dy_parts, long long dy_part_stride, const float* zo, float* dw, float* db, int M,
                             int K, int N, int lda, int ldn, int act, void* stream) {
  VMTL_ENTER();
  if (!x || !dyo || !dw || M <= 0 || M > 64 || K <= 0 || N <= 0 || lda < K || ldn < N || x_parts < 1 || dy_parts < 1)
    return VMTL_ERR_ARG;
  FcWgP p{x, dyo, zo, dw, db, x_part_stride, dy_part_stride, x_scale, x_parts, dy_parts, M, K, N, lda, ldn, act};
  hipLaunchKernelGGL(fc_wgrad_kernel, dim3(cdiv(N, 8)), dim3(256), 0, (hipStream_t)stream, p);
  return vmtl_check_launch();
}

// ------------------------------------------------------------------ per-image spatial reductions
// part[s][b][c] = sum over the s-th slice of HW of x[b][hw][c] (* y[b][hw][c] when y is given).
// Threads keep a fixed float4 channel group; 256 / CQ row lanes walk the slice in parallel.
__global__ __launch_bounds__(256) void hw_reduce_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                        float* __restrict__ part, int B, int HW, int Cs, int rows_per) {
  __shared__ f32x4 red[256];
  const int CQ = Cs >> 2;
  const int cq = min(CQ, 256);         // channel groups handled by this workgroup column
  const int RP = 256 / cq;             // row lanes
  const int tid = threadIdx.x;
  const int ql = tid % cq, rg = tid / cq;
  const int q = blockIdx.x * 256 + ql;
  const int b = blockIdx.y, s = blockIdx.z;
  const int r_begin = s * rows_per, r_end = min(HW, r_begin + rows_per);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (rg < RP && q < CQ) {
    const float* xb = x + (size_t)b * HW * Cs + (size_t)q * 4;
    const float* yb = y != nullptr ? y + (size_t)b * HW * Cs + (size_t)q * 4 : nullptr;
    // 4 rows per trip: the loads are independent, the chain of adds is what serialises a plain loop
    f32x4 a1 = {0.f, 0.f, 0.f, 0.f}, a2 = a1, a3 = a1;
    int r = r_begin + rg;
    for (; r + 3 * RP < r_end; r += 4 * RP) {
      f32x4 v0 = *reinterpret_cast<const f32x4*>(xb + (size_t)r * Cs);
      f32x4 v1 = *reinterpret_cast<const f32x4*>(xb + (size_t)(r + RP) * Cs);
      f32x4 v2 = *reinterpret_cast<const f32x4*>(xb + (size_t)(r + 2 * RP) * Cs);
      f32x4 v3 = *reinterpret_cast<const f32x4*>(xb + (size_t)(r + 3 * RP) * Cs);
      if (yb != nullptr) {
        v0 *= *reinterpret_cast<const f32x4*>(yb + (size_t)r * Cs);
        v1 *= *reinterpret_cast<const f32x4*>(yb + (size_t)(r + RP) * Cs);
        v2 *= *reinterpret_cast<const f32x4*>(yb + (size_t)(r + 2 * RP) * Cs);
        v3 *= *reinterpret_cast<const f32x4*>(yb + (size_t)(r + 3 * RP) * Cs);
      }
      acc += v0;
      a1 += v1;
      a2 += v2;
      a3 += v3;
    }
    for (; r < r_end; r += RP) {
      f32x4 v = *reinterpret_cast<const f32x4*>(xb + (size_t)r * Cs);
      if (yb != nullptr) v *= *reinterpret_cast<const f32x4*>(yb + (size_t)r * Cs);
      acc += v;
    }
    acc += a1 + (a2 + a3);
  }
  red[tid] = acc;
  __syncthreads();
  if (rg == 0 && q < CQ) {
    f32x4 t = red[ql];
    for (int g = 1; g < RP; ++g) t += red[g * cq + ql];  // fixed order: deterministic
    *reinterpret_cast<f32x4*>(part + ((size_t)s * B + b) * Cs + (size_t)q * 4) = t;
  }
}

// number of HW slices (= leading dimension of the partial buffer)
extern "C" int vmtl_hw_reduce_parts(int B, int HW, int Cs) {
  if (B <= 0 || HW <= 0 || Cs <= 0) return 0;
  const int cols = cdiv(Cs >> 2, 256);
  // ~128 workgroups; every extra slice is an extra operand read in the GEMMs that consume the partials
  int want = cdiv(128, B * cols);
  int cap = cdiv(HW, 32);                 // at least 32 rows per slice
  int S = want < cap ? want : cap;
  if (S < 1) S = 1;
  if (S > 32) S = 32;
  return cdiv(HW, cdiv(HW, S));           // drop empty slices
}

extern "C" int vmtl_hw_reduce(const float* x, const float* y, float* part, int B, int HW, int Cs, void* stream) {
  VMTL_ENTER();
  if (!x || !part || (Cs & 3) || B <= 0 || HW <= 0) return VMTL_ERR_ARG;
  const int S = vmtl_hw_reduce_parts(B, HW, Cs);
  const int rows_per = cdiv(HW, S);
  hipLaunchKernelGGL(hw_reduce_kernel, dim3(cdiv(Cs >> 2, 256), B, S), dim3(256), 0, (hipStream_t)stream, x, y, part, B,
                     HW, Cs, rows_per);
  return vmtl_check_launch();
}

// y[b][hw][c] = x[b][hw][c] * s[b][c] + t[b][c] * t_scale     (t nullable)
__global__ __launch_bounds__(256) void channel_scale_add_kernel(const float* __restrict__ x, const float* __restrict__ s,
                                                                const float* __restrict__ t, float t_scale,
                                                                float* __restrict__ y, int HW, int Cs, long long total4) {
  const int CQ = Cs >> 2;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
    const int q = (int)(i % CQ);
    const int b = (int)(i / ((long long)CQ * HW));
    f32x4 v = reinterpret_cast<const f32x4*>(x)[i] * *reinterpret_cast<const f32x4*>(s + (size_t)b * Cs + (size_t)q * 4);
    if (t != nullptr) v += *reinterpret_cast<const f32x4*>(t + (size_t)b * Cs + (size_t)q * 4) * t_scale;
    reinterpret_cast<f32x4*>(y)[i] = v;
  }
}

extern "C" int vmtl_channel_scale_add(const float* x, const float* s, const float* t, float t_scale, float* y, int B,
                                      int HW, int Cs, void* stream) {
  VMTL_ENTER();
  if (!x || !s || !y || (Cs & 3) || B <= 0 || HW <= 0) return VMTL_ERR_ARG;
  const long long total4 = (long long)B * HW * (Cs >> 2);
  long long blocks = (total4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(channel_scale_add_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, s, t, t_scale,
                     y, HW, Cs, total4);
  return vmtl_check_launch();
}
