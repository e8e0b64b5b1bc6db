// Micro-benchmark for the next round: fp32-accurate GEMM through bf16 MFMAs ("bf16x3" operand split).
//
//   C[M][N] = A[M][K] * B[N][K]^T,  fp32 in / fp32 out, 128x144 workgroup tile, 4 waves, BK = 32,
//   register-staged double-buffered LDS tiles - the structure of conv_igemm_kernel's widest tile.
//
//   f32   : v_mfma_f32_16x16x4_f32 (what the product uses today)
//   bf16x3: every fp32 operand is split EXACTLY into three bf16 values by truncation
//           (a = a1 + a2 + a3, 8 significand bits each) while it is staged into LDS; a product is
//           a1b1 + a1b2 + a2b1 + a1b3 + a2b2 + a3b1 (six v_mfma_f32_16x16x32_bf16, fp32 accumulate);
//           the dropped terms are below 2^-24 relative.  bf16 MFMA issues 16x the FLOPs of fp32 MFMA per
//           cycle, so six of them cost 6/16 of one fp32 MFMA.
//
// Prints the max relative error of both against an fp64 host reference (small problem) and the time on a
// decoder-block-2-sized problem.   build: hipcc -O3 --offload-arch=gfx950 gemm_bf16x3_vs_f32.hip -o gemm_ub
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define BM 128
#define BN 144
#define BK 32
#define TM 2
#define TN 9

// ------------------------------------------------------------------ fp32 MFMA baseline
__global__ __launch_bounds__(256) void gemm_f32(const float* __restrict__ A, const float* __restrict__ B,
                                                float* __restrict__ C, int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                // [2][BM][32]
  float* Bs = smem + 2 * BM * 32;  // [2][BN][32]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, lq = lane >> 4;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int k4 = tid & 7, r0 = tid >> 3;
  f32x4 ra[4], rb[5];
  auto load = [&](int kk) {
#pragma unroll
    for (int i = 0; i < 4; ++i) ra[i] = *reinterpret_cast<const f32x4*>(A + (size_t)(m0 + r0 + 32 * i) * K + kk + k4 * 4);
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int n = n0 + r0 + 32 * i;
      rb[i] = (r0 + 32 * i < BN && n < N) ? *reinterpret_cast<const f32x4*>(B + (size_t)n * K + kk + k4 * 4) : (f32x4){0, 0, 0, 0};
    }
  };
  auto store = [&](int buf) {
    const int ks = (k4 ^ (r0 & 7)) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(As + (buf * BM + r0 + 32 * i) * 32 + ks) = ra[i];
#pragma unroll
    for (int i = 0; i < 5; ++i)
      if (r0 + 32 * i < BN) *reinterpret_cast<f32x4*>(Bs + (buf * BN + r0 + 32 * i) * 32 + ks) = rb[i];
  };
  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};
  load(0);
  store(0);
  __syncthreads();
  int cur = 0;
  for (int kk = 0; kk < K; kk += BK) {
    const bool more = kk + BK < K;
    if (more) load(kk + BK);
    const float* a = As + (cur * BM + wv * 32 + l15) * 32;
    const float* b = Bs + (cur * BN + l15) * 32;
#pragma unroll
    for (int kg = 0; kg < 2; ++kg) {
      const int so = ((kg * 4 + lq) ^ (l15 & 7)) * 4;
      f32x4 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const f32x4*>(a + i * 16 * 32 + so);
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const f32x4*>(b + j * 16 * 32 + so);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i][e], fb[j][e], acc[i][j], 0, 0, 0);
    }
    if (more) store(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wv * 32 + i * 16 + 4 * lq + r, n = n0 + j * 16 + l15;
        if (n < N) C[(size_t)m * N + n] = acc[i][j][r];
      }
}

// ------------------------------------------------------------------ bf16x3 split
// LDS plane p of a tile: [rows][32 bf16] = 64-byte rows; the 16-byte slot s of row r lives at s ^ ((r >> 2) & 3)
// (rows r, r+4, r+8, r+12 of a 16-lane fragment read would otherwise share a bank group).
__device__ __forceinline__ void split3(f32x4 v, u32x2& p1, u32x2& p2, u32x2& p3) {
  unsigned x[4], r1[4], r2[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    x[e] = __float_as_uint(v[e]);
    const float f1 = v[e] - __uint_as_float(x[e] & 0xFFFF0000u);  // exact: the low 16 significand bits
    r1[e] = __float_as_uint(f1);
    const float f2 = f1 - __uint_as_float(r1[e] & 0xFFFF0000u);
    r2[e] = __float_as_uint(f2);
  }
  // pack the high halves of two floats into one dword: {lo = elem 0, hi = elem 1}
  p1 = (u32x2){__builtin_amdgcn_perm(x[1], x[0], 0x07060302u), __builtin_amdgcn_perm(x[3], x[2], 0x07060302u)};
  p2 = (u32x2){__builtin_amdgcn_perm(r1[1], r1[0], 0x07060302u), __builtin_amdgcn_perm(r1[3], r1[2], 0x07060302u)};
  p3 = (u32x2){__builtin_amdgcn_perm(r2[1], r2[0], 0x07060302u), __builtin_amdgcn_perm(r2[3], r2[2], 0x07060302u)};
}

__global__ __launch_bounds__(256) void gemm_bf16x3(const float* __restrict__ A, const float* __restrict__ B,
                                                   float* __restrict__ C, int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem8[];
  // [2 buffers][3 planes][rows][64 bytes]
  unsigned char* As = smem8;
  unsigned char* Bs = smem8 + 2 * 3 * BM * 64;
  constexpr int BNR = 160;  // B rows padded to a multiple of 32 loader rows
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, lq = lane >> 4;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int k4 = tid & 7, r0 = tid >> 3;
  f32x4 ra[4], rb[5];
  auto load = [&](int kk) {
#pragma unroll
    for (int i = 0; i < 4; ++i) ra[i] = *reinterpret_cast<const f32x4*>(A + (size_t)(m0 + r0 + 32 * i) * K + kk + k4 * 4);
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int n = n0 + r0 + 32 * i;
      rb[i] = (r0 + 32 * i < BN && n < N) ? *reinterpret_cast<const f32x4*>(B + (size_t)n * K + kk + k4 * 4) : (f32x4){0, 0, 0, 0};
    }
  };
  auto store = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = r0 + 32 * i;
      const int off = row * 64 + (((k4 >> 1) ^ ((row >> 2) & 3)) << 4) + ((k4 & 1) << 3);
      u32x2 p1, p2, p3;
      split3(ra[i], p1, p2, p3);
      *reinterpret_cast<u32x2*>(As + ((buf * 3 + 0) * BM) * 64 + off) = p1;
      *reinterpret_cast<u32x2*>(As + ((buf * 3 + 1) * BM) * 64 + off) = p2;
      *reinterpret_cast<u32x2*>(As + ((buf * 3 + 2) * BM) * 64 + off) = p3;
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int row = r0 + 32 * i;
      if (row < BN) {
        const int off = row * 64 + (((k4 >> 1) ^ ((row >> 2) & 3)) << 4) + ((k4 & 1) << 3);
        u32x2 p1, p2, p3;
        split3(rb[i], p1, p2, p3);
        *reinterpret_cast<u32x2*>(Bs + ((buf * 3 + 0) * BNR) * 64 + off) = p1;
        *reinterpret_cast<u32x2*>(Bs + ((buf * 3 + 1) * BNR) * 64 + off) = p2;
        *reinterpret_cast<u32x2*>(Bs + ((buf * 3 + 2) * BNR) * 64 + off) = p3;
      }
    }
  };
  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};
  load(0);
  store(0);
  __syncthreads();
  int cur = 0;
  for (int kk = 0; kk < K; kk += BK) {
    const bool more = kk + BK < K;
    if (more) load(kk + BK);
    // lane (l15, lq) holds k = 8*lq .. 8*lq+7 of its row: the 16-byte slot lq
    bf16x8 fa[TM][3];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int row = wv * 32 + i * 16 + l15;
      const int off = row * 64 + ((lq ^ ((row >> 2) & 3)) << 4);
#pragma unroll
      for (int s = 0; s < 3; ++s) fa[i][s] = *reinterpret_cast<const bf16x8*>(As + ((cur * 3 + s) * BM) * 64 + off);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int row = j * 16 + l15;
      const int off = row * 64 + ((lq ^ ((row >> 2) & 3)) << 4);
      bf16x8 fb[3];
#pragma unroll
      for (int s = 0; s < 3; ++s) fb[s] = *reinterpret_cast<const bf16x8*>(Bs + ((cur * 3 + s) * BNR) * 64 + off);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        f32x4 c = acc[i][j];
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][2], fb[0], c, 0, 0, 0);  // smallest terms first
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][1], fb[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][0], fb[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][1], fb[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][0], fb[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][0], fb[0], c, 0, 0, 0);
        acc[i][j] = c;
      }
    }
    if (more) store(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wv * 32 + i * 16 + 4 * lq + r, n = n0 + j * 16 + l15;
        if (n < N) C[(size_t)m * N + n] = acc[i][j][r];
      }
}

// ------------------------------------------------------------------ bf16x3 with operands split ONCE
// A3 / B3: three bf16 planes [3][rows][K] written by a pre-pass (in the product: by the producing kernel's
// epilogue / the weight packer); the GEMM then stages 16-byte bf16 chunks without any VALU work.
__global__ void split_planes(const float* __restrict__ x, unsigned short* __restrict__ p3, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = x[i];
  const unsigned b0 = __float_as_uint(v);
  const float f1 = v - __uint_as_float(b0 & 0xFFFF0000u);
  const unsigned b1 = __float_as_uint(f1);
  const float f2 = f1 - __uint_as_float(b1 & 0xFFFF0000u);
  p3[i] = (unsigned short)(b0 >> 16);
  p3[n + i] = (unsigned short)(b1 >> 16);
  p3[2 * n + i] = (unsigned short)(__float_as_uint(f2) >> 16);
}

__global__ __launch_bounds__(256) void gemm_bf16x3_presplit(const unsigned short* __restrict__ A3,
                                                            const unsigned short* __restrict__ B3,
                                                            float* __restrict__ C, int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem8[];
  unsigned char* As = smem8;                    // [2][3][BM][64 B]
  unsigned char* Bs = smem8 + 2 * 3 * BM * 64;  // [2][3][BN][64 B]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, lq = lane >> 4;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  // a 32-k chunk of one row of one plane is 64 bytes = 4 x 16 B: thread -> (row = tid >> 2, slot = tid & 3)
  const int slot = tid & 3, rr = tid >> 2;  // 64 rows per pass
  const size_t planeA = (size_t)M * K, planeB = (size_t)N * K;
  u32x4 ra[3][2], rb[3][3];
  auto load = [&](int kk) {
#pragma unroll
    for (int s = 0; s < 3; ++s) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
        ra[s][i] = *reinterpret_cast<const u32x4*>(A3 + s * planeA + (size_t)(m0 + rr + 64 * i) * K + kk + slot * 8);
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int row = rr + 64 * i, n = n0 + row;
        rb[s][i] = (row < BN && n < N) ? *reinterpret_cast<const u32x4*>(B3 + s * planeB + (size_t)n * K + kk + slot * 8)
                                       : (u32x4){0, 0, 0, 0};
      }
    }
  };
  auto store = [&](int buf) {
#pragma unroll
    for (int s = 0; s < 3; ++s) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = rr + 64 * i;
        *reinterpret_cast<u32x4*>(As + ((buf * 3 + s) * BM + row) * 64 + ((slot ^ ((row >> 2) & 3)) << 4)) = ra[s][i];
      }
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int row = rr + 64 * i;
        if (row < BN) *reinterpret_cast<u32x4*>(Bs + ((buf * 3 + s) * BN + row) * 64 + ((slot ^ ((row >> 2) & 3)) << 4)) = rb[s][i];
      }
    }
  };
  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};
  load(0);
  store(0);
  __syncthreads();
  int cur = 0;
  for (int kk = 0; kk < K; kk += BK) {
    const bool more = kk + BK < K;
    if (more) load(kk + BK);
    bf16x8 fa[TM][3];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int row = wv * 32 + i * 16 + l15;
      const int off = row * 64 + ((lq ^ ((row >> 2) & 3)) << 4);
#pragma unroll
      for (int s = 0; s < 3; ++s) fa[i][s] = *reinterpret_cast<const bf16x8*>(As + ((cur * 3 + s) * BM) * 64 + off);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int row = j * 16 + l15;
      const int off = row * 64 + ((lq ^ ((row >> 2) & 3)) << 4);
      bf16x8 fb[3];
#pragma unroll
      for (int s = 0; s < 3; ++s) fb[s] = *reinterpret_cast<const bf16x8*>(Bs + ((cur * 3 + s) * BN) * 64 + off);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        f32x4 c = acc[i][j];
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][2], fb[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][1], fb[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][0], fb[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][1], fb[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][0], fb[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][0], fb[0], c, 0, 0, 0);
        acc[i][j] = c;
      }
    }
    if (more) store(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wv * 32 + i * 16 + 4 * lq + r, n = n0 + j * 16 + l15;
        if (n < N) C[(size_t)m * N + n] = acc[i][j][r];
      }
}

static unsigned short *gA3 = nullptr, *gB3 = nullptr;

static float time_kernel(int which, const float* A, const float* B, float* C, int M, int N, int K, int reps) {
  const dim3 grid(M / BM, (N + BN - 1) / BN);
  const size_t lds_f32 = (size_t)2 * (BM + BN) * 32 * 4, lds_bf = (size_t)2 * 3 * (BM + 160) * 64;
  hipFuncSetAttribute((const void*)gemm_f32, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_f32);
  hipFuncSetAttribute((const void*)gemm_bf16x3, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bf);
  const size_t lds_ps = (size_t)2 * 3 * (BM + BN) * 64;
  hipFuncSetAttribute((const void*)gemm_bf16x3_presplit, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_ps);
  if (which == 2) {
    (void)hipFree(gA3); (void)hipFree(gB3);
    (void)hipMalloc(&gA3, (size_t)3 * M * K * 2); (void)hipMalloc(&gB3, (size_t)3 * N * K * 2);
    split_planes<<<(unsigned)(((long long)M * K + 255) / 256), 256>>>(A, gA3, (long long)M * K);
    split_planes<<<(unsigned)(((long long)N * K + 255) / 256), 256>>>(B, gB3, (long long)N * K);
  }
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int r = 0; r < reps + 1; ++r) {
    if (r == 1) (void)hipEventRecord(e0);
    if (which == 0) gemm_f32<<<grid, 256, lds_f32>>>(A, B, C, M, N, K);
    else if (which == 1) gemm_bf16x3<<<grid, 256, lds_bf>>>(A, B, C, M, N, K);
    else gemm_bf16x3_presplit<<<grid, 256, lds_ps>>>(gA3, gB3, C, M, N, K);
  }
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

int main() {
  // ---- accuracy on a small problem against fp64
  {
    const int M = 256, N = 144, K = 1024;
    std::vector<float> hA((size_t)M * K), hB((size_t)N * K), hC((size_t)M * N);
    srand(1);
    for (auto& v : hA) v = (rand() / (float)RAND_MAX - 0.5f) * 4.f;
    for (auto& v : hB) v = (rand() / (float)RAND_MAX - 0.5f) * 0.1f;
    float *dA, *dB, *dC;
    (void)hipMalloc(&dA, hA.size() * 4); (void)hipMalloc(&dB, hB.size() * 4); (void)hipMalloc(&dC, hC.size() * 4);
    (void)hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
    std::vector<double> ref((size_t)M * N);
    double scale = 0;
    for (int m = 0; m < M; ++m)
      for (int n = 0; n < N; ++n) {
        double s = 0;
        for (int k = 0; k < K; ++k) s += (double)hA[(size_t)m * K + k] * (double)hB[(size_t)n * K + k];
        ref[(size_t)m * N + n] = s;
        scale = fmax(scale, fabs(s));
      }
    for (int which = 0; which < 3; ++which) {
      time_kernel(which, dA, dB, dC, M, N, K, 1);
      (void)hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost);
      double err = 0;
      for (size_t i = 0; i < hC.size(); ++i) err = fmax(err, fabs((double)hC[i] - ref[i]));
      printf("%s: max |err| / max |C| = %.3e   (K = %d)\n", which == 0 ? "f32            " : which == 1 ? "bf16x3         " : "bf16x3 presplit", err / scale, K);
    }
    (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dC);
  }
  // ---- time on decoder-block-2-like shapes
  const int shapes[][3] = {{65536, 144, 1216}, {262144, 144, 608}, {16384, 288, 2432}};
  for (auto& s : shapes) {
    const int M = s[0], N = s[1], K = s[2];
    float *dA, *dB, *dC;
    (void)hipMalloc(&dA, (size_t)M * K * 4); (void)hipMalloc(&dB, (size_t)N * K * 4); (void)hipMalloc(&dC, (size_t)M * N * 4);
    (void)hipMemset(dA, 0x3c, (size_t)M * K * 4); (void)hipMemset(dB, 0x3c, (size_t)N * K * 4);
    const double fl = 2.0 * M * N * K;
    const float t0 = time_kernel(0, dA, dB, dC, M, N, K, 5), t1 = time_kernel(1, dA, dB, dC, M, N, K, 5),
                t2 = time_kernel(2, dA, dB, dC, M, N, K, 5);
    printf("M=%7d N=%4d K=%5d   f32 MFMA %7.1f us %6.1f TF | bf16x3 split in staging %7.1f us %6.1f TF x%.2f | operands pre-split "
           "%7.1f us %6.1f TF x%.2f\n", M, N, K, t0 * 1e3, fl / (t0 * 1e-3) / 1e12, t1 * 1e3, fl / (t1 * 1e-3) / 1e12, t0 / t1,
           t2 * 1e3, fl / (t2 * 1e-3) / 1e12, t0 / t2);
    (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dC);
  }
  return 0;
}
