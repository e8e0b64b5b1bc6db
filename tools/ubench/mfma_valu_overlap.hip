// Micro-benchmark: do MFMA work and VALU work of DIFFERENT waves on one SIMD overlap?
// Each wave alternates a block of NM MFMAs (matrix pipe) with a block of NV v_fma (vector ALU).
// mode 0: MFMA only, 1: VALU only, 2: both in separate blocks (what a compiler emits for "loads, then MFMAs"),
// 3: both, interleaved 1 MFMA : NV/NM VALU inside the wave.  Waves per SIMD = blocks per CU (1..3).
// build: hipcc -O3 --offload-arch=gfx950 -DVOP=0|1|2 mfma_valu_overlap.hip -o mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define NM 32
#define NV 128

// VOP: 0 = v_fma_f32 (fp32 FMA lanes), 1 = v_add_u32 (integer ALU), 2 = v_pk_fma_f32
#ifndef VOP
#define VOP 0
#endif
#if VOP == 0
#define VALU_OP(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(a))
#elif VOP == 1
#define VALU_OP(x) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(b))
#else
#define VALU_OP(x) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(pb))
#endif

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
#if VOP == 2
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 v[8] = {{1, 1}, {2, 2}, {3, 3}, {4, 4}, {5, 5}, {6, 6}, {7, 7}, {8, 8}};
  f32x2 pb = {b, a};
#else
  float v[8] = {1, 2, 3, 4, 5, 6, 7, 8};
#endif
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0 || MODE == 2) {
#pragma unroll
      for (int i = 0; i < NM; ++i) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[i & 3]) : "v"(a), "v"(b));
    }
    if (MODE == 1 || MODE == 2) {
#pragma unroll
      for (int i = 0; i < NV; ++i) VALU_OP(v[i & 7]);
    }
    if (MODE == 3) {
#pragma unroll
      for (int i = 0; i < NM; ++i) {
        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[i & 3]) : "v"(a), "v"(b));
#pragma unroll
        for (int j = 0; j < NV / NM; ++j) VALU_OP(v[(i * (NV / NM) + j) & 7]);
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#if VOP == 2
  for (int i = 0; i < 8; ++i) s += v[i][0] + v[i][1];
#else
  for (int i = 0; i < 8; ++i) s += v[i];
#endif
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
float run(float* d, int blocks, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks, 256>>>(d, 10);
  hipEventRecord(e0);
  k<MODE><<<blocks, 256>>>(d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  float* d; hipMalloc(&d, 256 * 3 * 256 * 4 * sizeof(float));
  const int iters = 2000;
  for (int wps = 1; wps <= 3; ++wps) {
    const int blocks = 256 * wps;
    float t0 = run<0>(d, blocks, iters), t1 = run<1>(d, blocks, iters), t2 = run<2>(d, blocks, iters), t3 = run<3>(d, blocks, iters);
    printf("VOP %d waves/SIMD %d: mfma-only %.3f ms  valu-only %.3f ms  separate blocks %.3f ms  interleaved %.3f ms  (sum %.3f, max %.3f)\n",
           VOP, wps, t0, t1, t2, t3, t0 + t1, t0 > t1 ? t0 : t1);
  }
  return 0;
}
