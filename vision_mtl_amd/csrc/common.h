// Shared device/host helpers for the vmtl HIP kernels (gfx950 / CDNA4 only).
//
// Activation layout everywhere in this library: NHWC fp32, shape [B][H][W][Cs]
// where Cs ("storage channels") = round_up(C, 4) and channels c in [C, Cs) are
// always ZERO.  Every kernel that writes an activation keeps that invariant, so
// 16-byte loads along the channel axis are always aligned and K-padding in the
// implicit GEMMs contributes exact zeros.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#define VMTL_OK 0
#define VMTL_ERR_ARG (-1)
#define VMTL_ERR_LAUNCH (-2)
#define VMTL_ERR_UNSUPPORTED (-3)

// activation codes shared by the C-ABI and the kernels
#define VMTL_ACT_NONE 0
#define VMTL_ACT_RELU 1
#define VMTL_ACT_HSWISH 2
#define VMTL_ACT_HSIGMOID 3
#define VMTL_ACT_SIGMOID 4

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// Every launching entry point starts with VMTL_ENTER(): hipGetLastError() is per-thread state shared with
// every other HIP user in the process (e.g. an ignored status inside the framework that hosts us), and
// the status returned by vmtl_check_launch() must describe OUR launch only.
#define VMTL_ENTER() ((void)hipGetLastError())

extern thread_local int vmtl_last_hip_error;  // version.hip; read back with vmtl_last_error_string()

static inline int vmtl_check_launch() {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) vmtl_last_hip_error = (int)e;
  return e == hipSuccess ? VMTL_OK : VMTL_ERR_LAUNCH;
}

// Tuning overrides (VMTL_FORCE_TILE, VMTL_BF16X3, ...) are read from the environment ONCE, on first use - never per
// launch (getenv walks the whole environment block: host cost on the eager path).  vmtl_reload_env() bumps the epoch so
// that the next use re-reads them (the parity tests toggle them between cases).
extern int vmtl_env_epoch;  // version.hip
struct EnvInt {
  const char* name;
  int dflt;
  int value = 0;
  int epoch = 0;
};
static inline int env_int(EnvInt& e) {
  if (e.epoch != vmtl_env_epoch) {
    const char* s = getenv(e.name);
    e.value = s ? atoi(s) : e.dflt;
    e.epoch = vmtl_env_epoch;
  }
  return e.value;
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline long long cdivll(long long a, long long b) { return (a + b - 1) / b; }

__device__ __forceinline__ float act_fwd(float v, int act) {
  switch (act) {
    case VMTL_ACT_RELU: return v > 0.f ? v : 0.f;
    case VMTL_ACT_HSWISH: {
      float r = fminf(fmaxf(v + 3.f, 0.f), 6.f);
      return v * r * (1.f / 6.f);
    }
    case VMTL_ACT_HSIGMOID: return fminf(fmaxf(v + 3.f, 0.f), 6.f) * (1.f / 6.f);
    case VMTL_ACT_SIGMOID: return 1.f / (1.f + __expf(-v));
    default: return v;
  }
}

// derivative of act at pre-activation v (torch conventions: relu'(0)=0,
// hardswish' = 0 for v<-3, 1 for v>3, (2v+3)/6 between (torch uses < and >,
// the boundary points take the middle formula), hardsigmoid' = 1/6 on (-3,3)).
__device__ __forceinline__ float act_grad(float v, int act) {
  switch (act) {
    case VMTL_ACT_RELU: return v > 0.f ? 1.f : 0.f;
    case VMTL_ACT_HSWISH:
      return v < -3.f ? 0.f : (v > 3.f ? 1.f : (2.f * v + 3.f) * (1.f / 6.f));
    case VMTL_ACT_HSIGMOID: return (v > -3.f && v < 3.f) ? (1.f / 6.f) : 0.f;
    case VMTL_ACT_SIGMOID: {
      float s = 1.f / (1.f + __expf(-v));
      return s * (1.f - s);
    }
    default: return 1.f;
  }
}

// 64-lane wavefront sum via DPP-free shuffles
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// XCD-aware bijective remap of a linear workgroup id: workgroups that end up
// with consecutive remapped ids share an XCD (and therefore an L2).  Speed
// only; any placement is correct.
// Sum over the four lane quarters (lanes l, l+16, l+32, l+48 of a wave; every lane gets the total).
// (Round 3 tried gfx950's v_permlane16_swap / v_permlane32_swap here - two VALU instructions instead of two ds_bpermute
// round trips.  Every kernel test passed, but MTAN's end-to-end gradients came out up to 18 % wrong in builds where the
// surrounding epilogue code was scheduled differently (tests/test_tight_grads_gpu.py; each of two unrelated source changes
// alone hid it): the swaps sit right in front of an EXEC-mask change, a hazard the compiler does not pad.  ds_bpermute it is.)
__device__ __forceinline__ float quarter_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}

// every lane gets the value its column's lane of quarter 0 holds (lane & 15)
__device__ __forceinline__ float quarter0_bcast(float v) { return __shfl(v, threadIdx.x & 15, 64); }

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}
