// Deterministic two-stage per-channel (column) reduction skeleton shared by the
// BatchNorm, bias-gradient, depthwise-weight-gradient and stitch-gradient kernels.
#pragma once
#include "common.h"

#define RED_THREADS 256
#define RED_MAX_BLOCKS 1024
#ifndef VMTL_RIF
#define VMTL_RIF 2  // rows whose loads are in flight per thread in the split load/accumulate sweeps
#endif

// Column reduction skeleton.  The [M][CQ] float4 matrix is swept by T = (256/CQ)*CQ
// threads so that a thread keeps ONE float4 column (q fixed) while stepping rows;
// consecutive threads touch consecutive 16-byte chunks (fully coalesced).  Matrices
// with CQ > 256 are handled in column panels of 256.
template <int K, typename F>
__device__ __forceinline__ void column_reduce(int M, int CQ, int Cs, float* partial, F f) {
  __shared__ f32x4 red[RED_THREADS];
  const int nblk = gridDim.x;
  const int rows_per_blk = (M + nblk - 1) / nblk;
  const int r_begin = blockIdx.x * rows_per_blk;
  const int r_end = min(M, r_begin + rows_per_blk);
  for (int q0 = 0; q0 < CQ; q0 += RED_THREADS) {
    const int cq = min(CQ - q0, RED_THREADS);  // columns in this panel
    const int rpt = RED_THREADS / cq;           // rows swept per iteration
    const int T = rpt * cq;
    const int t = threadIdx.x;
    const int q = q0 + t % cq, ro = t / cq;
    f32x4 acc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) acc[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (t < T)
      for (int r = r_begin + ro; r < r_end; r += rpt) f(r, q, acc);
#pragma unroll
    for (int k = 0; k < K; ++k) {
      __syncthreads();
      red[t] = acc[k];
      __syncthreads();
      if (t < cq) {
        f32x4 s = red[t];
        for (int j = 1; j < rpt; ++j) s += red[t + j * cq];
        *reinterpret_cast<f32x4*>(partial + ((size_t)blockIdx.x * K + k) * Cs + (size_t)(q0 + t) * 4) = s;
      }
    }
  }
}

// column_reduce with a per-thread state built once per column panel (hoists per-channel parameters)
template <int K, typename Init, typename F>
__device__ __forceinline__ void column_reduce_init(int M, int CQ, int Cs, float* partial, Init init, F f) {
  __shared__ f32x4 red[RED_THREADS];
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
    f32x4 acc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) acc[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (t < T) {
      auto st = init(q);
      for (int r = r_begin + ro; r < r_end; r += rpt) f(r, q, st, acc);
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
      __syncthreads();
      red[t] = acc[k];
      __syncthreads();
      if (t < cq) {
        f32x4 s = red[t];
        for (int j = 1; j < rpt; ++j) s += red[t + j * cq];
        *reinterpret_cast<f32x4*>(partial + ((size_t)blockIdx.x * K + k) * Cs + (size_t)(q0 + t) * 4) = s;
      }
    }
  }
}

// column_reduce_init with the row loop split into LOAD and ACCUMULATE so that two rows' loads are issued before
// either row's arithmetic (the plain form has one row = 2-3 loads in flight per thread; these sweeps are
// latency-bound at 4 workgroups per CU)
template <int K, typename Init, typename Load, typename Accum>
__device__ __forceinline__ void column_reduce_init2(int M, int CQ, int Cs, float* partial, Init init, Load load, Accum accum) {
  __shared__ f32x4 red[RED_THREADS];
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
    f32x4 acc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) acc[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (t < T) {
      auto st = init(q);
      int r = r_begin + ro;
      for (; r + (VMTL_RIF - 1) * rpt < r_end; r += VMTL_RIF * rpt) {
        decltype(load(r, q)) l[VMTL_RIF];
#pragma unroll
        for (int u = 0; u < VMTL_RIF; ++u) l[u] = load(r + u * rpt, q);
#pragma unroll
        for (int u = 0; u < VMTL_RIF; ++u) accum(l[u], r + u * rpt, q, st, acc);
      }
      for (; r < r_end; r += rpt) {
        auto l0 = load(r, q);
        accum(l0, r, q, st, acc);
      }
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
      __syncthreads();
      red[t] = acc[k];
      __syncthreads();
      if (t < cq) {
        f32x4 s = red[t];
        for (int j = 1; j < rpt; ++j) s += red[t + j * cq];
        *reinterpret_cast<f32x4*>(partial + ((size_t)blockIdx.x * K + k) * Cs + (size_t)(q0 + t) * 4) = s;
      }
    }
  }
}

static inline int red_blocks(int M) {
  int nb = cdiv(M, 16);  // small maps with many channels still get a few hundred workgroups
  if (nb > RED_MAX_BLOCKS) nb = RED_MAX_BLOCKS;
  if (nb < 1) nb = 1;
  return nb;
}


// Sum partial[(b*K + k)*Cs + c] over b = 0..nblk-1 for the workgroup's channel, in fp64, by a
// 256-thread workgroup (fixed association order -> run-to-run reproducible).  Every thread returns the sum.
__device__ __forceinline__ double block_rows_sum(const float* __restrict__ partial, int nblk, int K, int k, int Cs,
                                                 int c, double* sh /* >= 4 doubles */) {
  double s = 0.0;
  for (int b = threadIdx.x; b < nblk; b += blockDim.x) s += (double)partial[((size_t)b * K + k) * Cs + c];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  double t = 0.0;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sh[i];
  return t;
}

// fp64 sum over a 256-thread workgroup; every thread returns the total
__device__ __forceinline__ double block_sum(double s, double* sh /* >= 4 doubles */) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  double t = 0.0;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sh[i];
  return t;
}

// Row sweep without a reduction: same thread -> (row offset, float4 column) mapping as
// column_reduce, so a thread keeps ONE channel quad for all its rows and per-channel parameters
// are loaded once (init) instead of per element.  Consecutive threads touch consecutive 16-byte
// chunks of a row (then the next row): fully coalesced for any Cs.
template <typename Init, typename Body>
__device__ __forceinline__ void column_sweep(int M, int CQ, Init init, Body body) {
  const int nblk = gridDim.x;
  const int rows_per_blk = (M + nblk - 1) / nblk;
  const int r_begin = blockIdx.x * rows_per_blk;
  const int r_end = min(M, r_begin + rows_per_blk);
  for (int q0 = 0; q0 < CQ; q0 += RED_THREADS) {
    const int cq = min(CQ - q0, RED_THREADS);
    const int rpt = RED_THREADS / cq;
    const int t = threadIdx.x;
    if (t >= rpt * cq) continue;
    const int q = q0 + t % cq, ro = t / cq;
    auto st = init(q);
    for (int r = r_begin + ro; r < r_end; r += rpt) body(r, q, st);
  }
}

// column_sweep with LOAD and BODY split: two rows' loads are issued before either row's arithmetic / store
template <typename Init, typename Load, typename Body>
__device__ __forceinline__ void column_sweep2(int M, int CQ, Init init, Load load, Body body) {
  const int nblk = gridDim.x;
  const int rows_per_blk = (M + nblk - 1) / nblk;
  const int r_begin = blockIdx.x * rows_per_blk;
  const int r_end = min(M, r_begin + rows_per_blk);
  for (int q0 = 0; q0 < CQ; q0 += RED_THREADS) {
    const int cq = min(CQ - q0, RED_THREADS);
    const int rpt = RED_THREADS / cq;
    const int t = threadIdx.x;
    if (t >= rpt * cq) continue;
    const int q = q0 + t % cq, ro = t / cq;
    auto st = init(q);
    int r = r_begin + ro;
    for (; r + (VMTL_RIF - 1) * rpt < r_end; r += VMTL_RIF * rpt) {
      decltype(load(r, q)) l[VMTL_RIF];
#pragma unroll
      for (int u = 0; u < VMTL_RIF; ++u) l[u] = load(r + u * rpt, q);
#pragma unroll
      for (int u = 0; u < VMTL_RIF; ++u) body(l[u], r + u * rpt, q, st);
    }
    for (; r < r_end; r += rpt) {
      auto l0 = load(r, q);
      body(l0, r, q, st);
    }
  }
}

static inline int sweep_blocks(long long M, int Cs) {
  // ~16 float4 per thread per block keeps enough loads in flight without starving the grid
  long long per_blk = (long long)RED_THREADS * 16 / (Cs >> 2 > 0 ? (Cs >> 2) : 1);
  if (per_blk < 1) per_blk = 1;
  long long nb = (M + per_blk - 1) / per_blk;
  if (nb > 8192) nb = 8192;
  if (nb < 1) nb = 1;
  return (int)nb;
}

// Chan's parallel merge of (count, mean, M2) per channel quad: (n, mean, m2) <- merged with (nb, mb, m2b)
__device__ __forceinline__ void chan_merge(float& n, f32x4& mean, f32x4& m2, float nb, f32x4 mb, f32x4 m2b) {
  if (nb <= 0.f) return;
  const float nt = n + nb;
  const f32x4 d = mb - mean;
  mean += d * (nb / nt);
  m2 += m2b + d * d * (n * nb / nt);
  n = nt;
}
