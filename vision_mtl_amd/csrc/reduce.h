// Deterministic two-stage per-channel (column) reduction skeleton shared by the
// BatchNorm, bias-gradient, depthwise-weight-gradient and stitch-gradient kernels.
#pragma once
#include "common.h"

#define RED_THREADS 256
#define RED_MAX_BLOCKS 1024

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

static inline int red_blocks(int M) {
  int nb = cdiv(M, 64);
  if (nb > RED_MAX_BLOCKS) nb = RED_MAX_BLOCKS;
  if (nb < 1) nb = 1;
  return nb;
}

