// Weight (un)packing between the torch parameter layouts the drop-in boundary keeps
// (Conv2d: (Cout,Cin,KH,KW); ConvTranspose2d: (Cin,Cout,2,2); depthwise: (C,1,K,K))
// and the [row][tap*Cs + c] matrices the implicit-GEMM kernels contract against,
// plus the fused Adam update over the flat parameter arena.
//
//   packed[(r1*R0 + r0)][t'*Cs + c] = src[r1*sr1 + r0*sr0 + t*st + c*sc],   c < C, else 0
//   t' = flip ? T-1-t : t
//
//   conv fwd   : R1=1 R0=Cout T=KH*KW C=Cin  sr0=Cin*KK st=1 sc=KK        flip=0
//   conv dgrad : R1=1 R0=Cin  T=KK    C=Cout sr0=KK     st=1 sc=Cin*KK    flip=1
//   convT fwd  : R1=4 R0=Cout T=1     C=Cin  sr1=1 sr0=4 st=0 sc=Cout*4   flip=0
//   convT bwd  : R1=1 R0=Cin  T=4     C=Cout sr0=Cout*4 st=1 sc=4         flip=0
//   depthwise  : R1=1 R0=1    T=KK    C=C    st=1 sc=KK                   flip=0
//
// Adam follows torch.optim.Adam (reference vision_mtl/training_lit.py:51,87).
#include "common.h"

// scale (nullable): a per-INPUT-channel factor folded into the operand - the cross-stitch scale of the tensor the conv
// reads (reference models/cross_stitch_model.py:32-37: y[a] = w[a,a,(c)] * x[a], always followed by a dense conv in the
// CSNet walk, :108-142): conv(W, s*x) = conv(W*s, x).  smode 1: the packed column channel c is the input channel
// (forward layout), smode 2: the packed row r0 is (data-gradient layout); element scale[ch * sstride] (sstride 0: one
// scalar for the layer).
__global__ __launch_bounds__(256) void pack_kernel(const float* __restrict__ src, float* __restrict__ dst, int R1,
                                                   int R0, int T, int C, int Cs, long long sr1, long long sr0,
                                                   long long st, long long sc, int flip, long long total,
                                                   const float* __restrict__ scale, int sstride, int smode) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cs);
    long long rest = i / Cs;
    const int tp = (int)(rest % T);
    rest /= T;
    const int r0 = (int)(rest % R0);
    const int r1 = (int)(rest / R0);
    const int t = flip ? T - 1 - tp : tp;
    float v = c < C ? src[r1 * sr1 + r0 * sr0 + t * st + c * sc] : 0.f;
    if (smode != 0 && c < C) v *= scale[(size_t)(smode == 1 ? c : r0) * sstride];
    dst[i] = v;
  }
}

// ---- batched form: one launch packs MANY weight tensors (every conv of the model, forward and
// data-gradient layouts) from a descriptor table in device memory.  A step then pays one launch
// instead of ~135 five-microsecond ones.
struct PackDesc {
  const float* src;
  float* dst;
  long long sr1, sr0, st, sc;
  long long start;  // first flat work index of this descriptor
  const float* scale;  // see pack_kernel (null: none)
  int R1, R0, T, C, Cs, flip;
  int sstride, smode;
};

// One thread per packed element.  The element -> descriptor search runs over a copy of the table's `start` column in LDS
// (it ran over global memory: 7 dependent loads per element, and the index arithmetic was 64-bit - the launch took 125 us
// for ~3 M elements at the head of every step, in front of the first conv); offsets inside one operand fit 32 bits.
#define PACK_MAXD 1024
__global__ __launch_bounds__(256) void pack_batch_kernel(const PackDesc* __restrict__ descs, int n, long long total) {
  __shared__ long long starts[PACK_MAXD];
  const bool in_lds = n <= PACK_MAXD;
  if (in_lds) {
    for (int k = threadIdx.x; k < n; k += 256) starts[k] = descs[k].start;
    __syncthreads();
  }
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    int lo = 0, hi = n - 1;  // last descriptor with start <= i
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if ((in_lds ? starts[mid] : descs[mid].start) <= i) lo = mid;
      else hi = mid - 1;
    }
    const PackDesc d = descs[lo];
    const unsigned j = (unsigned)(i - d.start);  // < R1*R0*T*Cs < 2^31 (host check)
    const unsigned Cs = (unsigned)d.Cs, T = (unsigned)d.T, R0 = (unsigned)d.R0;
    unsigned rest = j / Cs;
    const int c = (int)(j - rest * Cs);
    const unsigned q1 = rest / T;
    const int tp = (int)(rest - q1 * T);
    const unsigned r1 = q1 / R0;
    const int r0 = (int)(q1 - r1 * R0);
    const int t = d.flip ? d.T - 1 - tp : tp;
    float v = c < d.C ? d.src[(long long)r1 * d.sr1 + r0 * d.sr0 + t * d.st + c * d.sc] : 0.f;
    if (d.smode != 0 && c < d.C) v *= d.scale[(size_t)(d.smode == 1 ? c : r0) * d.sstride];
    d.dst[j] = v;
  }
}

extern "C" int vmtl_pack_desc_bytes(void) { return (int)sizeof(PackDesc); }

// descs: device array of n descriptors laid out as struct PackDesc (see vmtl_pack_desc_bytes);
// total = sum of R1*R0*T*Cs over the table.
extern "C" int vmtl_pack_weights_batch(const void* descs, int n, long long total, void* stream) {
  VMTL_ENTER();
  if (!descs || n <= 0 || total <= 0) return VMTL_ERR_ARG;
  long long nb = cdivll(total, 256);
  if (nb > 8192) nb = 8192;
  hipLaunchKernelGGL(pack_batch_kernel, dim3((int)nb), dim3(256), 0, (hipStream_t)stream,
                     (const PackDesc*)descs, n, total);
  return vmtl_check_launch();
}

// writes only channels [0, C) of every `group`-wide tap group (dst may point at a channel offset):
// lets two weight tensors share one packed operand (fused heads).
__global__ __launch_bounds__(256) void pack_slice_kernel(const float* __restrict__ src, float* __restrict__ dst, int R0,
                                                         int T, int C, int group, long long sr0, long long st,
                                                         long long sc, int flip, long long total) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    long long rest = i / C;
    const int tp = (int)(rest % T);
    const int r0 = (int)(rest / T);
    const int t = flip ? T - 1 - tp : tp;
    dst[((size_t)r0 * T + tp) * group + c] = src[r0 * sr0 + t * st + c * sc];
  }
}

// inverse: grad[torch index] = sum over slabs of packed[slab][...].  A workgroup handles 64
// consecutive torch-layout elements x 4 slab groups: lane -> element (consecutive lanes read
// consecutive packed floats when C is the fast axis), wave -> slabs z = wave, wave+4, ...; the four
// partial sums are added in wave order through LDS, so the split-K reduction of the weight gradient
// is deterministic.
__global__ __launch_bounds__(256) void unpack_kernel(const float* __restrict__ packed, float* __restrict__ grad,
                                                     int R1, int R0, int T, int C, int Cs, long long sr1,
                                                     long long sr0, long long st, long long sc, int flip,
                                                     int nslabs, long long slab_stride, long long total) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, zg = threadIdx.x >> 6;
  for (long long base = (long long)blockIdx.x * 64; base < total; base += (long long)gridDim.x * 64) {
    const long long i = base + lane;
    float s = 0.f;
    long long dst = 0;
    if (i < total) {
      const int c = (int)(i % C);
      long long rest = i / C;
      const int t = (int)(rest % T);
      rest /= T;
      const int r0 = (int)(rest % R0);
      const int r1 = (int)(rest / R0);
      const int tp = flip ? T - 1 - t : t;
      dst = r1 * sr1 + r0 * sr0 + t * st + c * sc;
      const float* src = packed + ((size_t)(r1 * R0 + r0) * T + tp) * Cs + c;
      int z = zg;
      for (; z + 12 < nslabs; z += 16) {  // 4 independent loads in flight
        const float a0 = src[(size_t)z * slab_stride], a1 = src[(size_t)(z + 4) * slab_stride];
        const float a2 = src[(size_t)(z + 8) * slab_stride], a3 = src[(size_t)(z + 12) * slab_stride];
        s += a0;
        s += a1;
        s += a2;
        s += a3;
      }
      for (; z < nslabs; z += 4) s += src[(size_t)z * slab_stride];
    }
    red[zg][lane] = s;
    __syncthreads();
    if (zg == 0 && i < total) grad[dst] = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
    __syncthreads();
  }
}

static inline int pk_grid(long long total) {
  long long nb = cdivll(total, 256);
  if (nb > 4096) nb = 4096;
  if (nb < 1) nb = 1;
  return (int)nb;
}

extern "C" int vmtl_pack_weights(const float* src, float* dst, int R1, int R0, int T, int C, int Cs, long long sr1,
                                 long long sr0, long long st, long long sc, int flip, void* stream) {
  VMTL_ENTER();
  if (!src || !dst || R1 <= 0 || R0 <= 0 || T <= 0 || C <= 0 || C > Cs) return VMTL_ERR_ARG;
  const long long total = (long long)R1 * R0 * T * Cs;
  hipLaunchKernelGGL(pack_kernel, dim3(pk_grid(total)), dim3(256), 0, (hipStream_t)stream, src, dst, R1, R0, T, C, Cs,
                     sr1, sr0, st, sc, flip, total, (const float*)nullptr, 0, 0);
  return vmtl_check_launch();
}

// vmtl_pack_weights with the cross-stitch scale of the conv's input folded in (see pack_kernel): smode 1 / 2
extern "C" int vmtl_pack_weights_scaled(const float* src, float* dst, int R1, int R0, int T, int C, int Cs, long long sr1,
                                        long long sr0, long long st, long long sc, int flip, const float* scale,
                                        int sstride, int smode, void* stream) {
  VMTL_ENTER();
  if (!src || !dst || !scale || R1 <= 0 || R0 <= 0 || T <= 0 || C <= 0 || C > Cs || (smode != 1 && smode != 2) || sstride < 0)
    return VMTL_ERR_ARG;
  const long long total = (long long)R1 * R0 * T * Cs;
  hipLaunchKernelGGL(pack_kernel, dim3(pk_grid(total)), dim3(256), 0, (hipStream_t)stream, src, dst, R1, R0, T, C, Cs,
                     sr1, sr0, st, sc, flip, total, scale, sstride, smode);
  return vmtl_check_launch();
}

extern "C" int vmtl_pack_weights_slice(const float* src, float* dst, int R0, int T, int C, int group, long long sr0,
                                       long long st, long long sc, int flip, void* stream) {
  VMTL_ENTER();
  if (!src || !dst || R0 <= 0 || T <= 0 || C <= 0 || C > group) return VMTL_ERR_ARG;
  const long long total = (long long)R0 * T * C;
  hipLaunchKernelGGL(pack_slice_kernel, dim3(pk_grid(total)), dim3(256), 0, (hipStream_t)stream, src, dst, R0, T, C,
                     group, sr0, st, sc, flip, total);
  return vmtl_check_launch();
}

extern "C" int vmtl_unpack_weights(const float* packed, float* grad, int R1, int R0, int T, int C, int Cs,
                                   long long sr1, long long sr0, long long st, long long sc, int flip, int nslabs,
                                   long long slab_stride, void* stream) {
  VMTL_ENTER();
  if (!packed || !grad || R1 <= 0 || R0 <= 0 || T <= 0 || C <= 0 || C > Cs || nslabs <= 0) return VMTL_ERR_ARG;
  if (slab_stride <= 0) slab_stride = (long long)R1 * R0 * T * Cs;  // slabs hold exactly these rows
  const long long total = (long long)R1 * R0 * T * C;
  long long nb = cdivll(total, 64);
  if (nb > 8192) nb = 8192;
  hipLaunchKernelGGL(unpack_kernel, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, packed, grad, R1, R0, T, C, Cs,
                     sr1, sr0, st, sc, flip, nslabs, slab_stride, total);
  return vmtl_check_launch();
}

// ---- weight gradient of a conv whose input carried a folded cross-stitch scale s (W' = W * s[ci] was the operand):
// the slabs hold dL/dW'.  dL/dW[co][ci][t] = dL/dW' * s[ci];  dL/ds[ci] = sum_{co,t} dL/dW'[co][ci][t] * W[co][ci][t]
// (the stitch layer's weight gradient, reference models/cross_stitch_model.py:32-37, without a pass over activations).
// Step 1 (this kernel, the slab sum of unpack_kernel): grad = g' * s and prod = g' * W in the torch layout (Cout, Cin, T).
__global__ __launch_bounds__(256) void unpack_stitch_kernel(const float* __restrict__ packed, float* __restrict__ grad,
                                                            float* __restrict__ prod, const float* __restrict__ w,
                                                            const float* __restrict__ scale, int sstride, int R0, int T,
                                                            int C, int Cs, int nslabs, long long slab_stride,
                                                            long long total) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, zg = threadIdx.x >> 6;
  for (long long base = (long long)blockIdx.x * 64; base < total; base += (long long)gridDim.x * 64) {
    const long long i = base + lane;  // torch index (r0 * C + c) * T + t
    float s = 0.f;
    int c = 0;
    if (i < total) {
      const int t = (int)(i % T);
      long long rest = i / T;
      c = (int)(rest % C);
      const int r0 = (int)(rest / C);
      const float* src = packed + ((size_t)r0 * T + t) * Cs + c;
      for (int z = zg; z < nslabs; z += 4) s += src[(size_t)z * slab_stride];
    }
    red[zg][lane] = s;
    __syncthreads();
    if (zg == 0 && i < total) {
      const float g = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
      grad[i] = g * scale[(size_t)c * sstride];
      prod[i] = g * w[i];
    }
    __syncthreads();
  }
}

// Step 2: ds[ci] = sum_{co,t} prod[co][ci][t] (one workgroup per input channel, fp64 accumulation);
// reduce_all: one scalar = the sum over channels too (layer-wise stitching), through `tmp` [C]
__global__ __launch_bounds__(256) void stitch_wsum_kernel(const float* __restrict__ prod, float* __restrict__ out, int R0,
                                                          int C, int T) {
  __shared__ double sh[4];
  const int c = blockIdx.x;
  double a = 0.0;
  for (int j = threadIdx.x; j < R0 * T; j += 256) {
    const int r0 = j / T, t = j - r0 * T;
    a += (double)prod[((size_t)r0 * C + c) * T + t];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) out[c] = (float)(sh[0] + sh[1] + sh[2] + sh[3]);
}

__global__ __launch_bounds__(256) void sum_vec_kernel(const float* __restrict__ v, int n, float* __restrict__ out) {
  __shared__ double sh[4];
  double a = 0.0;
  for (int j = threadIdx.x; j < n; j += 256) a += (double)v[j];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (float)(sh[0] + sh[1] + sh[2] + sh[3]);
}

// packed: nslabs slabs [R0][T][Cs] (forward packing of a (R0 = Cout, C = Cin, T taps) conv weight); grad: dL/dW in the
// torch layout; w: the weight; scale / sstride: the folded stitch factors; ds: their gradient (C entries, or ONE when
// reduce_all); work: R0*C*T + C floats of scratch.
extern "C" int vmtl_unpack_weights_stitch(const float* packed, float* grad, const float* w, const float* scale, int sstride,
                                          float* ds, float* work, int R0, int T, int C, int Cs, int nslabs,
                                          long long slab_stride, int reduce_all, void* stream) {
  VMTL_ENTER();
  if (!packed || !grad || !w || !scale || !ds || !work || R0 <= 0 || T <= 0 || C <= 0 || C > Cs || nslabs <= 0 || sstride < 0)
    return VMTL_ERR_ARG;
  if (slab_stride <= 0) slab_stride = (long long)R0 * T * Cs;
  const long long total = (long long)R0 * C * T;
  long long nb = cdivll(total, 64);
  if (nb > 8192) nb = 8192;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(unpack_stitch_kernel, dim3((int)nb), dim3(256), 0, st, packed, grad, work, w, scale, sstride, R0, T, C,
                     Cs, nslabs, slab_stride, total);
  float* per_c = reduce_all ? work + total : ds;
  hipLaunchKernelGGL(stitch_wsum_kernel, dim3(C), dim3(256), 0, st, work, per_c, R0, C, T);
  if (reduce_all) hipLaunchKernelGGL(sum_vec_kernel, dim3(1), dim3(256), 0, st, per_c, C, ds);
  return vmtl_check_launch();
}

// ------------------------------------------------------------------ up2 (upsample+concat+conv3x3) operands
// Tap bookkeeping (see conv_igemm.hip, UP2): with output-row parity a and low-res tap ty, the 3x3 taps
// dh that land on that low-res row are R(0,0)={0}, R(0,1)={1,2}, R(1,0)={0,1}, R(1,1)={2}.  For the
// backward 4x4/stride-2/pad-1 convolution over dY, tap kh pairs with dh in {2-kh, 3-kh} (clipped to 0..2).
__device__ __forceinline__ void up2_R(int a, int t, int& lo, int& hi) {
  if (a == 0) { lo = t == 0 ? 0 : 1; hi = t == 0 ? 0 : 2; }
  else { lo = t == 0 ? 0 : 2; hi = t == 0 ? 1 : 2; }
}

// dst[phase][n][kk]: kk < 4*C0s -> (ty,tx,c) summed taps; kk >= 4*C0s -> (tap, cj) plain copy of the skip channels
__global__ __launch_bounds__(256) void pack_up2_fwd_kernel(const float* __restrict__ w, float* __restrict__ dst,
                                                           int Cout, int C0, int C0s, int C1, int C1s,
                                                           long long total) {
  const int Cin = C0 + C1, Ktot = 4 * C0s + 9 * C1s;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int kk = (int)(i % Ktot);
    const long long rest = i / Ktot;
    const int n = (int)(rest % Cout), phase = (int)(rest / Cout);
    const int a = phase >> 1, b = phase & 1;
    float v = 0.f;
    const float* wn = w + (size_t)n * Cin * 9;
    if (kk < 4 * C0s) {
      const int t2 = kk / C0s, c = kk - t2 * C0s;
      if (c < C0) {
        int hl, hh, wl, wh;
        up2_R(a, t2 >> 1, hl, hh);
        up2_R(b, t2 & 1, wl, wh);
        for (int dh = hl; dh <= hh; ++dh)
          for (int dw = wl; dw <= wh; ++dw) v += wn[c * 9 + dh * 3 + dw];
      }
    } else {
      const int k2 = kk - 4 * C0s;
      const int tap = k2 / C1s, c = k2 - tap * C1s;
      if (c < C1) v = wn[(C0 + c) * 9 + tap];
    }
    dst[i] = v;
  }
}

// data-gradient operand of the low-res input: dst[c][(kh*4+kw)*Cos + n] = sum of W[n][c][dh][dw] over the
// (dh, dw) paired with (kh, kw); rows c < C0 (4x4 / stride 2 / pad 1 convolution over dY)
__global__ __launch_bounds__(256) void pack_up2_dgrad_kernel(const float* __restrict__ w, float* __restrict__ dst,
                                                             int Cout, int Cos, int C0, int Cin, long long total) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int n = (int)(i % Cos);
    long long rest = i / Cos;
    const int pos = (int)(rest % 16), c = (int)(rest / 16);
    const int kh = pos >> 2, kw = pos & 3;
    float v = 0.f;
    if (n < Cout)
      for (int dh = max(0, 2 - kh); dh <= min(2, 3 - kh); ++dh)
        for (int dw = max(0, 2 - kw); dw <= min(2, 3 - kw); ++dw) v += w[((size_t)n * Cin + c) * 9 + dh * 3 + dw];
    dst[i] = v;
  }
}

// folds the slabs of the low-res weight gradient G[z][c][(kh*4+kw)*Cos + n] back onto the 3x3 taps:
// dW[n][c][dh][dw] = sum_z sum_{kh in {2-dh,3-dh}} sum_{kw in {2-dw,3-dw}} G[z][c][kh][kw][n]   (c < C0)
// A workgroup handles 64 consecutive elements x 4 slab groups (wave w sums slabs w, w+4, ...; the four partial sums
// are added in wave order through LDS: deterministic), two slabs' loads in flight per thread.  One thread per element
// walking all slabs left the narrow full-resolution layers (20 K elements, 64+ slabs) with 78 workgroups of 256 serial
// dependent loads: 96 us.
__global__ __launch_bounds__(256) void unpack_up2_kernel(const float* __restrict__ slabs, float* __restrict__ grad,
                                                         int Cout, int Cos, int C0, int Cin, int nslabs,
                                                         long long slab_stride, long long total) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, zg = threadIdx.x >> 6;
  for (long long base = (long long)blockIdx.x * 64; base < total; base += (long long)gridDim.x * 64) {
    const long long i = base + lane;
    float s = 0.f;
    long long dst = 0;
    if (i < total) {
      const int n = (int)(i % Cout);
      const long long rest = i / Cout;
      const int tap = (int)(rest % 9), c = (int)(rest / 9);
      const int dh = tap / 3, dw = tap - dh * 3;
      dst = ((long long)n * Cin + c) * 9 + tap;
      const float* g0 = slabs + (size_t)c * 16 * Cos + n;
      const int o00 = ((2 - dh) * 4 + (2 - dw)) * Cos, o01 = ((2 - dh) * 4 + (3 - dw)) * Cos;
      const int o10 = ((3 - dh) * 4 + (2 - dw)) * Cos, o11 = ((3 - dh) * 4 + (3 - dw)) * Cos;
      int z = zg;
      for (; z + 4 < nslabs; z += 8) {
        const float* ga = g0 + (size_t)z * slab_stride;
        const float* gb = g0 + (size_t)(z + 4) * slab_stride;
        const float a0 = ga[o00], a1 = ga[o01], a2 = ga[o10], a3 = ga[o11];
        const float b0 = gb[o00], b1 = gb[o01], b2 = gb[o10], b3 = gb[o11];
        s += (a0 + a1) + (a2 + a3);
        s += (b0 + b1) + (b2 + b3);
      }
      for (; z < nslabs; z += 4) {
        const float* ga = g0 + (size_t)z * slab_stride;
        s += (ga[o00] + ga[o01]) + (ga[o10] + ga[o11]);
      }
    }
    red[zg][lane] = s;
    __syncthreads();
    if (zg == 0 && i < total) grad[dst] = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
    __syncthreads();
  }
}

extern "C" int vmtl_pack_up2_fwd(const float* w, float* dst, int Cout, int C0, int C0s, int C1, int C1s, void* stream) {
  VMTL_ENTER();
  if (!w || !dst || Cout <= 0 || C0 <= 0 || C0 > C0s || C1 < 0 || C1 > C1s) return VMTL_ERR_ARG;
  const long long total = 4ll * Cout * (4 * C0s + 9 * C1s);
  hipLaunchKernelGGL(pack_up2_fwd_kernel, dim3(pk_grid(total)), dim3(256), 0, (hipStream_t)stream, w, dst, Cout, C0,
                     C0s, C1, C1s, total);
  return vmtl_check_launch();
}

extern "C" int vmtl_pack_up2_dgrad(const float* w, float* dst, int Cout, int Cos, int C0, int Cin, void* stream) {
  VMTL_ENTER();
  if (!w || !dst || Cout <= 0 || Cout > Cos || C0 <= 0 || C0 > Cin) return VMTL_ERR_ARG;
  const long long total = (long long)C0 * 16 * Cos;
  hipLaunchKernelGGL(pack_up2_dgrad_kernel, dim3(pk_grid(total)), dim3(256), 0, (hipStream_t)stream, w, dst, Cout, Cos,
                     C0, Cin, total);
  return vmtl_check_launch();
}

extern "C" int vmtl_unpack_up2(const float* slabs, float* grad, int Cout, int Cos, int C0, int Cin, int nslabs,
                               void* stream) {
  VMTL_ENTER();
  if (!slabs || !grad || Cout <= 0 || Cout > Cos || C0 <= 0 || C0 > Cin || nslabs <= 0) return VMTL_ERR_ARG;
  const long long total = (long long)C0 * 9 * Cout;
  long long nb = cdivll(total, 64);
  if (nb > 8192) nb = 8192;
  hipLaunchKernelGGL(unpack_up2_kernel, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, slabs, grad, Cout,
                     Cos, C0, Cin, nslabs, (long long)C0 * 16 * Cos, total);
  return vmtl_check_launch();
}

// ------------------------------------------------------------------ fused Adam over a flat arena
// step_ptr[0] holds the (already incremented) step count as a float so the update can sit in a hipGraph.
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v,
                                                   const float* __restrict__ step_ptr, float lr, float b1, float b2,
                                                   float eps, float wd, float gscale, long long n) {
  const float step = step_ptr[0];
  const float bc1 = 1.f - powf(b1, step);
  const float bc2 = 1.f - powf(b2, step);
  const float step_size = lr / bc1;
  const float inv_sqrt_bc2 = 1.f / sqrtf(bc2);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float gi = g[i] * gscale;
    const float pi = p[i];
    if (wd != 0.f) gi += wd * pi;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
    p[i] = pi - step_size * (mi / denom);
  }
}

extern "C" int vmtl_adam_step(float* p, const float* g, float* m, float* v, const float* step_ptr, float lr, float b1,
                              float b2, float eps, float weight_decay, float grad_scale, long long n, void* stream) {
  VMTL_ENTER();
  if (!p || !g || !m || !v || !step_ptr || n <= 0) return VMTL_ERR_ARG;
  hipLaunchKernelGGL(adam_kernel, dim3(pk_grid(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, step_ptr, lr, b1,
                     b2, eps, weight_decay, grad_scale, n);
  return vmtl_check_launch();
}
