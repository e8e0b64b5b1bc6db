// HBM-bound gather / elementwise kernels of the dense-prediction path (NHWC fp32).
//
//   concat2      : channel concat of two sources, each optionally nearest-x2 upsampled and/or
//                  zero-padded into the destination canvas.  Covers
//                    smp DecoderBlock (interpolate nearest x2 + cat[x, skip])       [3P]
//                    reference utils/model_utils.py:46-58 (pad + cat[x2, x1])
//                    reference models/mtan_model.py:65,152 (cat)
//                    reference models/cross_stitch_model.py:126-134
//   maxpool2     : reference models/mtan_model.py:49,81,364,388 (MaxPool2d(2))
//   bilinear_up2 : reference models/mtan_model.py:125,143-144 (Upsample x2, align_corners=True)
//   spatial_mean / channel_scale : timm SqueezeExcite squeeze + excite multiply    [3P]
//   stitch       : reference models/cross_stitch_model.py:32-37 (diagonal einsum)
//   sigmoid / argmax : reference lit_module.py:133-144 (postprocess_raw_out)
//   nchw<->nhwc  : boundary layout change (reference tensors are NCHW)
#include "common.h"

static inline int ew_grid(long long total, int per_block = 256) {
  long long nb = cdivll(total, per_block);
  if (nb > 8192) nb = 8192;
  if (nb < 1) nb = 1;
  return (int)nb;
}

#define GRID_STRIDE(i, total)                                                           \
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (total);     \
       i += (long long)gridDim.x * blockDim.x)

// (channel quad, column, row, image) of flat float4 index i over a [B][H][W][CQ] map.  The decode is the bulk of the VALU work of
// these streaming kernels: with 64-bit operands each division is ~100 instructions (bilinear x2 ran at 2.8 TB/s); every map of
// this library has fewer than 2^31 float4 elements, so the 32-bit form is the one that runs (uniform switch).
struct Pix4 { int q, w, h, b; };
__device__ __forceinline__ Pix4 decode_pix(long long i, int CQ, int W, int H, bool small) {
  Pix4 p;
  if (small) {
    const unsigned u = (unsigned)i, pix = u / (unsigned)CQ;
    p.q = (int)(u - pix * (unsigned)CQ);
    const unsigned t = pix / (unsigned)W;
    p.w = (int)(pix - t * (unsigned)W);
    const unsigned b = t / (unsigned)H;
    p.h = (int)(t - b * (unsigned)H);
    p.b = (int)b;
  } else {
    p.q = (int)(i % CQ);
    const long long pix = i / CQ;
    p.w = (int)(pix % W);
    p.h = (int)((pix / W) % H);
    p.b = (int)(pix / ((long long)W * H));
  }
  return p;
}

// ------------------------------------------------------------------ concat2
struct CatSrc {
  const float* p;  // [B][Hs][Ws][Cs]
  int Hs, Ws, C, Cs;
  int up;          // 1 or 2 (nearest)
  int oh, ow;      // placement offset of the (upsampled) source inside the destination
};

// one thread per output float4 (16-byte store); a quad that lies inside one source at a 16-byte
// aligned channel offset is fetched with one 16-byte load, otherwise element by element
__global__ __launch_bounds__(256) void concat2_kernel(CatSrc s0, CatSrc s1, float* __restrict__ y, int B, int H,
                                                      int W, int Cd, long long total4) {
  const int CQ = Cd >> 2;
  GRID_STRIDE(i, total4) {
    const int q = (int)(i % CQ);
    const unsigned pix = (unsigned)(i / CQ);
    const unsigned w = pix % (unsigned)W, hb = pix / (unsigned)W;
    const unsigned h = hb % (unsigned)H, b = hb / (unsigned)H;
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    const int c0 = q * 4;
    // pixel of each source that feeds (h, w), or -1
    long long p0 = -1, p1 = -1;
    {
      const int hh = (int)h - s0.oh, ww = (int)w - s0.ow;
      if (hh >= 0 && ww >= 0 && hh < s0.Hs * s0.up && ww < s0.Ws * s0.up)
        p0 = ((long long)b * s0.Hs + hh / s0.up) * s0.Ws + ww / s0.up;
    }
    if (s1.C > 0) {
      const int hh = (int)h - s1.oh, ww = (int)w - s1.ow;
      if (hh >= 0 && ww >= 0 && hh < s1.Hs * s1.up && ww < s1.Ws * s1.up)
        p1 = ((long long)b * s1.Hs + hh / s1.up) * s1.Ws + ww / s1.up;
    }
    if (c0 + 3 < s0.C) {
      if (p0 >= 0) o = *reinterpret_cast<const f32x4*>(s0.p + (size_t)p0 * s0.Cs + c0);
    } else if (c0 >= s0.C && ((c0 - s0.C) & 3) == 0 && c0 - s0.C + 3 < s1.C) {
      if (p1 >= 0) o = *reinterpret_cast<const f32x4*>(s1.p + (size_t)p1 * s1.Cs + (c0 - s0.C));
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = c0 + e;
        float v = 0.f;
        if (c < s0.C) {
          if (p0 >= 0) v = s0.p[(size_t)p0 * s0.Cs + c];
        } else if (c < s0.C + s1.C) {
          if (p1 >= 0) v = s1.p[(size_t)p1 * s1.Cs + (c - s0.C)];
        }
        o[e] = v;
      }
    }
    reinterpret_cast<f32x4*>(y)[i] = o;
  }
}

// gradient of ONE source: gathers (and 2x2-sums for an upsampled source) its channel slice;
// one thread per source float4, 16-byte loads when the slice starts at a multiple of 4 channels
__global__ __launch_bounds__(256) void concat2_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int B,
                                                          int H, int W, int Cd, int c_off, int Hs, int Ws, int C,
                                                          int Cs, int up, int oh, int ow, long long total4) {
  const int CQ = Cs >> 2;
  const bool aligned = (c_off & 3) == 0;
  GRID_STRIDE(i, total4) {
    const int q = (int)(i % CQ);
    const unsigned pix = (unsigned)(i / CQ);
    const unsigned ws = pix % (unsigned)Ws, hb = pix / (unsigned)Ws;
    const unsigned hs = hb % (unsigned)Hs, b = hb / (unsigned)Hs;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int c0 = q * 4;
    if (c0 < C)
      for (int u = 0; u < up; ++u)
        for (int t = 0; t < up; ++t) {
          const int h = (int)hs * up + u + oh, w = (int)ws * up + t + ow;
          if (h < 0 || w < 0 || h >= H || w >= W) continue;
          const float* src = dy + ((size_t)((long long)b * H + h) * W + w) * Cd + c_off + c0;
          if (aligned && c0 + 3 < C) {
            acc += *reinterpret_cast<const f32x4*>(src);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (c0 + e < C) acc[e] += src[e];
          }
        }
    reinterpret_cast<f32x4*>(dx)[i] = acc;
  }
}

extern "C" int vmtl_concat2(const float* a, int Ha, int Wa, int Ca, int Csa, int upa, int oha, int owa,
                            const float* b, int Hb, int Wb, int Cb, int Csb, int upb, int ohb, int owb, float* y,
                            int B, int H, int W, int Cd, void* stream) {
  VMTL_ENTER();
  if (!a || !y || B <= 0 || Ca + Cb > Cd || (Cd & 3) || (Csa & 3) || (Cb && (Csb & 3)) || (upa != 1 && upa != 2))
    return VMTL_ERR_ARG;
  if ((long long)B * H * W > 0x7fffffffLL) return VMTL_ERR_ARG;
  if (b == nullptr && Cb != 0) return VMTL_ERR_ARG;
  if (Cb && upb != 1 && upb != 2) return VMTL_ERR_ARG;
  CatSrc s0{a, Ha, Wa, Ca, Csa, upa, oha, owa};
  CatSrc s1{b, Hb, Wb, Cb, Csb, Cb ? upb : 1, ohb, owb};
  const long long total4 = (long long)B * H * W * (Cd >> 2);
  hipLaunchKernelGGL(concat2_kernel, dim3(ew_grid(total4)), dim3(256), 0, (hipStream_t)stream, s0, s1, y, B, H, W, Cd,
                     total4);
  return vmtl_check_launch();
}

extern "C" int vmtl_concat2_bwd(const float* dy, float* dx, int B, int H, int W, int Cd, int c_off, int Hs, int Ws,
                                int C, int Cs, int up, int oh, int ow, void* stream) {
  VMTL_ENTER();
  if (!dy || !dx || c_off < 0 || c_off + C > Cd || (Cs & 3) || (Cd & 3) || (up != 1 && up != 2)) return VMTL_ERR_ARG;
  if ((long long)B * Hs * Ws > 0x7fffffffLL) return VMTL_ERR_ARG;
  const long long total4 = (long long)B * Hs * Ws * (Cs >> 2);
  hipLaunchKernelGGL(concat2_bwd_kernel, dim3(ew_grid(total4)), dim3(256), 0, (hipStream_t)stream, dy, dx, B, H, W, Cd,
                     c_off, Hs, Ws, C, Cs, up, oh, ow, total4);
  return vmtl_check_launch();
}

// ------------------------------------------------------------------ maxpool 2x2 / stride 2
// First maximum in row-major window order wins (strict >), as ATen's CPU kernel records it.
__global__ __launch_bounds__(256) void maxpool2_kernel(const float* __restrict__ x, float* __restrict__ y, int B,
                                                       int H, int W, int Cs, long long total4) {
  const int CQ = Cs >> 2, Ho = H >> 1, Wo = W >> 1;
  GRID_STRIDE(i, total4) {
    const Pix4 px = decode_pix(i, CQ, Wo, Ho, total4 < (1ll << 31));
    const int q = px.q, wo = px.w, ho = px.h, b = px.b;
    const float* base = x + ((size_t)(b * H + 2 * ho) * W + 2 * wo) * Cs + (size_t)q * 4;
    f32x4 m = *reinterpret_cast<const f32x4*>(base);
    const f32x4 v1 = *reinterpret_cast<const f32x4*>(base + Cs);
    const f32x4 v2 = *reinterpret_cast<const f32x4*>(base + (size_t)W * Cs);
    const f32x4 v3 = *reinterpret_cast<const f32x4*>(base + (size_t)W * Cs + Cs);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (v1[e] > m[e] || v1[e] != v1[e]) m[e] = v1[e];
      if (v2[e] > m[e] || v2[e] != v2[e]) m[e] = v2[e];
      if (v3[e] > m[e] || v3[e] != v3[e]) m[e] = v3[e];
    }
    reinterpret_cast<f32x4*>(y)[i] = m;
  }
}

__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                           float* __restrict__ dx, int B, int H, int W, int Cs,
                                                           long long total4) {
  // one thread per OUTPUT pixel quad: recompute the argmax, write all four input gradients
  const int CQ = Cs >> 2, Ho = H >> 1, Wo = W >> 1;
  GRID_STRIDE(i, total4) {
    const Pix4 px = decode_pix(i, CQ, Wo, Ho, total4 < (1ll << 31));
    const int q = px.q, wo = px.w, ho = px.h, b = px.b;
    const size_t o00 = ((size_t)(b * H + 2 * ho) * W + 2 * wo) * Cs + (size_t)q * 4;
    const size_t o01 = o00 + Cs, o10 = o00 + (size_t)W * Cs, o11 = o10 + Cs;
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(x + o00);
    const f32x4 v1 = *reinterpret_cast<const f32x4*>(x + o01);
    const f32x4 v2 = *reinterpret_cast<const f32x4*>(x + o10);
    const f32x4 v3 = *reinterpret_cast<const f32x4*>(x + o11);
    const f32x4 g = reinterpret_cast<const f32x4*>(dy)[i];
    f32x4 g0, g1, g2, g3;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      int am = 0;
      float m = v0[e];
      if (v1[e] > m || v1[e] != v1[e]) { m = v1[e]; am = 1; }
      if (v2[e] > m || v2[e] != v2[e]) { m = v2[e]; am = 2; }
      if (v3[e] > m || v3[e] != v3[e]) { m = v3[e]; am = 3; }
      g0[e] = am == 0 ? g[e] : 0.f;
      g1[e] = am == 1 ? g[e] : 0.f;
      g2[e] = am == 2 ? g[e] : 0.f;
      g3[e] = am == 3 ? g[e] : 0.f;
    }
    *reinterpret_cast<f32x4*>(dx + o00) = g0;
    *reinterpret_cast<f32x4*>(dx + o01) = g1;
    *reinterpret_cast<f32x4*>(dx + o10) = g2;
    *reinterpret_cast<f32x4*>(dx + o11) = g3;
  }
}

extern "C" int vmtl_maxpool2_fwd(const float* x, float* y, int B, int H, int W, int Cs, void* stream) {
  VMTL_ENTER();
  if (!x || !y || (Cs & 3) || (H & 1) || (W & 1) || H < 2 || W < 2) return VMTL_ERR_ARG;
  const long long total4 = (long long)B * (H / 2) * (W / 2) * (Cs >> 2);
  hipLaunchKernelGGL(maxpool2_kernel, dim3(ew_grid(total4)), dim3(256), 0, (hipStream_t)stream, x, y, B, H, W, Cs,
                     total4);
  return vmtl_check_launch();
}

extern "C" int vmtl_maxpool2_bwd(const float* x, const float* dy, float* dx, int B, int H, int W, int Cs,
                                 void* stream) {
  VMTL_ENTER();
  if (!x || !dy || !dx || (Cs & 3) || (H & 1) || (W & 1) || H < 2 || W < 2) return VMTL_ERR_ARG;
  const long long total4 = (long long)B * (H / 2) * (W / 2) * (Cs >> 2);
  hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(ew_grid(total4)), dim3(256), 0, (hipStream_t)stream, x, dy, dx, B, H,
                     W, Cs, total4);
  return vmtl_check_launch();
}

// ------------------------------------------------------------------ bilinear x2, align_corners=True
// src = dst * (Hin-1)/(Hout-1).  ATen computes the scale as a float division and the
// source index as scale*dst (area_pixel_compute_source_index with align_corners).
__device__ __forceinline__ void bil_coord(int o, int in_size, float scale, int& i0, int& i1, float& l1) {
  const float s = scale * (float)o;
  i0 = (int)s;
  if (i0 > in_size - 1) i0 = in_size - 1;
  i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  l1 = s - (float)i0;
}

__global__ __launch_bounds__(256) void bilinear_up2_kernel(const float* __restrict__ x, float* __restrict__ y, int B,
                                                           int H, int W, int Cs, float sh, float sw,
                                                           long long total4) {
  const int CQ = Cs >> 2, Ho = 2 * H, Wo = 2 * W;
  const bool small = total4 < (1ll << 31);
  GRID_STRIDE(i, total4) {
    const Pix4 px = decode_pix(i, CQ, Wo, Ho, small);
    const int q = px.q, wo = px.w, ho = px.h, b = px.b;
    int h0, h1, w0, w1;
    float lh, lw;
    bil_coord(ho, H, sh, h0, h1, lh);
    bil_coord(wo, W, sw, w0, w1, lw);
    const float* base = x + (size_t)b * H * W * Cs + (size_t)q * 4;
    const f32x4 v00 = *reinterpret_cast<const f32x4*>(base + ((size_t)h0 * W + w0) * Cs);
    const f32x4 v01 = *reinterpret_cast<const f32x4*>(base + ((size_t)h0 * W + w1) * Cs);
    const f32x4 v10 = *reinterpret_cast<const f32x4*>(base + ((size_t)h1 * W + w0) * Cs);
    const f32x4 v11 = *reinterpret_cast<const f32x4*>(base + ((size_t)h1 * W + w1) * Cs);
    const float h0l = 1.f - lh, w0l = 1.f - lw;
    reinterpret_cast<f32x4*>(y)[i] = h0l * (w0l * v00 + lw * v01) + lh * (w0l * v10 + lw * v11);
  }
}

// Scatter-free backward: every input pixel gathers from the (at most 4x4) output pixels whose
// interpolation footprint touches it; contributions are summed in a fixed order.
__global__ __launch_bounds__(256) void bilinear_up2_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx,
                                                               int B, int H, int W, int Cs, float sh, float sw,
                                                               long long total4) {
  const int CQ = Cs >> 2, Ho = 2 * H, Wo = 2 * W;
  const bool small = total4 < (1ll << 31);
  // reciprocal scales once per thread (the candidate ranges below carry a margin of one pixel on each side, which absorbs the
  // rounding difference to a true division; every candidate is re-tested exactly with bil_coord)
  const float ish = 1.f / fmaxf(sh, 1e-20f), isw = 1.f / fmaxf(sw, 1e-20f);
  GRID_STRIDE(i, total4) {
    const Pix4 px = decode_pix(i, CQ, W, H, small);
    const int q = px.q, w = px.w, h = px.h, b = px.b;
    // candidate output rows: those with floor(sh*ho) in {h-1, h} (clamped in float: sh may be 0)
    const float fHo = (float)(Ho - 1), fWo = (float)(Wo - 1);
    int ho_lo = h > 0 ? (int)fminf(floorf((float)(h - 1) * ish), fHo) - 1 : 0;
    int ho_hi = (int)fminf(ceilf((float)(h + 1) * ish) + 1.f, fHo);
    int wo_lo = w > 0 ? (int)fminf(floorf((float)(w - 1) * isw), fWo) - 1 : 0;
    int wo_hi = (int)fminf(ceilf((float)(w + 1) * isw) + 1.f, fWo);
    ho_lo = max(ho_lo, 0); wo_lo = max(wo_lo, 0);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const float* base = dy + (size_t)b * Ho * Wo * Cs + (size_t)q * 4;
    for (int ho = ho_lo; ho <= ho_hi; ++ho) {
      int h0, h1; float lh;
      bil_coord(ho, H, sh, h0, h1, lh);
      float wh = 0.f;
      if (h0 == h) wh += 1.f - lh;
      if (h1 == h) wh += lh;
      if (wh == 0.f) continue;
      for (int wo = wo_lo; wo <= wo_hi; ++wo) {
        int w0, w1; float lw;
        bil_coord(wo, W, sw, w0, w1, lw);
        float ww = 0.f;
        if (w0 == w) ww += 1.f - lw;
        if (w1 == w) ww += lw;
        if (ww == 0.f) continue;
        acc += (wh * ww) * *reinterpret_cast<const f32x4*>(base + ((size_t)ho * Wo + wo) * Cs);
      }
    }
    reinterpret_cast<f32x4*>(dx)[i] = acc;
  }
}

static inline float bil_scale(int in_size) {
  const int out = 2 * in_size;
  return out > 1 ? (float)(in_size - 1) / (float)(out - 1) : 0.f;
}

extern "C" int vmtl_bilinear_up2_fwd(const float* x, float* y, int B, int H, int W, int Cs, void* stream) {
  VMTL_ENTER();
  if (!x || !y || (Cs & 3) || B <= 0 || H <= 0 || W <= 0) return VMTL_ERR_ARG;
  const long long total4 = (long long)B * 4 * H * W * (Cs >> 2);
  hipLaunchKernelGGL(bilinear_up2_kernel, dim3(ew_grid(total4)), dim3(256), 0, (hipStream_t)stream, x, y, B, H, W, Cs,
                     bil_scale(H), bil_scale(W), total4);
  return vmtl_check_launch();
}

extern "C" int vmtl_bilinear_up2_bwd(const float* dy, float* dx, int B, int H, int W, int Cs, void* stream) {
  VMTL_ENTER();
  if (!dy || !dx || (Cs & 3) || B <= 0 || H <= 0 || W <= 0) return VMTL_ERR_ARG;
  const long long total4 = (long long)B * H * W * (Cs >> 2);
  hipLaunchKernelGGL(bilinear_up2_bwd_kernel, dim3(ew_grid(total4)), dim3(256), 0, (hipStream_t)stream, dy, dx, B, H,
                     W, Cs, bil_scale(H), bil_scale(W), total4);
  return vmtl_check_launch();
}

// ------------------------------------------------------------------ squeeze-excite pieces
// spatial mean: one workgroup per (image, 64-quad channel panel); rows strided over waves.
__global__ __launch_bounds__(256) void spatial_mean_kernel(const float* __restrict__ x, float* __restrict__ y, int HW,
                                                           int Cs) {
  __shared__ f32x4 red[256];
  const int CQ = Cs >> 2;
  const int b = blockIdx.y;
  const int q = blockIdx.x * 64 + (threadIdx.x & 63);
  const int ro = threadIdx.x >> 6;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (q < CQ)
    for (int r = ro; r < HW; r += 4) acc += *reinterpret_cast<const f32x4*>(x + ((size_t)b * HW + r) * Cs + (size_t)q * 4);
  red[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x < 64 && q < CQ) {
    f32x4 s = red[threadIdx.x] + red[threadIdx.x + 64] + red[threadIdx.x + 128] + red[threadIdx.x + 192];
    *reinterpret_cast<f32x4*>(y + (size_t)b * Cs + (size_t)q * 4) = s * (1.f / (float)HW);
  }
}

// mode 0: y = x * s[b][c]            (forward, and dx = dy * s)
// mode 1: y = g / HW  broadcast       (backward of the spatial mean: g is [B][Cs])
__global__ __launch_bounds__(256) void channel_bcast_kernel(const float* __restrict__ x, const float* __restrict__ s,
                                                            float* __restrict__ y, int HW, int Cs, int mode,
                                                            long long total4) {
  const int CQ = Cs >> 2;
  GRID_STRIDE(i, total4) {
    const int q = (int)(i % CQ);
    const int b = (int)(i / ((long long)CQ * HW));
    const f32x4 sv = *reinterpret_cast<const f32x4*>(s + (size_t)b * Cs + (size_t)q * 4);
    if (mode == 0) reinterpret_cast<f32x4*>(y)[i] = reinterpret_cast<const f32x4*>(x)[i] * sv;
    else reinterpret_cast<f32x4*>(y)[i] = sv * (1.f / (float)HW);
  }
}

// ds[b][c] = sum_hw dy*x
__global__ __launch_bounds__(256) void channel_scale_bwd_s_kernel(const float* __restrict__ x,
                                                                  const float* __restrict__ dy,
                                                                  float* __restrict__ ds, int HW, int Cs) {
  __shared__ f32x4 red[256];
  const int CQ = Cs >> 2;
  const int b = blockIdx.y;
  const int q = blockIdx.x * 64 + (threadIdx.x & 63);
  const int ro = threadIdx.x >> 6;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (q < CQ)
    for (int r = ro; r < HW; r += 4) {
      const size_t off = ((size_t)b * HW + r) * Cs + (size_t)q * 4;
      acc += *reinterpret_cast<const f32x4*>(x + off) * *reinterpret_cast<const f32x4*>(dy + off);
    }
  red[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x < 64 && q < CQ)
    *reinterpret_cast<f32x4*>(ds + (size_t)b * Cs + (size_t)q * 4) =
        red[threadIdx.x] + red[threadIdx.x + 64] + red[threadIdx.x + 128] + red[threadIdx.x + 192];
}

extern "C" int vmtl_spatial_mean(const float* x, float* y, int B, int HW, int Cs, void* stream) {
  VMTL_ENTER();
  if (!x || !y || (Cs & 3) || B <= 0 || HW <= 0) return VMTL_ERR_ARG;
  hipLaunchKernelGGL(spatial_mean_kernel, dim3(cdiv(Cs >> 2, 64), B), dim3(256), 0, (hipStream_t)stream, x, y, HW, Cs);
  return vmtl_check_launch();
}

extern "C" int vmtl_channel_bcast(const float* x, const float* s, float* y, int B, int HW, int Cs, int mode,
                                  void* stream) {
  VMTL_ENTER();
  if (!s || !y || (mode == 0 && !x) || (Cs & 3) || B <= 0 || HW <= 0) return VMTL_ERR_ARG;
  const long long total4 = (long long)B * HW * (Cs >> 2);
  hipLaunchKernelGGL(channel_bcast_kernel, dim3(ew_grid(total4)), dim3(256), 0, (hipStream_t)stream, x, s, y, HW, Cs,
                     mode, total4);
  return vmtl_check_launch();
}

extern "C" int vmtl_channel_scale_bwd_s(const float* x, const float* dy, float* ds, int B, int HW, int Cs,
                                        void* stream) {
  VMTL_ENTER();
  if (!x || !dy || !ds || (Cs & 3) || B <= 0 || HW <= 0) return VMTL_ERR_ARG;
  hipLaunchKernelGGL(channel_scale_bwd_s_kernel, dim3(cdiv(Cs >> 2, 64), B), dim3(256), 0, (hipStream_t)stream, x, dy,
                     ds, HW, Cs);
  return vmtl_check_launch();
}

// ------------------------------------------------------------------ cross-stitch diagonal scale
// y[m][c] = w[c*wstride] * x[m][c]  (wstride 0: one scalar for the whole layer).  In place allowed.
__global__ __launch_bounds__(256) void stitch_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                     float* __restrict__ y, int C, int Cs, int wstride,
                                                     long long total4) {
  const int CQ = Cs >> 2;
  GRID_STRIDE(i, total4) {
    const int q = (int)(i % CQ);
    f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = q * 4 + e;
      v[e] = c < C ? v[e] * w[(size_t)c * wstride] : 0.f;
    }
    reinterpret_cast<f32x4*>(y)[i] = v;
  }
}

extern "C" int vmtl_stitch(const float* x, const float* w, float* y, long long M, int C, int Cs, int wstride,
                           void* stream) {
  VMTL_ENTER();
  if (!x || !w || !y || (Cs & 3) || M <= 0 || C > Cs) return VMTL_ERR_ARG;
  const long long total4 = M * (Cs >> 2);
  hipLaunchKernelGGL(stitch_kernel, dim3(ew_grid(total4)), dim3(256), 0, (hipStream_t)stream, x, w, y, C, Cs, wstride,
                     total4);
  return vmtl_check_launch();
}

// ------------------------------------------------------------------ plain elementwise
// mode 0: y = a + b   mode 1: y = sigmoid(a)   mode 2: y = b * a * (1 - a)  (sigmoid backward from output a)
// mode 3: y = a * s (s = *scalar_ptr)
__global__ __launch_bounds__(256) void ew_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                 float* __restrict__ y, int mode, long long total) {
  GRID_STRIDE(i, total) {
    const float av = a[i];
    float r;
    if (mode == 0) r = av + b[i];
    else if (mode == 1) r = 1.f / (1.f + expf(-av));
    else if (mode == 2) r = b[i] * av * (1.f - av);
    else r = av * b[0];
    y[i] = r;
  }
}

__global__ __launch_bounds__(256) void axpby_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                    float* __restrict__ y, float wa, float wb, long long total) {
  GRID_STRIDE(i, total) y[i] = wa * a[i] + wb * b[i];
}

// y = wa * a + wb * b (the weighted sum of the two task losses, reference lit_module.py:127-129)
extern "C" int vmtl_axpby(const float* a, const float* b, float* y, float wa, float wb, long long total, void* stream) {
  VMTL_ENTER();
  if (!a || !b || !y || total <= 0) return VMTL_ERR_ARG;
  hipLaunchKernelGGL(axpby_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, a, b, y, wa, wb, total);
  return vmtl_check_launch();
}

// y = a + b (+ c) (+ d) on float4 lanes: the gradient sum of an activation with 2..4 consumers (ops.fork) - autograd's own
// accumulation would be n-1 ATen launches over 3 (n-1) tensor passes, this is one launch over n + 1
__global__ __launch_bounds__(256) void add_n_kernel(const f32x4* __restrict__ a, const f32x4* __restrict__ b,
                                                    const f32x4* __restrict__ c, const f32x4* __restrict__ d,
                                                    f32x4* __restrict__ y, long long n4) {
  GRID_STRIDE(i, n4) {
    f32x4 r = a[i] + b[i];
    if (c != nullptr) r += c[i];
    if (d != nullptr) r += d[i];
    y[i] = r;
  }
}

// c / d may be null; total % 4 == 0 and 16-byte aligned operands (activations in this library always are)
extern "C" int vmtl_add_n(const float* a, const float* b, const float* c, const float* d, float* y, long long total,
                          void* stream) {
  VMTL_ENTER();
  if (!a || !b || !y || total <= 0 || (total & 3) || (d && !c)) return VMTL_ERR_ARG;
  if ((((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)d | (uintptr_t)y) & 15) != 0) return VMTL_ERR_ARG;
  hipLaunchKernelGGL(add_n_kernel, dim3(ew_grid(total / 4)), dim3(256), 0, (hipStream_t)stream, (const f32x4*)a,
                     (const f32x4*)b, (const f32x4*)c, (const f32x4*)d, (f32x4*)y, total / 4);
  return vmtl_check_launch();
}

extern "C" int vmtl_eltwise(const float* a, const float* b, float* y, int mode, long long total, void* stream) {
  VMTL_ENTER();
  if (!a || !y || total <= 0 || mode < 0 || mode > 3 || (mode != 1 && !b)) return VMTL_ERR_ARG;
  hipLaunchKernelGGL(ew_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, a, b, y, mode, total);
  return vmtl_check_launch();
}

// argmax over the channel axis; element (b,c,hw) at z[b*sb + c*sc + hw*sp] (first maximum wins,
// like torch.argmax on ties)
__global__ __launch_bounds__(256) void argmax_kernel(const float* __restrict__ z, long long* __restrict__ out, int HW,
                                                     int C, long long sb, long long sc, long long sp, long long P) {
  GRID_STRIDE(i, P) {
    const long long b = i / HW, hw = i - b * HW;
    const float* r = z + b * sb + hw * sp;
    float m = r[0];
    int am = 0;
    for (int c = 1; c < C; ++c) {
      const float v = r[c * sc];
      if (v > m) { m = v; am = c; }
    }
    out[i] = am;
  }
}

extern "C" int vmtl_argmax_channels(const float* z, long long* out, int B, int HW, int C, long long sb, long long sc,
                                    long long sp, void* stream) {
  VMTL_ENTER();
  if (!z || !out || B <= 0 || HW <= 0 || C <= 0) return VMTL_ERR_ARG;
  const long long P = (long long)B * HW;
  hipLaunchKernelGGL(argmax_kernel, dim3(ew_grid(P)), dim3(256), 0, (hipStream_t)stream, z, out, HW, C, sb, sc, sp, P);
  return vmtl_check_launch();
}

// ------------------------------------------------------------------ boundary layout changes
// NCHW [B][C][H][W] -> NHWC [B][H][W][Cs] and back.  Per pixel the first Cw channels are written
// (c < C from x, C <= c < Cw zero); Cw == Cs writes whole pixels, Cw < Cs fills a channel slice of a
// wider tensor (y may point at a channel offset inside the pixel).
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y, int C,
                                                           int HW, int Cs, int Cw, long long total) {
  GRID_STRIDE(i, total) {
    const int c = (int)(i % Cw);
    const long long pix = i / Cw;
    const int hw = (int)(pix % HW);
    const long long b = pix / HW;
    y[(size_t)pix * Cs + c] = c < C ? x[((size_t)b * C + c) * HW + hw] : 0.f;
  }
}

__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ x, float* __restrict__ y, int C,
                                                           int HW, int Cs, long long total) {
  GRID_STRIDE(i, total) {
    const int hw = (int)(i % HW);
    const long long bc = i / HW;
    const int c = (int)(bc % C);
    const long long b = bc / C;
    y[i] = x[((size_t)b * HW + hw) * Cs + c];
  }
}

extern "C" int vmtl_nchw_to_nhwc(const float* x, float* y, int B, int C, int HW, int Cs, int Cw, void* stream) {
  VMTL_ENTER();
  if (!x || !y || C > Cw || Cw > Cs || B <= 0) return VMTL_ERR_ARG;
  const long long total = (long long)B * HW * Cw;
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, x, y, C, HW, Cs,
                     Cw, total);
  return vmtl_check_launch();
}

extern "C" int vmtl_nhwc_to_nchw(const float* x, float* y, int B, int C, int HW, int Cs, void* stream) {
  VMTL_ENTER();
  if (!x || !y || C > Cs || B <= 0) return VMTL_ERR_ARG;
  const long long total = (long long)B * C * HW;
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, x, y, C, HW, Cs,
                     total);
  return vmtl_check_launch();
}

// Dataset sample layout -> model input layout in one pass: x is [P][C] (HWC pixels as the reference's .npy / PNG
// samples store them: data_modules/cityscapes.py:69-83, nyuv2.py:100-141), y is the internal NHWC storage
// [P][Cs] with zero pad channels; y = x * scale (scale = 1/255 for 8-bit sources that were not rescaled on the host).
__global__ __launch_bounds__(256) void hwc_pad_kernel(const float* __restrict__ x, float* __restrict__ y, int C, int Cs,
                                                      float scale, long long total) {
  GRID_STRIDE(i, total) {
    const int c = (int)(i % Cs);
    const long long pix = i / Cs;
    y[i] = c < C ? x[pix * C + c] * scale : 0.f;
  }
}

extern "C" int vmtl_hwc_to_nhwc_pad(const float* x, float* y, long long P, int C, int Cs, float scale, void* stream) {
  VMTL_ENTER();
  if (!x || !y || P <= 0 || C <= 0 || C > Cs || (Cs & 3)) return VMTL_ERR_ARG;
  const long long total = P * Cs;
  hipLaunchKernelGGL(hwc_pad_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, x, y, C, Cs, scale, total);
  return vmtl_check_launch();
}

// n floats <- 0 (memset node: capturable).  The bias gradient of a conv that feeds a train-mode BatchNorm is
// analytically zero (the batch mean absorbs the bias: reference models/mtan_model.py:31-47 build exactly that), so
// those gradients are written as zeros instead of being summed out of dY.
extern "C" int vmtl_fill_zero(float* p, long long n, void* stream) {
  VMTL_ENTER();
  if (!p || n <= 0) return VMTL_ERR_ARG;
  return hipMemsetAsync(p, 0, (size_t)n * sizeof(float), (hipStream_t)stream) == hipSuccess ? VMTL_OK : VMTL_ERR_LAUNCH;
}
