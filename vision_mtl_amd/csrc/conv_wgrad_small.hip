// Weight gradient of the NARROW full-resolution 3x3 convs (<= 36 storage channels on both sides, stride 1, pad 1, width a
// multiple of 32): the layers where conv_wgrad_kernel is furthest from any roofline (csnet's 32 -> 16 / 16 -> 16 layers at 1 M
// pixels: 407 / 272 us at 47 / 36 TF; MTAN's 32 -> 32: 246 us; basic's 33 -> 33 / 33 -> 20: 277 / 197 us).  That kernel gathers
// im2col(x) per tap: every input pixel is fetched nine times (by up to three column-tile workgroups), 60 KB of loads per 32
// pixels for 16-36 useful channels.
//
// Here a workgroup walks DOWN a strip of 32 output columns.  Per step (one output row of the strip) it loads ONE new input row
// segment (34 pixels with the halo) and one dY row segment; the three input rows a 3x3 window needs sit in a ring of four LDS
// row slots, so x is read from memory once (+ 2 halo rows per strip segment and 2 halo columns per row).  The nine taps are
// shifted LDS views:  slab[co][tap*Cs + ci] = sum_pixels dY[p][co] * X[p + tap][ci]  with the MFMA's k = 4 consecutive pixels
// of the row (lane quarter q = pixel 4s+q), A = dY (rows co), B = X at the tap's shift (columns kk = tap*Cs + ci); the 16-wide
// kk tiles are dealt round-robin to the four waves (a tile that straddles two taps just has per-lane offsets).  LDS pixels are
// [channel] rows of 48 floats (== 16 mod 32: the four quarters of a fragment read start 16 banks apart); the floats past Cs /
// ldy are zeroed once and never written again - lanes of a kk (co) tile beyond Ktot (ldy) read them.
// One slab per workgroup, summed by vmtl_unpack_weights like the slabs of conv_wgrad_kernel (same [Nw][9*Cs] layout).
#include "common.h"

#define WS_SW 32                        // strip width = pixels per step
#define WS_LDP 48                       // LDS floats per pixel
#define WS_XROW ((WS_SW + 2) * WS_LDP)  // one input row slot (with the halo columns)
#define WS_YROW (WS_SW * WS_LDP)

struct WsP {
  const float* x;   // [B][H][W][Cs]
  const float* dy;  // [B][H][W][ldy]
  float* slabs;     // [segments][Nw][9*Cs]
  int B, H, W, Cs, ldy, Nw, Ktot;
  int R;            // output rows per strip segment
  int bands, strips;
};

template <int NI, int JW>
__global__ __launch_bounds__(256) void wgrad_small_kernel(WsP p) {
  __shared__ __attribute__((aligned(16))) float Xs[4 * WS_XROW];
  __shared__ __attribute__((aligned(16))) float Ys[2 * WS_YROW];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wn = tid >> 6;
  const int l15 = lane & 15, lq = lane >> 4;

  const int seg = blockIdx.x;
  const int band = seg % p.bands;
  const int t0 = seg / p.bands;
  const int strip = t0 % p.strips, b = t0 / p.strips;
  const int c0 = strip * WS_SW;
  const int r0 = band * p.R, r1 = min(p.H, r0 + p.R);

  // zero the whole LDS image once: the floats past Cs / ldy of every pixel stay zero (the loaders never touch them)
  for (int i = tid; i < (4 * WS_XROW) / 4; i += 256) reinterpret_cast<f32x4*>(Xs)[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int i = tid; i < (2 * WS_YROW) / 4; i += 256) reinterpret_cast<f32x4*>(Ys)[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ---- loaders: per-thread constants (pixel of the segment, channel quad), the row is a scalar ----
  const int CQ = p.Cs >> 2, YQ = p.ldy >> 2;
  constexpr unsigned OOB = 0xFFFFFFFFu;
  int xoff[2], xlds[2];  // byte offset inside an image row of x (or -1), LDS float offset inside a row slot
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int idx = tid + it * 256;
    const int px = idx / CQ, q = idx - px * CQ;
    const int c = c0 - 1 + px;
    const bool ok = idx < (WS_SW + 2) * CQ && c >= 0 && c < p.W;
    xoff[it] = ok ? (c * p.Cs + q * 4) * 4 : -1;
    xlds[it] = idx < (WS_SW + 2) * CQ ? px * WS_LDP + q * 4 : -1;
  }
  int yoff[2], ylds[2];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int idx = tid + it * 256;
    const int px = idx / YQ, q = idx - px * YQ;
    const bool ok = idx < WS_SW * YQ;
    yoff[it] = ok ? ((c0 + px) * p.ldy + q * 4) * 4 : -1;
    ylds[it] = ok ? px * WS_LDP + q * 4 : -1;
  }
  const __amdgpu_buffer_rsrc_t rs_x =
      __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)((unsigned)p.B * p.H * p.W * p.Cs * 4u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_dy =
      __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, (int)((unsigned)p.B * p.H * p.W * p.ldy * 4u), 0x00020000);
  auto bload = [](__amdgpu_buffer_rsrc_t r, unsigned off) -> f32x4 {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
  };
  // input row rr of image b (zeros outside the image): scalar row base + thread constant
  auto load_x = [&](int rr, f32x4* rx) {
    const bool rok = rr >= 0 && rr < p.H;
    const unsigned base = (unsigned)(b * p.H + rr) * (unsigned)p.W * (unsigned)p.Cs * 4u;
#pragma unroll
    for (int it = 0; it < 2; ++it) rx[it] = bload(rs_x, (rok & (xoff[it] >= 0)) ? base + (unsigned)xoff[it] : OOB);
  };
  auto load_y = [&](int rr, f32x4* ry) {
    const bool rok = rr < r1;  // rows past the segment (or the image) contribute nothing
    const unsigned base = (unsigned)(b * p.H + rr) * (unsigned)p.W * (unsigned)p.ldy * 4u;
#pragma unroll
    for (int it = 0; it < 2; ++it) ry[it] = bload(rs_dy, (rok & (yoff[it] >= 0)) ? base + (unsigned)yoff[it] : OOB);
  };
  auto store_x = [&](int rr, const f32x4* rx) {  // row rr lives in slot (rr + 1) & 3
    float* xs = Xs + ((rr + 1) & 3) * WS_XROW;
#pragma unroll
    for (int it = 0; it < 2; ++it)
      if (xlds[it] >= 0) *reinterpret_cast<f32x4*>(xs + xlds[it]) = rx[it];
  };
  auto store_y = [&](int buf, const f32x4* ry) {
    float* ys = Ys + buf * WS_YROW;
#pragma unroll
    for (int it = 0; it < 2; ++it)
      if (ylds[it] >= 0) *reinterpret_cast<f32x4*>(ys + ylds[it]) = ry[it];
  };

  // ---- this wave's kk tiles: j = wn + 4*jw; per lane the tap shift and channel of column kk = 16*j + l15 ----
  int bdh[JW];   // tap row offset -1..1 (0 for a dead lane)
  int bcol[JW];  // float offset inside a row slot of pixel (lq + 1 + dw), channel ci - or of a zero pad for a dead lane
#pragma unroll
  for (int jw = 0; jw < JW; ++jw) {
    const int kk = (wn + 4 * jw) * 16 + l15;
    if (kk < p.Ktot) {
      const int tap = kk / p.Cs, ci = kk - tap * p.Cs;
      const int th = tap / 3;
      bdh[jw] = th - 1;
      bcol[jw] = (lq + tap - 3 * th) * WS_LDP + ci;  // 1 + dw = tap % 3
    } else {
      bdh[jw] = 0;
      bcol[jw] = lq * WS_LDP + p.Cs;  // Cs <= 36 < 48: a zero pad float of every pixel this lane would touch
    }
  }
  const int acol = lq * WS_LDP + l15;  // dY: pixel lq, channel l15 (+ 16 i)

  f32x4 acc[NI][JW];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int jw = 0; jw < JW; ++jw) acc[i][jw] = (f32x4){0.f, 0.f, 0.f, 0.f};

  f32x4 rx[2], ry[2];
  __syncthreads();  // the zero fill is complete before the first row lands
  // prologue: rows r0-1, r0, r0+1 of x and row r0 of dY
  load_x(r0 - 1, rx);
  store_x(r0 - 1, rx);
  load_x(r0, rx);
  store_x(r0, rx);
  load_x(r0 + 1, rx);
  store_x(r0 + 1, rx);
  load_y(r0, ry);
  store_y(0, ry);
  __syncthreads();

  for (int r = r0; r < r1; ++r) {
    const int buf = (r - r0) & 1;
    const bool more = r + 1 < r1;
    if (more) {  // the next step's operands travel under this step's MFMAs
      load_x(r + 2, rx);
      load_y(r + 1, ry);
    }
    const float* ys = Ys + buf * WS_YROW + acol;
    const float* xb[JW];
#pragma unroll
    for (int jw = 0; jw < JW; ++jw) xb[jw] = Xs + ((r + bdh[jw] + 1) & 3) * WS_XROW + bcol[jw];
#pragma unroll
    for (int s = 0; s < WS_SW / 4; ++s) {  // pixels 4s .. 4s+3 of the row, lane quarter q = pixel 4s+q
      float fa[NI], fb[JW];
#pragma unroll
      for (int i = 0; i < NI; ++i) fa[i] = ys[4 * s * WS_LDP + i * 16];
#pragma unroll
      for (int jw = 0; jw < JW; ++jw) fb[jw] = xb[jw][4 * s * WS_LDP];
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int jw = 0; jw < JW; ++jw)
          acc[i][jw] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i], fb[jw], acc[i][jw], 0, 0, 0);
    }
    if (more) {
      store_x(r + 2, rx);  // slot (r + 3) & 3 held row r - 2: dead since the previous step's barrier
      store_y(buf ^ 1, ry);
    }
    __syncthreads();
  }

  // D layout of 16x16x4: column (kk) = lane & 15, row (co) = 4 * (lane >> 4) + register
  float* slab = p.slabs + (size_t)seg * p.Nw * p.Ktot;
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int jw = 0; jw < JW; ++jw) {
      const int kk = (wn + 4 * jw) * 16 + l15;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int co = i * 16 + 4 * lq + e;
        if (co < p.Nw && kk < p.Ktot) slab[(size_t)co * p.Ktot + kk] = acc[i][jw][e];
      }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
static void ws_geometry(int B, int H, int W, int* R, int* bands, int* strips) {
  // ~1024 workgroups (four per CU; 38 KB of LDS each; 512 / 768 / 2048 measured 1-5 % slower), at least 8 rows per segment (two
  // halo rows are loaded on top)
  static EnvInt e_target{"VMTL_WS_TARGET", 1024};  // tuning aid
  const int st = W / WS_SW;
  long long nb = cdivll((long long)env_int(e_target), (long long)B * st);
  if (nb < 1) nb = 1;
  int rows = cdiv(H, (int)nb);
  if (rows < 8) rows = H < 8 ? H : 8;
  *R = rows;
  *bands = cdiv(H, rows);
  *strips = st;
}

extern "C" int vmtl_conv3x3_wgrad_small_supported(int Cs, int ldy, int W) {
  static EnvInt e{"VMTL_WGRAD_SMALL", 1};  // tuning aid: 0 = conv_wgrad_kernel everywhere
  // 33..36 channels on BOTH sides (basic's last decoder conv: 3 co tiles for 33 rows, 24 kk tile slots for 20.25 tiles) stays on
  // conv_wgrad_kernel and its VALU tail row: 274 us there against 284-290 here (tools/bench_conv.py, "halo wgrad" column)
  return env_int(e) != 0 && Cs >= 4 && Cs <= 36 && !(Cs & 3) && ldy >= 4 && ldy <= 36 && !(ldy & 3) && !(Cs > 32 && ldy > 32) &&
         W >= WS_SW && W % WS_SW == 0;
}

// slabs the caller provides: vmtl_conv3x3_wgrad_small_slabs(B, H, W) * Nw * 9 * Cs floats
extern "C" int vmtl_conv3x3_wgrad_small_slabs(int B, int H, int W) {
  if (B <= 0 || H <= 0 || W < WS_SW || W % WS_SW) return 0;
  int R, bands, strips;
  ws_geometry(B, H, W, &R, &bands, &strips);
  return B * strips * bands;
}

template <int NI>
static int ws_launch(const WsP& p, int jwn, int grid, hipStream_t st) {
  switch (jwn) {
    case 1: hipLaunchKernelGGL((wgrad_small_kernel<NI, 1>), dim3(grid), dim3(256), 0, st, p); break;
    case 2: hipLaunchKernelGGL((wgrad_small_kernel<NI, 2>), dim3(grid), dim3(256), 0, st, p); break;
    case 3: hipLaunchKernelGGL((wgrad_small_kernel<NI, 3>), dim3(grid), dim3(256), 0, st, p); break;
    case 4: hipLaunchKernelGGL((wgrad_small_kernel<NI, 4>), dim3(grid), dim3(256), 0, st, p); break;
    case 5: hipLaunchKernelGGL((wgrad_small_kernel<NI, 5>), dim3(grid), dim3(256), 0, st, p); break;
    case 6: hipLaunchKernelGGL((wgrad_small_kernel<NI, 6>), dim3(grid), dim3(256), 0, st, p); break;
    default: return VMTL_ERR_UNSUPPORTED;
  }
  return vmtl_check_launch();
}

// x [B][H][W][Cs], dy [B][H][W][ldy] (3x3, stride 1, pad 1: same extent), slabs [nslabs][Nw][9*Cs] with
// nslabs = vmtl_conv3x3_wgrad_small_slabs(B, H, W); vmtl_unpack_weights(..., nslabs) sums them
extern "C" int vmtl_conv3x3_wgrad_small(const float* x, const float* dy, float* slabs, int nslabs, int B, int H, int W, int Cs,
                                        int ldy, int Nw, void* stream) {
  VMTL_ENTER();
  if (!x || !dy || !slabs || B <= 0 || H <= 0 || Nw <= 0 || Nw > ldy) return VMTL_ERR_ARG;
  if (!vmtl_conv3x3_wgrad_small_supported(Cs, ldy, W)) return VMTL_ERR_UNSUPPORTED;
  if ((long long)B * H * W * Cs * 4 >= (1ll << 32) || (long long)B * H * W * ldy * 4 >= (1ll << 32)) return VMTL_ERR_UNSUPPORTED;
  WsP p;
  p.x = x; p.dy = dy; p.slabs = slabs; p.B = B; p.H = H; p.W = W; p.Cs = Cs; p.ldy = ldy; p.Nw = Nw; p.Ktot = 9 * Cs;
  ws_geometry(B, H, W, &p.R, &p.bands, &p.strips);
  const int grid = B * p.strips * p.bands;
  if (nslabs != grid) return VMTL_ERR_ARG;
  const int nj = cdiv(p.Ktot, 16), jwn = cdiv(nj, 4);
  hipStream_t st = (hipStream_t)stream;
  switch (cdiv(Nw, 16)) {
    case 1: return ws_launch<1>(p, jwn, grid, st);
    case 2: return ws_launch<2>(p, jwn, grid, st);
    case 3: return ws_launch<3>(p, jwn, grid, st);
    default: return VMTL_ERR_UNSUPPORTED;
  }
}
