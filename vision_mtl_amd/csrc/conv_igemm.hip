// Implicit-GEMM convolution on exact-fp32 MFMA (v_mfma_f32_16x16x4_f32), NHWC.
//
// Replaces the ATen conv2d / conv_transpose2d dispatches reached from
//   reference vision_mtl/utils/model_utils.py:71,74 (DoubleConv 3x3),
//   reference vision_mtl/models/mtan_model.py:31-46,105-129,214-216,369 (1x1, 3x3, ConvT),
//   smp UnetDecoder Conv2dReLU / SegmentationHead and timm pointwise convs
//   (reference vision_mtl/utils/model_utils.py:25-34, models/basic_model.py:30-41).
//
// One kernel serves forward and data-gradient: both are "gather rows of an NHWC tensor per
// filter tap, contract against a packed [row][tap*Cs+c] weight matrix".  dgrad of a stride-1
// conv is the same contraction over dY with the tap-flipped, transposed packing (pack.hip).
// A second kernel computes the weight gradient as a deterministic split-K GEMM over pixels.
//
// GEMM view (forward):  Y[m][n] = sum_kk  Xcol[m][kk] * Wp[n][kk]
//   m  = (b, ho, wo)            M    = B*Ho*Wo
//   kk = tap*Cs + ci            Ktot = KH*KW*Cs   (Cs % 4 == 0, pad channels are 0)
//   n  = output channel         rows n >= Nw of Wp are treated as 0
//
// Why 16x16x4 and not 32x32x2: same MFMA rate (64 FLOP/clk/SIMD), but the model's channel counts
// are 33/67/135/270/540 (+ concat 151/294/580/1072): 16-granular N tiles (48/80/144/...) keep
// 70-95 % of the issued MFMAs useful where 32-granular tiles keep ~50 %.
//
// LDS tiles are [row][BK] with kk contiguous: one ds_read_b128 gives a lane 4 consecutive kk of its
// row; lane quarter q = lane>>4 takes kk 4q..4q+3 of each 16-wide k-group and element e feeds MFMA e
// (k quadruple {e, 4+e, 8+e, 12+e}) - identical mapping for A and B.  Rows are 128 bytes (8 slots of
// 16 bytes) and slot s of row r lives at slot s ^ (r & 7): with that swizzle every 16-lane group of the
// fragment ds_read_b128 (and every 8-lane group of the staging ds_write_b128) touches 16 (8) distinct
// slots - conflict-free, where the padded [row][BK+4] layout measured 36 % conflict cycles.
#include <mutex>
#include <unordered_map>

#include "common.h"

#define BK 32
#define LDT BK  // un-padded 128-byte rows; the 16-byte slot index is XOR-swizzled with (row & 7)

struct ConvP {
  const float* x;     // [B][H][W][Cs]
  const float* wp;    // [Nw][Ktot]
  const float* bias;  // [Nw] or nullptr
  float* y;           // [B][Ho][Wo][ldy]  (or pixel-shuffled, see shuffle)
  float* stats;       // optional [tiles_m][2][ldy] per-row-block column (mean, M2) for BatchNorm
  int B, H, W, Cs;
  int Ho, Wo, ldy;
  int Nw;             // valid weight rows (Cout, or 4*Cout for shuffle)
  int Cout;           // logical channels written non-zero per output pixel
  int KH, KW, stride, pad;
  int Ktot, M;
  int act;
  int shuffle;        // 1: rows n=(u*2+v)*Cout+co are scattered to (2h+u, 2w+v, co)
  int tiles_m, tiles_n;
  // UP2 mode (nearest-x2 upsample of x, concat with x2, 3x3 conv, evaluated as four 2x2 phase convs on
  // the low-res map): x is the LOW-res source [B][H][W][Cs]; x2 the full-res skip [B][2H][2W][C2s] (may
  // be null); wp holds 4 phase matrices [4][Nw][Ktot], Ktot = 4*Cs + 9*C2s; M = B*H*W rows per phase.
  const float* x2;
  int C2s;
  // split-K (launches without bias / activation / stats / shuffle only): blockIdx.y = K slice, every
  // slice writes its partial tile to ksl_out + slice * M * ldy; vmtl_sum_slabs adds them into y.
  int ksplit, ksteps_per_split;
  // BatchNorm + activation BACKWARD of the layer that produced the tensor whose gradient this launch computes
  // (data-gradient launches only; ez_x = null: off): y = acc * act'(gamma*xhat + beta), xhat = (ez_x - mean)*invstd,
  // and stats[tile_m][2][ldy] = per-row-block column sums (sum y, sum y*xhat) instead of (mean, M2).
  const float* ez_x;  // [M][ldy], the producer's pre-BatchNorm activation
  const float* ez_mean;
  const float* ez_invstd;
  const float* ez_gamma;
  const float* ez_beta;
  int ez_act;
};

// NT > 0: the last NT output columns of the tile ("tail") are not given an MFMA tile of their own; every
// lane dots its A fragment (already in registers) with the tail weight rows on the VALU.  With 33 / 20 /
// 67 output channels this keeps the MFMA tiles at 32 / 16 / 64 useful columns instead of padding to
// 48 / 32 / 80 (the VALU pipe is otherwise idle next to the matrix pipe).  Needs WAVES_N == 1.
//
// NS >= 2: the staging tiles are filled by LDS-DMA (global_load_lds_dwordx4: no staging registers, no
// ds_write pass) into NS buffers.  One wave-instruction writes 8 consecutive 128-byte rows lane-linearly,
// so the XOR swizzle moves to the SOURCE address: the lane that lands on (row, slot) fetches k-quad
// slot ^ (row & 7) (see k4 below).  Out-of-image taps / tile edges read a 16-byte zero page instead of
// being zero-filled in registers.  Up to NS-1 tiles are in flight across the (raw) barrier.
__device__ __attribute__((aligned(16))) float g_zero_page[4] = {0.f, 0.f, 0.f, 0.f};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ void glds16(const float* src, float* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// X3 = 1 (opt-in, VMTL_BF16X3=1; NOT the default path): every fp32 operand is split EXACTLY into three bf16
// values while it is staged (a = a1 + a2 + a3 by truncation, 8 significand bits each; LDS holds three
// [row][32 bf16] planes per tile) and a product is formed from six v_mfma_f32_16x16x32_bf16 with fp32
// accumulation: a1b1 + a1b2 + a2b1 + a1b3 + a2b2 + a3b1, dropped terms < 2^-24 relative.  Measured error is
// below the fp32-MFMA path's (tools/ubench/gemm_bf16x3_vs_f32.hip); see DESIGN.md section 7 for why it is
// not switched on.
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
  p1 = (u32x2){__builtin_amdgcn_perm(x[1], x[0], 0x07060302u), __builtin_amdgcn_perm(x[3], x[2], 0x07060302u)};
  p2 = (u32x2){__builtin_amdgcn_perm(r1[1], r1[0], 0x07060302u), __builtin_amdgcn_perm(r1[3], r1[2], 0x07060302u)};
  p3 = (u32x2){__builtin_amdgcn_perm(r2[1], r2[0], 0x07060302u), __builtin_amdgcn_perm(r2[3], r2[2], 0x07060302u)};
}

template <int TM, int TN, int WAVES_M, int WAVES_N, bool UP2 = false, int NT = 0, int NS = 0, int X3 = 0>
// The 128x160 tile took 264 registers: ONE wave per SIMD, nothing to overlap its loads with.  Its second launch bound asks
// for two (<= 256 registers: 216-238, no spills): 13-18 % faster on the 145..160-column layers (tools/bench_conv.py dgrad
// blk1-3.c1: 465 -> 405, 501 -> 426, 590 -> 484 us).  Only that tile: the same bound on every instantiation made the
// narrow ones use MORE registers (the allocator stops economising once two waves fit) and cost the step 0.07 ms.
__global__ __launch_bounds__(256, (TM * TN >= 20 && WAVES_N == 2) ? 2 : 1) void conv_igemm_kernel(ConvP p) {
  constexpr int BM = WAVES_M * TM * 16;
  constexpr int BNM = WAVES_N * TN * 16;  // columns covered by MFMA tiles
  constexpr int BN = BNM + NT;            // + tail columns
  constexpr int RA = (BM + 31) / 32;      // A rows per loader thread
  constexpr int RB = (BN + 31) / 32;      // B rows per loader thread
  constexpr int NBUF = NS >= 2 ? NS : 2;
  constexpr int BNR = NS >= 2 ? RB * 32 : BN;  // LDS rows of one B stage (LDS-DMA writes whole 8-row groups)
  static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
  static_assert(NT == 0 || WAVES_N == 1, "tail columns need all waves to span the full tile width");
  static_assert(NS == 0 || (NS >= 2 && NS <= 4 && BM % 32 == 0), "LDS-DMA staging: 2..4 buffers, whole 32-row passes");
  static_assert(X3 == 0 || (NS == 0 && NT == 0), "bf16x3 operands: register staging, no VALU tail columns");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                    // [NBUF][BM][LDT]
  float* Bs = smem + NBUF * BM * LDT;  // [NBUF][BNR][LDT]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = tid >> 6;
  const int wm = wv / WAVES_N, wn = wv % WAVES_N;
  const int l15 = lane & 15, lq = lane >> 4;

  const int nwg = p.tiles_m * p.tiles_n;
  const int bid = xcd_remap(blockIdx.x, nwg);
  const int tile_m = bid / p.tiles_n, tile_n = bid % p.tiles_n;
  const int n0 = tile_n * BN;
  // UP2: the row tiles are grouped by output phase (a, b) = (row parity, column parity)
  const int tiles_q = UP2 ? p.tiles_m >> 2 : p.tiles_m;
  const int phase = UP2 ? tile_m / tiles_q : 0;
  const int pa = phase >> 1, pb = phase & 1;
  const int m0 = (UP2 ? tile_m - phase * tiles_q : tile_m) * BM;
  const float* wp = p.wp + (UP2 ? (size_t)phase * p.Nw * p.Ktot : 0);

  // ---- per-thread loader geometry ----
  const int r0 = tid >> 3;  // 0..31
  // float4 column inside the BK chunk.  Register staging: the thread picks the LDS slot when it stores.
  // LDS-DMA: lane l of a wave lands on row (l>>3) of its 8-row group, slot (l&7) - it must FETCH the
  // k-quad whose swizzled home that slot is: (l&7) ^ (row&7), row&7 == (tid>>3)&7.
  const int k4 = NS >= 2 ? ((tid & 7) ^ (r0 & 7)) : (tid & 7);
  int pixbase[RA], hb[RA], wb[RA];
  // (b, ho, wo) of this thread's first row by two divisions, rows + 32 i by add-and-carry
  int lb, lho, lwo;
  {
    const int hw = p.Ho * p.Wo, m = m0 + r0;
    lb = m / hw;
    const int rem = m - lb * hw;
    lho = rem / p.Wo;
    lwo = rem - lho * p.Wo;
  }
#pragma unroll
  for (int i = 0; i < RA; ++i) {
    const int m = m0 + r0 + 32 * i;
    if (i > 0) {
      lwo += 32;
      while (lwo >= p.Wo) {
        lwo -= p.Wo;
        if (++lho == p.Ho) {
          lho = 0;
          ++lb;
        }
      }
    }
    if (r0 + 32 * i < BM && m < p.M) {
      const int b = lb, ho = lho, wo = lwo;
      pixbase[i] = b * p.H * p.W;
      hb[i] = UP2 ? ho - 1 + pa : ho * p.stride - p.pad;  // UP2: (Ho, Wo) are the low-res dims here
      wb[i] = UP2 ? wo - 1 + pb : wo * p.stride - p.pad;
    } else {
      pixbase[i] = 0;
      hb[i] = -(1 << 20);  // forces the bounds test to fail
      wb[i] = 0;
    }
  }
  // running (segment, tap, ci) of this thread's float4 column.  Segment 0 gathers from x with a
  // KHxKW tap grid (UP2: 2x2 on the low-res map); segment 1 (UP2 only) gathers 3x3 taps from x2.
  int kk = k4 * 4;
  int seg = 0, dh = 0, dw = 0, ci = kk;
  int cseg = p.Cs, kwseg = UP2 ? 2 : p.KW;
  auto normalize = [&]() {
    while (ci >= cseg) {
      ci -= cseg;
      if (++dw == kwseg) {
        dw = 0;
        ++dh;
        if (UP2 && seg == 0 && dh == 2) {
          if (p.C2s == 0) { dh = 1 << 20; break; }  // no skip: the K range ends here (kk >= Ktot anyway)
          seg = 1; dh = 0; cseg = p.C2s; kwseg = 3;
        }
      }
    }
  };
  normalize();

  f32x4 ra[RA], rb[RB];

  // Register staging goes through BUFFER loads: 32-bit byte offsets instead of 64-bit pointer arithmetic, and
  // an out-of-image tap / tile edge is an out-of-range offset, which the hardware answers with zeros - no
  // zero-fill moves, no divergent branch around the load (the narrow tiles were VALU-issue bound: 7 VALU
  // instructions per MFMA, measured with SQ_INSTS_VALU / SQ_INSTS_MFMA).
  const __amdgpu_buffer_rsrc_t rs_x =
      __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)((unsigned)p.B * p.H * p.W * p.Cs * 4u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_x2 = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(UP2 && p.x2 != nullptr ? p.x2 : p.x), 0, UP2 ? (int)(16u * p.B * p.H * p.W * p.C2s) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)wp, 0, (int)((unsigned)p.Nw * p.Ktot * 4u), 0x00020000);
  constexpr unsigned OOB = 0xFFFFFFFFu;
  auto bload = [](__amdgpu_buffer_rsrc_t r, unsigned off) -> f32x4 {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
  };

  // UP2: which gather segment(s) a BK chunk touches is wave-uniform (kchunk is a scalar).  Branching on it keeps the
  // buffer descriptor of each load uniform instead of a per-lane select between the two tensors (the lanes disagree
  // only in the one chunk that straddles the boundary): 437 -> 420 us on the 67 -> 33 layer (tools/bench_up2.py).
  int kchunk = 0;
  // byte offset of a gathered element = row part (fixed per thread row, computed once) + tap part (one per chunk):
  // ((pix + h*W + w)*Cs + ci)*4 with h = hb + dh, w = wb + dw splits into (pix + hb*W + wb)*Cs*4 + ((dh*W + dw)*Cs + ci)*4
  // (mod 2^32; in-range elements never wrap) - 2 integer multiplies per chunk instead of 2 per row and chunk
  // (v_mul_lo_u32 is quarter rate: they were 8 of the ~40 VALU instructions per MFMA-free slot of the narrow tiles)
  unsigned rowb0[RA], rowb1[RA];
#pragma unroll
  for (int i = 0; i < RA; ++i) {
    rowb0[i] = (unsigned)(pixbase[i] + hb[i] * p.W + wb[i]) * (unsigned)p.Cs * 4u;
    rowb1[i] = UP2 ? (unsigned)(4 * pixbase[i] + (2 * hb[i] + 1 - pa) * (2 * p.W) + (2 * wb[i] + 1 - pb)) *
                         (unsigned)p.C2s * 4u
                   : 0u;
  }
  unsigned wrowb[RB];  // weight rows likewise: n * Ktot * 4 once
  bool wrow_ok[RB];
#pragma unroll
  for (int i = 0; i < RB; ++i) {
    const int n = n0 + r0 + 32 * i;
    wrow_ok[i] = n < p.Nw && r0 + 32 * i < BN;
    wrowb[i] = (unsigned)n * (unsigned)p.Ktot * 4u;
  }
  auto load_a_seg0 = [&](int i, bool kok, unsigned tapb) {
    const int h = hb[i] + dh, w = wb[i] + dw;
    const bool ok = kok && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;
    ra[i] = bload(rs_x, ok ? rowb0[i] + tapb : OOB);
  };
  auto load_a_seg1 = [&](int i, bool kok, unsigned tapb) {
    // full-res skip: output pixel (2*h2 + pa, 2*w2 + pb), tap offset dh-1 / dw-1; hb = h2 - 1 + pa
    const int h = 2 * hb[i] + 1 - pa + dh, w = 2 * wb[i] + 1 - pb + dw;
    const bool ok = kok && (unsigned)h < (unsigned)(2 * p.H) && (unsigned)w < (unsigned)(2 * p.W);
    ra[i] = bload(rs_x2, ok ? rowb1[i] + tapb : OOB);
  };
  auto load_tile = [&]() {
    const bool kok = kk < p.Ktot;
    const int seg_end = 4 * p.Cs;  // first K index of segment 1
    const unsigned tap0 = ((unsigned)(dh * p.W + dw) * (unsigned)p.Cs + (unsigned)ci) * 4u;
    if (!UP2 || kchunk + BK <= seg_end || p.C2s == 0) {
#pragma unroll
      for (int i = 0; i < RA; ++i) load_a_seg0(i, kok && (!UP2 || seg == 0), tap0);
    } else {
      const unsigned tap1 = ((unsigned)(dh * (2 * p.W) + dw) * (unsigned)p.C2s + (unsigned)ci) * 4u;
      if (kchunk >= seg_end) {
#pragma unroll
        for (int i = 0; i < RA; ++i) load_a_seg1(i, kok, tap1);
      } else {  // the chunk straddling the boundary: per lane
#pragma unroll
        for (int i = 0; i < RA; ++i) {
          if (seg == 0) load_a_seg0(i, kok, tap0);
          else load_a_seg1(i, kok, tap1);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < RB; ++i) rb[i] = bload(rs_w, (kok && wrow_ok[i]) ? wrowb[i] + (unsigned)kk * 4u : OOB);
    kchunk += BK;
    // advance to the next BK chunk
    kk += BK;
    ci += BK;
    normalize();
  };
  // X3 LDS image (bytes): A planes [2][3][BM][64], then B planes [2][3][BN][64]; the 16-byte slot s of row r
  // sits at s ^ ((r >> 2) & 3) (rows r, r+4, r+8, r+12 of a fragment read would share a bank group otherwise)
  unsigned char* A3 = reinterpret_cast<unsigned char*>(smem);
  unsigned char* B3 = A3 + 2 * 3 * BM * 64;
  auto store_tile = [&](int buf) {
    if constexpr (X3) {
#pragma unroll
      for (int i = 0; i < RA; ++i) {
        const int row = r0 + 32 * i;
        if (BM % 32 == 0 || row < BM) {
          const int off = row * 64 + (((k4 >> 1) ^ ((row >> 2) & 3)) << 4) + ((k4 & 1) << 3);
          u32x2 p1, p2, p3;
          split3(ra[i], p1, p2, p3);
          *reinterpret_cast<u32x2*>(A3 + ((buf * 3 + 0) * BM) * 64 + off) = p1;
          *reinterpret_cast<u32x2*>(A3 + ((buf * 3 + 1) * BM) * 64 + off) = p2;
          *reinterpret_cast<u32x2*>(A3 + ((buf * 3 + 2) * BM) * 64 + off) = p3;
        }
      }
#pragma unroll
      for (int i = 0; i < RB; ++i) {
        const int row = r0 + 32 * i;
        if (BN % 32 == 0 || row < BN) {
          const int off = row * 64 + (((k4 >> 1) ^ ((row >> 2) & 3)) << 4) + ((k4 & 1) << 3);
          u32x2 p1, p2, p3;
          split3(rb[i], p1, p2, p3);
          *reinterpret_cast<u32x2*>(B3 + ((buf * 3 + 0) * BN) * 64 + off) = p1;
          *reinterpret_cast<u32x2*>(B3 + ((buf * 3 + 1) * BN) * 64 + off) = p2;
          *reinterpret_cast<u32x2*>(B3 + ((buf * 3 + 2) * BN) * 64 + off) = p3;
        }
      }
      return;
    }
    float* a = As + buf * BM * LDT;
    float* b = Bs + buf * BNR * LDT;
    const int ks = (k4 ^ (r0 & 7)) * 4;  // (r0 + 32 i) & 7 == r0 & 7
#pragma unroll
    for (int i = 0; i < RA; ++i)
      if (BM % 32 == 0 || r0 + 32 * i < BM) *reinterpret_cast<f32x4*>(a + (r0 + 32 * i) * LDT + ks) = ra[i];
#pragma unroll
    for (int i = 0; i < RB; ++i)
      if (BN % 32 == 0 || r0 + 32 * i < BN) *reinterpret_cast<f32x4*>(b + (r0 + 32 * i) * LDT + ks) = rb[i];
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // tail columns: per-lane partial dot products over this lane's k quads, kept as (even k, odd k) pairs so
  // the multiply-adds are v_pk_fma_f32 on register pairs that already sit next to each other
  f32x2 tacc[TM][NT > 0 ? NT : 1];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int t = 0; t < (NT > 0 ? NT : 1); ++t) tacc[i][t] = (f32x2){0.f, 0.f};

  int nk = (p.Ktot + BK - 1) / BK;
  if (p.ksplit > 1) {  // this workgroup's K slice: skip ahead, then run ksteps_per_split chunks
    const int k_begin = blockIdx.y * p.ksteps_per_split;
    nk = min(nk - k_begin, p.ksteps_per_split);
    kk += k_begin * BK;
    ci += k_begin * BK;
    kchunk += k_begin * BK;
    normalize();
  }
  // tail columns that hold real weight rows (the others multiply zeros): wave-uniform
  const int ntc = NT > 0 ? max(0, min(NT, p.Nw - (n0 + BNM))) : 0;
  // one BK chunk of MFMAs (+ VALU tail columns) on staging buffer `cur`
  auto compute = [&](int cur) {
    if constexpr (X3) {
      // lane (l15, lq) holds k = 8*lq .. 8*lq+7 of its row: the 16-byte slot lq of each plane
      bf16x8 fa[TM][3];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = (wm * TM + i) * 16 + l15;
        const int off = row * 64 + ((lq ^ ((row >> 2) & 3)) << 4);
#pragma unroll
        for (int s3 = 0; s3 < 3; ++s3) fa[i][s3] = *reinterpret_cast<const bf16x8*>(A3 + ((cur * 3 + s3) * BM) * 64 + off);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int row = (wn * TN + j) * 16 + l15;
        const int off = row * 64 + ((lq ^ ((row >> 2) & 3)) << 4);
        bf16x8 fb[3];
#pragma unroll
        for (int s3 = 0; s3 < 3; ++s3) fb[s3] = *reinterpret_cast<const bf16x8*>(B3 + ((cur * 3 + s3) * BN) * 64 + off);
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
      return;
    }
    // fragment rows are (tile base + l15) with tile bases multiples of 16: row & 7 == l15 & 7
    const float* a = As + cur * BM * LDT + (wm * TM * 16 + l15) * LDT;
    const float* b = Bs + cur * BNR * LDT + (wn * TN * 16 + l15) * LDT;
#pragma unroll
    for (int kg = 0; kg < BK / 16; ++kg) {
      const int so = ((kg * 4 + lq) ^ (l15 & 7)) * 4;
      f32x4 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const f32x4*>(a + i * 16 * LDT + so);
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const f32x4*>(b + j * 16 * LDT + so);
#ifdef VMTL_SETPRIO
      __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i][e], fb[j][e], acc[i][j], 0, 0, 0);
#ifdef VMTL_SETPRIO
      __builtin_amdgcn_s_setprio(0);
#endif
      if (NT > 0) {
        // tail weight rows BNM + t (wave-uniform row, per-quarter k slot): 4 distinct LDS addresses
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          if (t >= ntc) break;  // weight rows past Nw are zero: skip their FMAs (33 outputs = 32 MFMA columns + ONE tail)
          const f32x4 ft = *reinterpret_cast<const f32x4*>(Bs + cur * BNR * LDT + (BNM + t) * LDT +
                                                           (((kg * 4 + lq) ^ ((BNM + t) & 7)) * 4));
          const f32x2 ft_lo = __builtin_shufflevector(ft, ft, 0, 1), ft_hi = __builtin_shufflevector(ft, ft, 2, 3);
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            tacc[i][t] += __builtin_shufflevector(fa[i], fa[i], 0, 1) * ft_lo;
            tacc[i][t] += __builtin_shufflevector(fa[i], fa[i], 2, 3) * ft_hi;
          }
        }
      }
    }
  };

  if constexpr (NS >= 2) {
    constexpr int G = RA + RB;  // LDS-DMA instructions per wave and stage
    auto issue = [&](int buf) {
      const bool kok = kk < p.Ktot;
      float* ad = As + (buf * BM + wv * 8) * LDT;
#pragma unroll
      for (int i = 0; i < RA; ++i) {
        const float* src = g_zero_page;
        if (!UP2 || seg == 0) {
          const int h = hb[i] + dh, w = wb[i] + dw;
          if (kok && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W)
            src = p.x + (size_t)(pixbase[i] + h * p.W + w) * p.Cs + ci;
        } else {
          const int h = 2 * hb[i] + 1 - pa + dh, w = 2 * wb[i] + 1 - pb + dw;
          if (kok && (unsigned)h < (unsigned)(2 * p.H) && (unsigned)w < (unsigned)(2 * p.W))
            src = p.x2 + ((size_t)4 * pixbase[i] + (size_t)h * (2 * p.W) + w) * p.C2s + ci;
        }
        glds16(src, ad + i * 32 * LDT);
      }
      float* bd = Bs + (buf * BNR + wv * 8) * LDT;
#pragma unroll
      for (int i = 0; i < RB; ++i) {
        const int n = n0 + r0 + 32 * i;
        const float* src = g_zero_page;
        if (kok && n < p.Nw && r0 + 32 * i < BN) src = wp + (size_t)n * p.Ktot + kk;
        glds16(src, bd + i * 32 * LDT);
      }
      kk += BK;
      ci += BK;
      normalize();
    };
    int issued = 0;
    for (; issued < NS - 1 && issued < nk; ++issued) issue(issued);
    for (int kt = 0; kt < nk; ++kt) {
      // stage kt must have landed; up to NS-2 younger stages may stay in flight across the barrier
      const int pending = issued - kt - 1;
      if (NS >= 4 && pending >= 2) wait_vmcnt<2 * G>();
      else if (NS >= 3 && pending >= 1) wait_vmcnt<G>();
      else wait_vmcnt<0>();
      asm volatile("s_barrier" ::: "memory");  // raw: a __syncthreads() here would drain the DMA queue
      if (issued < nk) {  // refill the buffer everybody finished reading before this barrier
        issue(issued % NS);
        ++issued;
      }
      compute(kt % NS);
    }
    __syncthreads();  // the epilogue reuses the staging memory
  } else {
    load_tile();
    store_tile(0);
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
      const bool more = kt + 1 < nk;
      if (more) load_tile();  // global loads stay in flight under the MFMAs
      compute(cur);
      if (more) store_tile(cur ^ 1);
      __syncthreads();
      cur ^= 1;
    }
  }

  // ---- BatchNorm-backward epilogue (see ConvP::ez_x): dz = acc * act'(z) in place, with the column sums ----
  float zs1[TN], zs2[TN];
  if (p.ez_x != nullptr) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + (wn * TN + j) * 16 + l15;
      const bool nok = n < p.Cout;
      const float em = nok ? p.ez_mean[n] : 0.f, ei = nok ? p.ez_invstd[n] : 0.f;
      const float eg = nok ? p.ez_gamma[n] : 0.f, eb = nok ? p.ez_beta[n] : 0.f;
      float a1 = 0.f, a2 = 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = m0 + (wm * TM + i) * 16 + 4 * lq + r;
          float v = 0.f;
          if (nok && m < p.M) {
            const float xh = (p.ez_x[(size_t)m * p.ldy + n] - em) * ei;
            v = acc[i][j][r] * act_grad(eg * xh + eb, p.ez_act);
            a1 += v;
            a2 += v * xh;
          }
          acc[i][j][r] = v;
        }
      zs1[j] = a1;
      zs2[j] = a2;
    }
  }

  // ---- epilogue: bias + activation, zero the pad channels, store ----
  // C layout of 16x16x4: col = lane & 15, row = 4 * (lane >> 4) + reg
  const int hw = p.Ho * p.Wo;
  // split-K: slice blockIdx.y writes its partial tile into its own slab (UP2: a slab has the full-resolution layout)
  float* yout = p.ksplit > 1 ? p.y + (size_t)blockIdx.y * (UP2 ? 4 : 1) * p.M * p.ldy : p.y;
  // UP2: full-resolution pixel offset of each of this thread's output rows.  (b, h, w) of the thread's first row by two
  // integer divisions, the other rows by add-and-carry: the divisions per (row, register) of round 2 were ~20 per
  // thread = ~600 VALU instructions of an epilogue whose K loop has ~540 (the 67 -> 33 full-resolution layer runs
  // 9 K steps per workgroup: 13.4 VALU per MFMA, tools/bench_up2.py under rocprofv3 --pmc)
  size_t up2_off[TM][4];  // also the pixel-shuffle (ConvTranspose) scatter: pa = pb = 0 there, the (u, v) part is per column
  if (UP2 || p.shuffle) {
    const int mb = m0 + wm * TM * 16 + 4 * lq;
    const int b0 = mb / hw, rem0 = mb - b0 * hw;
    const int h0 = rem0 / p.Wo, w0 = rem0 - h0 * p.Wo;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int w_ = w0 + i * 16 + r, h_ = h0, b_ = b0;
        while (w_ >= p.Wo) {
          w_ -= p.Wo;
          if (++h_ == p.Ho) {
            h_ = 0;
            ++b_;
          }
        }
        up2_off[i][r] = ((size_t)(b_ * 2 * p.Ho + 2 * h_ + pa) * (2 * p.Wo) + 2 * w_ + pb) * p.ldy;
      }
  }
  float bv[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + (wn * TN + j) * 16 + l15;
    // pixel shuffle: column n = (u*2+v)*Cout + co goes to pixel (2h+u, 2w+v), channel co - one division per column
    int shco = n;
    size_t shuf_off = 0;
    if (p.shuffle) {  // uniform branch: plain launches pay no division here
      const int shq = n / p.Cout;
      shco = n - shq * p.Cout;
      shuf_off = ((size_t)(shq >> 1) * (2 * p.Wo) + (shq & 1)) * p.ldy + shco;
    }
    bv[j] = (p.bias != nullptr && n < p.Nw) ? p.bias[p.shuffle ? shco : n] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + (wm * TM + i) * 16 + 4 * lq + r;
        if (m >= p.M) continue;
        float v = act_fwd(acc[i][j][r] + bv[j], p.act);
        if (UP2) {
          if (n < p.ldy) {
            if (n >= p.Cout) v = 0.f;
            yout[up2_off[i][r] + n] = v;
          }
        } else if (!p.shuffle) {
          if (n < p.ldy) {
            if (n >= p.Cout) v = 0.f;
            yout[(size_t)m * p.ldy + n] = v;
          }
        } else if (n < p.Nw) {
          p.y[up2_off[i][r] + shuf_off] = v;
        }
      }
    }
  }

  // ---- tail columns: fold the four k quarters, lanes of quarter 0 own one output row each ----
  float tv[TM][NT > 0 ? NT : 1], txh[TM][NT > 0 ? NT : 1];
  size_t up2_tail_off[(UP2 && NT > 0) ? TM : 1];
  if (UP2 && NT > 0) {  // row (wm*TM + i)*16 + l15 of the tile: one division pair, + 16 rows per i by add-and-carry
    const int mb = m0 + wm * TM * 16 + l15;
    const int b0 = mb / hw, rem0 = mb - b0 * hw;
    const int h0 = rem0 / p.Wo, w0 = rem0 - h0 * p.Wo;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      int w_ = w0 + i * 16, h_ = h0, b_ = b0;
      while (w_ >= p.Wo) {
        w_ -= p.Wo;
        if (++h_ == p.Ho) {
          h_ = 0;
          ++b_;
        }
      }
      up2_tail_off[i] = ((size_t)(b_ * 2 * p.Ho + 2 * h_ + pa) * (2 * p.Wo) + 2 * w_ + pb) * p.ldy;
    }
  }
  if (NT > 0) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        float v = quarter_sum(tacc[i][t][0] + tacc[i][t][1]);
        const int n = n0 + BNM + t;
        const int m = m0 + (wm * TM + i) * 16 + l15;
        const float bt = (p.bias != nullptr && n < p.Nw) ? p.bias[n] : 0.f;
        v = (n < p.Cout) ? act_fwd(v + bt, p.act) : 0.f;
        if (p.ez_x != nullptr) {  // BatchNorm-backward epilogue on the tail column (one pixel per lane of quarter 0)
          float xh = 0.f;
          if (n < p.Cout && m < p.M) {
            xh = (p.ez_x[(size_t)m * p.ldy + n] - p.ez_mean[n]) * p.ez_invstd[n];
            v *= act_grad(p.ez_gamma[n] * xh + p.ez_beta[n], p.ez_act);
          } else {
            v = 0.f;
          }
          txh[i][t] = xh;
        }
        tv[i][t] = v;
        if (lq == 0 && m < p.M && n < p.ldy) {
          if (UP2) {
            yout[up2_tail_off[i] + n] = v;
          } else {
            yout[(size_t)m * p.ldy + n] = v;
          }
        }
      }
  }

  // ---- BatchNorm partials of this row block, straight from the accumulators: per column the
  // block-local MEAN and M2 = sum (v - mean)^2 (two in-register passes).  The finalize kernel
  // merges blocks with Chan's parallel-variance formula in fp64, so the variance never goes
  // through E[x^2] - E[x]^2 (which loses everything when |mean| >> std).
  if (p.stats != nullptr && p.ez_x != nullptr) {
    // column sums (sum dz, sum dz*xhat) of this row block: lane quarters -> waves -> one row pair per block
    float* red = smem;  // [2][WAVES_M][BN]; the staging tiles are dead after the last barrier
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      float a1 = zs1[j], a2 = zs2[j];
      a1 = quarter_sum(a1);
      a2 = quarter_sum(a2);
      if (lq == 0) {
        red[wm * BN + (wn * TN + j) * 16 + l15] = a1;
        red[(WAVES_M + wm) * BN + (wn * TN + j) * 16 + l15] = a2;
      }
    }
    if (NT > 0) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        float a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          a1 += tv[i][t];
          a2 += tv[i][t] * txh[i][t];
        }
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
          a1 += __shfl_xor(a1, o, 64);
          a2 += __shfl_xor(a2, o, 64);
        }
        if (lane == 0) {
          red[wm * BN + BNM + t] = a1;
          red[(WAVES_M + wm) * BN + BNM + t] = a2;
        }
      }
    }
    __syncthreads();
    for (int c = tid; c < BN; c += 256) {
      float a1 = 0.f, a2 = 0.f;
#pragma unroll
      for (int w = 0; w < WAVES_M; ++w) {
        a1 += red[w * BN + c];
        a2 += red[(WAVES_M + w) * BN + c];
      }
      const int n = n0 + c;
      if (n < p.ldy) {
        p.stats[((size_t)tile_m * 2 + 0) * p.ldy + n] = a1;
        p.stats[((size_t)tile_m * 2 + 1) * p.ldy + n] = a2;
      }
    }
  } else if (p.stats != nullptr) {
    // per-tile (mean, M2) of every column in ONE pass and one barrier: every wave sums d = v - pivot and d*d over its
    // TM*16 rows, pivot = the column's value in the wave's first row (shifted sums: accurate when |mean| >> std);
    // the WAVES_M waves of a column are merged with Chan's formula.  (Two passes - mean, then squared deviations -
    // with a barrier each cost 57 of the 466 us of the 67 -> 33 full-resolution UP2 conv: tools/bench_up2.py.)
    float* red = smem;  // [3][WAVES_M][BN]: sum d, sum d*d, pivot; the staging tiles are dead after the last barrier
    const int nvalid = min(BM, p.M - m0);
    auto colval = [&](int i, int j, int r, float& v) -> bool {
      const int m = m0 + (wm * TM + i) * 16 + 4 * lq + r;
      const int n = n0 + (wn * TN + j) * 16 + l15;
      v = (n < p.Cout) ? act_fwd(acc[i][j][r] + bv[j], p.act) : 0.f;
      return m < p.M;
    };
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      float v0;
      colval(0, j, 0, v0);
      const float pv = quarter0_bcast(v0);  // lane l15 (lq = 0) holds row 0 of the wave's rows for this column
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v;
          if (colval(i, j, r, v)) {
            const float d = v - pv;
            s1 += d;
            s2 += d * d;
          }
        }
      s1 = quarter_sum(s1);
      s2 = quarter_sum(s2);
      if (lq == 0) {
        const int c = (wn * TN + j) * 16 + l15;
        red[(0 * WAVES_M + wm) * BN + c] = s1;
        red[(1 * WAVES_M + wm) * BN + c] = s2;
        red[(2 * WAVES_M + wm) * BN + c] = pv;
      }
    }
    if (NT > 0) {  // tail columns: lanes of quarter 0 hold one row each -> fold the 16 rows of the tile
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const float pv = __shfl(tv[0][t], 0, 64);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
          if (m0 + (wm * TM + i) * 16 + l15 < p.M) {
            const float d = tv[i][t] - pv;
            s1 += d;
            s2 += d * d;
          }
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
          s1 += __shfl_xor(s1, o, 64);
          s2 += __shfl_xor(s2, o, 64);
        }
        if (lane == 0) {
          red[(0 * WAVES_M + wm) * BN + BNM + t] = s1;
          red[(1 * WAVES_M + wm) * BN + BNM + t] = s2;
          red[(2 * WAVES_M + wm) * BN + BNM + t] = pv;
        }
      }
    }
    __syncthreads();
    if (nvalid == BM) {
      // full row block (all but the last one): WAVES_M groups of TM*16 rows each - the merge needs no division (the
      // general form below pays four fp32 divisions per wave, ~160 VALU instructions on the workgroup's last live wave)
      constexpr float inv_n = 1.f / (TM * 16), inv_w = 1.f / WAVES_M;
      for (int c = tid; c < BN; c += 256) {
        float mw[WAVES_M], mean = 0.f, m2 = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES_M; ++w) {
          const float s1 = red[(0 * WAVES_M + w) * BN + c], s2 = red[(1 * WAVES_M + w) * BN + c];
          mw[w] = red[(2 * WAVES_M + w) * BN + c] + s1 * inv_n;
          m2 += s2 - s1 * s1 * inv_n;
          mean += mw[w];
        }
        mean *= inv_w;
#pragma unroll
        for (int w = 0; w < WAVES_M; ++w) m2 += (TM * 16) * (mw[w] - mean) * (mw[w] - mean);
        const int n = n0 + c;
        if (n < p.ldy) {
          p.stats[((size_t)tile_m * 2 + 0) * p.ldy + n] = mean;
          p.stats[((size_t)tile_m * 2 + 1) * p.ldy + n] = m2;
        }
      }
      return;
    }
    for (int c = tid; c < BN; c += 256) {
      float cn = 0.f, mean = 0.f, m2 = 0.f;
#pragma unroll
      for (int w = 0; w < WAVES_M; ++w) {
        const float nw = (float)max(0, min(TM * 16, nvalid - w * TM * 16));  // valid rows of wave w
        if (nw > 0.f) {
          const float s1 = red[(0 * WAVES_M + w) * BN + c], s2 = red[(1 * WAVES_M + w) * BN + c];
          const float mw = red[(2 * WAVES_M + w) * BN + c] + s1 / nw;
          const float qw = s2 - s1 * s1 / nw;
          const float tot = cn + nw, d = mw - mean;
          mean += d * (nw / tot);
          m2 += qw + d * d * (cn * nw / tot);
          cn = tot;
        }
      }
      const int n = n0 + c;
      if (n < p.ldy) {
        p.stats[((size_t)tile_m * 2 + 0) * p.ldy + n] = mean;
        p.stats[((size_t)tile_m * 2 + 1) * p.ldy + n] = m2;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// host-side tile selection + launchers
// ---------------------------------------------------------------------------
struct TileCfg { int bm, bn; float eff; };
// id -> <TM, TN, WAVES_M, WAVES_N[, tail]>
//  0 <2,2,4,1> 128x32    1 <2,3,4,1> 128x48    2 <2,4,4,1> 128x64    3 <2,5,4,1> 128x80
//  4 <4,3,2,2> 128x96    5 <4,4,2,2> 128x128   6 <2,9,4,1> 128x144   7 <4,5,2,2> 128x160
//  8 <1,2,4,1> 64x32     9 <1,4,4,1> 64x64    10 <2,4,2,2> 64x128   11 <1,9,4,1> 64x144
// 12 <2,2,4,1,+4> 128x(32+4)   13 <2,1,4,1,+4> 128x(16+4)   14 <2,4,4,1,+4> 128x(64+4)   (VALU tail columns)
// 15 <2,1,4,1> 128x16
// eff = measured MFMA-rate of the tile relative to the 144-wide one on MI355X (tools/bench_conv.py):
// narrow tiles re-stage the A operand more often per MFMA.
static const TileCfg kTiles[] = {{128, 32, 0.62f}, {128, 48, 0.72f}, {128, 64, 0.80f}, {128, 80, 0.88f},
                                 {128, 96, 0.92f}, {128, 128, 0.95f}, {128, 144, 1.0f}, {128, 160, 1.0f},
                                 {64, 32, 0.50f},  {64, 64, 0.85f},  {64, 128, 0.75f},  {64, 144, 1.0f},
                                 {128, 36, 0.62f}, {128, 20, 0.43f}, {128, 68, 0.85f}, {128, 16, 0.38f}};
#define VMTL_NTILES 16
#define VMTL_NBIG 12
static const int kBigIds[] = {0, 1, 2, 3, 4, 5, 6, 7, 12, 13, 14, 15};
static const int kSmallIds[] = {8, 9, 10, 11};

static int pick_from(const int* ids, int n, int ncols) {
  // minimise issued MFMA work / tile efficiency = (padded columns) / eff
  int best = ids[0];
  float best_cost = -1.f;
  for (int k = 0; k < n; ++k) {
    const int id = ids[k];
    const float cost = (float)((long long)cdiv(ncols, kTiles[id].bn) * kTiles[id].bn) / kTiles[id].eff;
    if (best_cost < 0.f || cost < best_cost) {
      best = id;
      best_cost = cost;
    }
  }
  return best;
}

static bool conv_bf16x3();

// VMTL_FORCE_TILE=id: every launch uses that tile configuration (cached, see env_int)
static int forced_tile() {
  static EnvInt e{"VMTL_FORCE_TILE", -1};
  const int id = env_int(e);
  return (id >= 0 && id < VMTL_NTILES) ? id : -1;
}

static int conv_pick_tile(int M, int ncols) {
  const int forced = forced_tile();  // tuning aid (tools/bench_conv.py), never set in production
  if (forced >= 0) return forced;
  const int big = pick_from(kBigIds, VMTL_NBIG, ncols);
  // too few workgroups for 256 CUs: halve the row block
  if ((long long)cdiv(M, 128) * cdiv(ncols, kTiles[big].bn) < 384) return pick_from(kSmallIds, 4, ncols);
  // opt-in bf16x3 operands need 1.5x the LDS: only the 64-row tiles keep two workgroups per CU
  // (measured: 65536 x 135 x 1215 at 164 us with 64x144 against 198 us with 128x144)
  if (conv_bf16x3() && kTiles[big].bn >= 80) return pick_from(kSmallIds, 4, ncols);
  return big;
}

extern "C" int vmtl_conv2d_stats_rows(int B, int Ho, int Wo, int ldy) {
  return cdiv(B * Ho * Wo, kTiles[conv_pick_tile(B * Ho * Wo, ldy)].bm);
}

// output rows covered by each stats row block (the last block may be partial)
extern "C" int vmtl_conv2d_stats_block(int B, int Ho, int Wo, int ldy) {
  return kTiles[conv_pick_tile(B * Ho * Wo, ldy)].bm;
}

template <int TM, int TN, int WMV, int WNV, bool UP2, int NT, int NS, int X3 = 0>
static int launch_conv_ns(ConvP& p, hipStream_t st);

// opt-in bf16x3 operand split for the wide tiles (VMTL_BF16X3=1); see the kernel comment
static bool conv_bf16x3() {
  static EnvInt e{"VMTL_BF16X3", 0};  // cached; the parity tests toggle it through vmtl_reload_env()
  return env_int(e) == 1;
}

// LDS-DMA staging depth: 0 = register staging.  VMTL_GLDS overrides (tuning aid).
static int conv_glds_stages() {
  static EnvInt e{"VMTL_GLDS", 0};
  const int v = env_int(e);
  return (v == 2 || v == 3) ? v : 0;
}

template <int TM, int TN, int WMV, int WNV, bool UP2 = false, int NT = 0>
static int launch_conv(ConvP& p, hipStream_t st) {
  if constexpr (NT == 0 && WNV * TN >= 5) {
    // long K loops only: the three planes need 1.5x the LDS (one workgroup per CU for the 128-row tiles), which a
    // 4-6 step loop cannot amortise (MTAN's full-resolution 1x1 convs ran 40 % slower with it)
    if (conv_bf16x3() && cdiv(p.Ktot, BK) >= 24) return launch_conv_ns<TM, TN, WMV, WNV, UP2, NT, 0, 1>(p, st);
  }
  if constexpr (!UP2 && (WMV * TM) % 2 == 0) {
    const int ns = conv_glds_stages();
    if (ns == 2) return launch_conv_ns<TM, TN, WMV, WNV, UP2, NT, 2>(p, st);
    if (ns == 3) return launch_conv_ns<TM, TN, WMV, WNV, UP2, NT, 3>(p, st);
  }
  return launch_conv_ns<TM, TN, WMV, WNV, UP2, NT, 0>(p, st);
}

template <int TM, int TN, int WMV, int WNV, bool UP2, int NT, int NS, int X3>
static int launch_conv_ns(ConvP& p, hipStream_t st) {
  constexpr int BM = WMV * TM * 16, BN = WNV * TN * 16 + NT;
  p.tiles_m = cdiv(p.M, BM) * (UP2 ? 4 : 1);
  p.tiles_n = cdiv(p.shuffle ? p.Nw : p.ldy, BN);
  // buffer addressing: every operand is reached through a 32-bit byte offset
  if ((long long)p.B * p.H * p.W * p.Cs * 4 >= (1ll << 32) || (long long)p.Nw * p.Ktot * 4 >= (1ll << 32) ||
      (UP2 && (long long)16 * p.B * p.H * p.W * p.C2s >= (1ll << 32)))
    return VMTL_ERR_UNSUPPORTED;
  constexpr int NBUF = NS >= 2 ? NS : 2;
  constexpr int BNR = NS >= 2 ? (BN + 31) / 32 * 32 : BN;
  const size_t lds = X3 ? (size_t)2 * 3 * (BM + BN) * 64 : (size_t)NBUF * (BM + BNR) * LDT * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<TM, TN, WMV, WNV, UP2, NT, NS, X3>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((conv_igemm_kernel<TM, TN, WMV, WNV, UP2, NT, NS, X3>),
                     dim3(p.tiles_m * p.tiles_n, p.ksplit > 1 ? p.ksplit : 1), dim3(256), lds, st, p);
  return vmtl_check_launch();
}

// y = sum of the K-slice slabs (+ bias[n], n = column of a [rows][ldy] matrix; pad columns n >= Cout stay 0)
__global__ __launch_bounds__(256) void sum_slabs_kernel(const float* __restrict__ slabs, float* __restrict__ y,
                                                        int nslabs, long long n4, const float* __restrict__ bias,
                                                        int ldy, int Cout) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    f32x4 s = reinterpret_cast<const f32x4*>(slabs)[i];
    for (int z = 1; z < nslabs; ++z) s += reinterpret_cast<const f32x4*>(slabs)[(size_t)z * n4 + i];
    if (bias != nullptr) {
      const int n = (int)((i * 4) % ldy);
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (n + e < Cout) s[e] += bias[n + e];
    }
    reinterpret_cast<f32x4*>(y)[i] = s;
  }
}

// K slices a forward/dgrad launch of this shape would use (1 = none).  Only launches without
// bias/activation/stats/shuffle are split; the caller then passes a workspace of
// splits * B*Ho*Wo * ldy floats (vmtl_conv2d_fwd_ws) and the partial tiles are summed in slice order.
// K slices for a tile grid of `blocks` workgroups and nk K steps: split while the grid alone leaves the chip under-filled
// (fewer than VMTL_KSPLIT_BLOCKS workgroups, default 512 = two per CU: one 4-wave workgroup per CU is one wave per SIMD
// and hides no latency - decoder block 0 at bs 32 ran at 82 TF on 256 workgroups), at least 16 K steps per slice
static int ksplit_for(long long blocks, int nk, int ldy) {
  static EnvInt e_blocks{"VMTL_KSPLIT_BLOCKS", 512};  // tuning aid
  const int target = env_int(e_blocks);
  if (blocks >= target || nk < 32 || (ldy & 3)) return 1;
  long long s = cdivll(target, blocks);
  if (s > 8) s = 8;
  if (s > nk / 16) s = nk / 16;
  return s < 2 ? 1 : (int)s;
}

extern "C" int vmtl_conv2d_ksplit(int B, int Ho, int Wo, int ldy, int Ktot) {
  const int M = B * Ho * Wo;
  const int id = conv_pick_tile(M, ldy);
  return ksplit_for((long long)cdiv(M, kTiles[id].bm) * cdiv(ldy, kTiles[id].bn), cdiv(Ktot, BK), ldy);
}

static int conv2d_fwd_impl(const float* x, const float* wp, const float* bias, float* y, float* stats,
                           int B, int H, int W, int Cs, int Ho, int Wo, int ldy, int Nw, int Cout,
                           int KH, int KW, int stride, int pad, int act, int shuffle, const float* ez_x,
                           const float* ez_mean, const float* ez_invstd, const float* ez_gamma, const float* ez_beta,
                           int ez_act, void* stream) {
  VMTL_ENTER();
  if (!x || !wp || !y) return VMTL_ERR_ARG;
  if (Cs <= 0 || (Cs & 3) || B <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0) return VMTL_ERR_ARG;
  if (KH <= 0 || KW <= 0 || stride <= 0 || pad < 0 || Nw <= 0 || Cout <= 0 || ldy <= 0) return VMTL_ERR_ARG;
  if (!shuffle && (Cout > ldy || Nw > ldy)) return VMTL_ERR_ARG;
  if (shuffle && (Nw != 4 * Cout || Cout > ldy || stats)) return VMTL_ERR_ARG;
  // output extent must match the conv arithmetic
  if ((H + 2 * pad - KH) / stride + 1 != Ho || (W + 2 * pad - KW) / stride + 1 != Wo) return VMTL_ERR_ARG;
  if ((long long)B * Ho * Wo > 0x7fffffffLL || (long long)B * H * W > 0x7fffffffLL) return VMTL_ERR_ARG;
  ConvP p;
  p.x = x; p.wp = wp; p.bias = bias; p.y = y; p.stats = stats;
  p.B = B; p.H = H; p.W = W; p.Cs = Cs; p.Ho = Ho; p.Wo = Wo; p.ldy = ldy; p.Nw = Nw; p.Cout = Cout;
  p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad; p.Ktot = KH * KW * Cs; p.M = B * Ho * Wo;
  p.act = act; p.shuffle = shuffle; p.x2 = nullptr; p.C2s = 0; p.ksplit = 1; p.ksteps_per_split = 0;
  p.ez_x = ez_x; p.ez_mean = ez_mean; p.ez_invstd = ez_invstd; p.ez_gamma = ez_gamma; p.ez_beta = ez_beta; p.ez_act = ez_act;
  hipStream_t st = (hipStream_t)stream;
  if (shuffle && ldy > Cout &&  // the scatter only writes co < Cout: keep the pad-channel invariant
      hipMemsetAsync(y, 0, (size_t)B * 4 * Ho * Wo * ldy * sizeof(float), st) != hipSuccess)
    return VMTL_ERR_LAUNCH;
  switch (conv_pick_tile(p.M, shuffle ? Nw : ldy)) {
    case 0: return launch_conv<2, 2, 4, 1>(p, st);
    case 1: return launch_conv<2, 3, 4, 1>(p, st);
    case 2: return launch_conv<2, 4, 4, 1>(p, st);
    case 3: return launch_conv<2, 5, 4, 1>(p, st);
    case 4: return launch_conv<4, 3, 2, 2>(p, st);
    case 5: return launch_conv<4, 4, 2, 2>(p, st);
    case 6: return launch_conv<2, 9, 4, 1>(p, st);
    case 7: return launch_conv<4, 5, 2, 2>(p, st);
    case 8: return launch_conv<1, 2, 4, 1>(p, st);
    case 9: return launch_conv<1, 4, 4, 1>(p, st);
    case 10: return launch_conv<2, 4, 2, 2>(p, st);
    case 11: return launch_conv<1, 9, 4, 1>(p, st);
    case 12: return launch_conv<2, 2, 4, 1, false, 4>(p, st);
    case 13: return launch_conv<2, 1, 4, 1, false, 4>(p, st);
    case 14: return launch_conv<2, 4, 4, 1, false, 4>(p, st);
    default: return launch_conv<2, 1, 4, 1>(p, st);
  }
}

extern "C" int vmtl_conv2d_fwd(const float* x, const float* wp, const float* bias, float* y, float* stats,
                               int B, int H, int W, int Cs, int Ho, int Wo, int ldy, int Nw, int Cout,
                               int KH, int KW, int stride, int pad, int act, int shuffle, void* stream) {
  return conv2d_fwd_impl(x, wp, bias, y, stats, B, H, W, Cs, Ho, Wo, ldy, Nw, Cout, KH, KW, stride, pad, act, shuffle,
                         nullptr, nullptr, nullptr, nullptr, nullptr, 0, stream);
}

// Data gradient with the BatchNorm + activation backward of the producer of the differentiated tensor fused into
// the epilogue (ConvP::ez_x): y = dz, stats[vmtl_conv2d_stats_rows(...)][2][ldy] = per-row-block (sum dz, sum dz*xhat).
// The caller finishes with vmtl_bn_bwd_finalize + vmtl_bn_bwd_apply (no reduce pass over x and dy).
extern "C" int vmtl_conv2d_bnbwd(const float* x, const float* wp, float* y, float* stats, const float* ez_x,
                                 const float* ez_mean, const float* ez_invstd, const float* ez_gamma,
                                 const float* ez_beta, int ez_act, int B, int H, int W, int Cs, int Ho, int Wo, int ldy,
                                 int Nw, int Cout, int KH, int KW, int stride, int pad, void* stream) {
  if (!stats || !ez_x || !ez_mean || !ez_invstd || !ez_gamma || !ez_beta) return VMTL_ERR_ARG;
  return conv2d_fwd_impl(x, wp, nullptr, y, stats, B, H, W, Cs, Ho, Wo, ldy, Nw, Cout, KH, KW, stride, pad, 0, 0, ez_x,
                         ez_mean, ez_invstd, ez_gamma, ez_beta, ez_act, stream);
}

// split-K form of vmtl_conv2d_fwd for plain contractions (no bias / act / stats / shuffle): `ws` holds
// vmtl_conv2d_ksplit(...) * B*Ho*Wo*ldy floats.  With ksplit == 1 it is exactly vmtl_conv2d_fwd.
extern "C" int vmtl_conv2d_fwd_ws(const float* x, const float* wp, const float* bias, float* y, float* ws, int B, int H,
                                  int W, int Cs, int Ho, int Wo, int ldy, int Nw, int Cout, int KH, int KW, int stride,
                                  int pad, void* stream) {
  VMTL_ENTER();
  const int splits = vmtl_conv2d_ksplit(B, Ho, Wo, ldy, KH * KW * Cs);
  if (splits <= 1 || ws == nullptr)
    return vmtl_conv2d_fwd(x, wp, bias, y, nullptr, B, H, W, Cs, Ho, Wo, ldy, Nw, Cout, KH, KW, stride, pad, 0, 0, stream);
  if (!x || !wp || !y || Cs <= 0 || (Cs & 3) || Nw > ldy || Cout > ldy) return VMTL_ERR_ARG;
  if ((H + 2 * pad - KH) / stride + 1 != Ho || (W + 2 * pad - KW) / stride + 1 != Wo) return VMTL_ERR_ARG;
  ConvP p;
  p.x = x; p.wp = wp; p.bias = nullptr; p.y = ws; p.stats = nullptr; p.x2 = nullptr; p.C2s = 0;
  p.B = B; p.H = H; p.W = W; p.Cs = Cs; p.Ho = Ho; p.Wo = Wo; p.ldy = ldy; p.Nw = Nw; p.Cout = Cout;
  p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad; p.Ktot = KH * KW * Cs; p.M = B * Ho * Wo;
  p.act = 0; p.shuffle = 0; p.ez_x = nullptr; p.ez_mean = p.ez_invstd = p.ez_gamma = p.ez_beta = nullptr; p.ez_act = 0;
  p.ksplit = splits;
  p.ksteps_per_split = cdiv(cdiv(p.Ktot, BK), splits);
  hipStream_t st = (hipStream_t)stream;
  int rc;
  switch (conv_pick_tile(p.M, ldy)) {
    case 0: rc = launch_conv<2, 2, 4, 1>(p, st); break;
    case 1: rc = launch_conv<2, 3, 4, 1>(p, st); break;
    case 2: rc = launch_conv<2, 4, 4, 1>(p, st); break;
    case 3: rc = launch_conv<2, 5, 4, 1>(p, st); break;
    case 4: rc = launch_conv<4, 3, 2, 2>(p, st); break;
    case 5: rc = launch_conv<4, 4, 2, 2>(p, st); break;
    case 6: rc = launch_conv<2, 9, 4, 1>(p, st); break;
    case 7: rc = launch_conv<4, 5, 2, 2>(p, st); break;
    case 8: rc = launch_conv<1, 2, 4, 1>(p, st); break;
    case 9: rc = launch_conv<1, 4, 4, 1>(p, st); break;
    case 10: rc = launch_conv<2, 4, 2, 2>(p, st); break;
    case 11: rc = launch_conv<1, 9, 4, 1>(p, st); break;
    case 12: rc = launch_conv<2, 2, 4, 1, false, 4>(p, st); break;
    case 13: rc = launch_conv<2, 1, 4, 1, false, 4>(p, st); break;
    case 14: rc = launch_conv<2, 4, 4, 1, false, 4>(p, st); break;
    default: rc = launch_conv<2, 1, 4, 1>(p, st); break;
  }
  if (rc) return rc;
  const long long n4 = (long long)p.M * ldy / 4;
  long long nb = cdivll(n4, 256);
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(sum_slabs_kernel, dim3((int)nb), dim3(256), 0, st, ws, y, splits, n4, bias, ldy, Cout);
  return vmtl_check_launch();
}

// ---- nearest-x2 upsample + concat + 3x3 conv as four 2x2 phase convolutions ------------------------
// y[b, 2h+a, 2w+c, n] = sum_{ty,tx,ci} Weff[a][c][n][ty][tx][ci] * xl[b, h-1+a+ty, w-1+c+tx, ci]
//                     + sum_{dh,dw,cj} W[n][C0+cj][dh][dw] * skip[b, 2h+a+dh-1, 2w+c+dw-1, cj]
// (Weff = the 3x3 taps that fall on the same low-res pixel, pre-summed: vmtl_pack_up2_fwd).  Identical
// to conv3x3(cat[nearest2(xl), skip]) up to fp32 summation order, with 4 instead of 9 taps on the
// upsampled channels and no materialised upsample / concat tensor.
static int up2_pick_tile(int Mq, int ncols) {
  const int forced = forced_tile();
  if (forced >= 0) return forced;
  int big = pick_from(kBigIds, VMTL_NBIG, ncols);
  // 65-68 columns: the phase convs run faster on five MFMA column fragments (80 columns, 12 idle) than on four plus
  // four VALU tail columns - measured 364 vs 392 us on the 135+16 -> 67 layer (tools/bench_up2.py); the plain 3x3
  // layers of that width prefer the tail (241 vs 263 us, tools/bench_conv.py)
  if (big == 14) big = 3;
  if ((long long)cdiv(Mq, 128) * 4 * cdiv(ncols, kTiles[big].bn) < 384) return pick_from(kSmallIds, 4, ncols);
  if (conv_bf16x3() && kTiles[big].bn >= 80) return pick_from(kSmallIds, 4, ncols);
  return big;
}

extern "C" int vmtl_conv2d_up2_stats_block(int B, int H2, int W2, int ldy) {
  return kTiles[up2_pick_tile(B * H2 * W2, ldy)].bm;
}

// K slices of the phase-decomposed conv for this shape (1 = none): the deep decoder blocks at small batch are a few dozen
// workgroups with a K loop of 80-150 steps (decoder block 0 at bs 8: 64 workgroups, 21 TF before)
extern "C" int vmtl_conv2d_up2_ksplit(int B, int H2, int W2, int ldy, int Ktot) {
  const int Mq = B * H2 * W2;
  const int id = up2_pick_tile(Mq, ldy);
  return ksplit_for((long long)cdiv(Mq, kTiles[id].bm) * 4 * cdiv(ldy, kTiles[id].bn), cdiv(Ktot, BK), ldy);
}

static int up2_launch(ConvP& p, int id, hipStream_t st);

// stats (optional): [4 * ceil(B*H2*W2 / block)][2][ldy]; only valid when block divides B*H2*W2
extern "C" int vmtl_conv2d_up2_fwd(const float* xl, const float* skip, const float* wp_eff, float* y, float* stats,
                                   int B, int H2, int W2, int C0s, int C1s, int ldy, int Cout, void* stream) {
  VMTL_ENTER();
  if (!xl || !wp_eff || !y || B <= 0 || H2 <= 0 || W2 <= 0 || C0s <= 0 || (C0s & 3) || (C1s & 3) || C1s < 0)
    return VMTL_ERR_ARG;
  if ((skip == nullptr) != (C1s == 0) || Cout <= 0 || Cout > ldy) return VMTL_ERR_ARG;
  if ((long long)B * H2 * W2 * 4 > 0x7fffffffLL) return VMTL_ERR_ARG;
  ConvP p;
  p.x = xl; p.x2 = skip; p.wp = wp_eff; p.bias = nullptr; p.y = y; p.stats = stats;
  p.B = B; p.H = H2; p.W = W2; p.Cs = C0s; p.C2s = C1s; p.Ho = H2; p.Wo = W2; p.ldy = ldy; p.Nw = Cout;
  p.Cout = Cout; p.KH = 2; p.KW = 2; p.stride = 1; p.pad = 0; p.Ktot = 4 * C0s + 9 * C1s; p.M = B * H2 * W2;
  p.act = 0; p.shuffle = 0; p.ksplit = 1; p.ksteps_per_split = 0;
  p.ez_x = nullptr; p.ez_mean = p.ez_invstd = p.ez_gamma = p.ez_beta = nullptr; p.ez_act = 0;
  const int id = up2_pick_tile(p.M, ldy);
  if (stats && (p.M % kTiles[id].bm)) return VMTL_ERR_ARG;
  return up2_launch(p, id, (hipStream_t)stream);
}

// split-K form (no statistics): ws = vmtl_conv2d_up2_ksplit(...) * B*2H2*2W2*ldy floats; the slices' partial outputs
// (full-resolution layout each) are summed in slice order
extern "C" int vmtl_conv2d_up2_fwd_ws(const float* xl, const float* skip, const float* wp_eff, float* y, float* ws, int B,
                                      int H2, int W2, int C0s, int C1s, int ldy, int Cout, void* stream) {
  VMTL_ENTER();
  const int splits = vmtl_conv2d_up2_ksplit(B, H2, W2, ldy, 4 * C0s + 9 * C1s);
  if (splits <= 1 || ws == nullptr)
    return vmtl_conv2d_up2_fwd(xl, skip, wp_eff, y, nullptr, B, H2, W2, C0s, C1s, ldy, Cout, stream);
  if (!xl || !wp_eff || !y || B <= 0 || H2 <= 0 || W2 <= 0 || C0s <= 0 || (C0s & 3) || (C1s & 3) || C1s < 0)
    return VMTL_ERR_ARG;
  if ((skip == nullptr) != (C1s == 0) || Cout <= 0 || Cout > ldy || (ldy & 3)) return VMTL_ERR_ARG;
  if ((long long)B * H2 * W2 * 4 > 0x7fffffffLL) return VMTL_ERR_ARG;
  ConvP p;
  p.x = xl; p.x2 = skip; p.wp = wp_eff; p.bias = nullptr; p.y = ws; p.stats = nullptr;
  p.B = B; p.H = H2; p.W = W2; p.Cs = C0s; p.C2s = C1s; p.Ho = H2; p.Wo = W2; p.ldy = ldy; p.Nw = Cout;
  p.Cout = Cout; p.KH = 2; p.KW = 2; p.stride = 1; p.pad = 0; p.Ktot = 4 * C0s + 9 * C1s; p.M = B * H2 * W2;
  p.act = 0; p.shuffle = 0; p.ksplit = splits; p.ksteps_per_split = cdiv(cdiv(p.Ktot, BK), splits);
  p.ez_x = nullptr; p.ez_mean = p.ez_invstd = p.ez_gamma = p.ez_beta = nullptr; p.ez_act = 0;
  hipStream_t st = (hipStream_t)stream;
  const int rc = up2_launch(p, up2_pick_tile(p.M, ldy), st);
  if (rc) return rc;
  const long long n4 = (long long)4 * p.M * ldy / 4;
  long long nb = cdivll(n4, 256);
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(sum_slabs_kernel, dim3((int)nb), dim3(256), 0, st, ws, y, splits, n4, (const float*)nullptr, ldy, Cout);
  return vmtl_check_launch();
}

static int up2_launch(ConvP& p, int id, hipStream_t st) {
  switch (id) {
    case 0: return launch_conv<2, 2, 4, 1, true>(p, st);
    case 1: return launch_conv<2, 3, 4, 1, true>(p, st);
    case 2: return launch_conv<2, 4, 4, 1, true>(p, st);
    case 3: return launch_conv<2, 5, 4, 1, true>(p, st);
    case 4: return launch_conv<4, 3, 2, 2, true>(p, st);
    case 5: return launch_conv<4, 4, 2, 2, true>(p, st);
    case 6: return launch_conv<2, 9, 4, 1, true>(p, st);
    case 7: return launch_conv<4, 5, 2, 2, true>(p, st);
    case 8: return launch_conv<1, 2, 4, 1, true>(p, st);
    case 9: return launch_conv<1, 4, 4, 1, true>(p, st);
    case 10: return launch_conv<2, 4, 2, 2, true>(p, st);
    case 11: return launch_conv<1, 9, 4, 1, true>(p, st);
    case 12: return launch_conv<2, 2, 4, 1, true, 4>(p, st);
    case 13: return launch_conv<2, 1, 4, 1, true, 4>(p, st);
    case 14: return launch_conv<2, 4, 4, 1, true, 4>(p, st);
    default: return launch_conv<2, 1, 4, 1, true>(p, st);
  }
}
