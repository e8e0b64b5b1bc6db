// Host side of vmtl_conv3x3_small (kernel and design notes: conv_small.h; instantiations: conv_small_cs*.hip).
#include "conv_small.h"

int vmtl_small_launch_cs16(SmallP& p, hipStream_t st);
int vmtl_small_launch_cs20(SmallP& p, hipStream_t st);
int vmtl_small_launch_cs32(SmallP& p, hipStream_t st);
int vmtl_small_launch_cs36(SmallP& p, hipStream_t st);

// ---------------------------------------------------------------------------------------------- host side
int small_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        v <= 0)
      v = 256;
    n = v;
  }
  return n;
}

// Persistent grid: two workgroups per CU.  With a statistics epilogue every workgroup must see the same number of
// tiles (the BatchNorm finalize weights all rows equally), so the grid is the largest divisor of the tile count that
// still fills at least half the slots; a workgroup then ACCUMULATES its tiles into one statistics row (512 rows
// instead of 8192 for 32 x 128 x 256: the one-workgroup-per-channel finalize kernels took 33 us on 8192 rows).
// Shapes without such a divisor keep one row per tile.
int small_grid(int ntiles, int* acc_rows) {
  const int slots = 2 * small_cus();
  int best = 0;
  for (int g = slots < ntiles ? slots : ntiles; g >= 1; --g)
    if (ntiles % g == 0) { best = g; break; }
  if (acc_rows) *acc_rows = 0;
  static EnvInt e_acc{"VMTL_SMALL_ACC", 1};  // VMTL_SMALL_ACC=0: tuning aid, one statistics row per tile
  const int allow = env_int(e_acc);
  if (allow && best * 2 >= (slots < ntiles ? slots : ntiles)) {
    if (acc_rows) *acc_rows = 1;
    return best;
  }
  return slots < ntiles ? slots : ntiles;
}

// 1 when vmtl_conv3x3_small handles a 3x3/s1/p1 conv with Cs input storage channels and Nw weight rows
extern "C" int vmtl_conv3x3_small_supported(int Cs, int Nw) {
  return (Cs == 20 || Cs == 36 || Cs == 32 || Cs == 16) && Nw >= 1 && Nw <= 36;
}

// statistics rows (= tiles); the statistics epilogues need full tiles: H % 4 == 0 and W % 32 == 0
extern "C" int vmtl_conv3x3_small_tiles(int B, int H, int W) { return B * cdiv(H, CSM_TH) * cdiv(W, CSM_TW); }

// rows of the statistics tensor [rows][2][ldy] written by ep_mode 1 / 2, and the pixels each row covers
extern "C" int vmtl_conv3x3_small_stat_rows(int B, int H, int W) {
  int acc = 0;
  const int nt = vmtl_conv3x3_small_tiles(B, H, W), g = small_grid(nt, &acc);
  return acc ? g : nt;
}

extern "C" int vmtl_conv3x3_small_stat_block(int B, int H, int W) {
  int acc = 0;
  const int nt = vmtl_conv3x3_small_tiles(B, H, W), g = small_grid(nt, &acc);
  return CSM_TH * CSM_TW * (acc ? nt / g : 1);
}

extern "C" int vmtl_conv3x3_small(const float* x, const float* x2, const float* pa, const float* pb, const float* pc,
                                  int act_in, float* a_out, const float* wp, const float* bias, float* y, float* yb,
                                  int Ca, float* stats, int ep_mode, const float* ez_x, const float* ez_mean,
                                  const float* ez_invstd, const float* ez_gamma, const float* ez_beta, int ez_act, int B,
                                  int H, int W, int Cs, int ldy, int Nw, int Cout, void* stream) {
  VMTL_ENTER();
  if (!x || !wp || !y || B <= 0 || H <= 0 || W <= 0 || ldy <= 0 || (ldy & 3)) return VMTL_ERR_ARG;
  if (!vmtl_conv3x3_small_supported(Cs, Nw) || Cout <= 0 || Cout > Nw || Nw > ldy || ldy > 36) return VMTL_ERR_ARG;
  if ((pa == nullptr) != (pc == nullptr) || (x2 != nullptr && (pa == nullptr || pb == nullptr))) return VMTL_ERR_ARG;
  if (a_out != nullptr && pa == nullptr) return VMTL_ERR_ARG;
  if (act_in != VMTL_ACT_NONE && act_in != VMTL_ACT_RELU) return VMTL_ERR_ARG;
  if (ep_mode < 0 || ep_mode > 2 || (ep_mode != 0 && (!stats || (H % CSM_TH) || (W % CSM_TW)))) return VMTL_ERR_ARG;
  if (ep_mode == 2 && (!ez_x || !ez_mean || !ez_invstd || !ez_gamma || !ez_beta)) return VMTL_ERR_ARG;
  // (mode 1 + bias is fine: the bias is added before the statistics are taken, as BatchNorm(conv(x) + b) needs)
  if ((ep_mode == 2 && bias != nullptr) || (ep_mode != 0 && yb != nullptr)) return VMTL_ERR_ARG;
  if (yb != nullptr && (Ca <= 0 || Ca >= Cout || (W & 3))) return VMTL_ERR_ARG;
  if ((long long)B * H * W * 36 > 0x7fffffffLL) return VMTL_ERR_UNSUPPORTED;  // 32-bit element offsets in the kernel
  SmallP p;
  p.x = x; p.x2 = x2; p.pa = pa; p.pb = pb; p.pc = pc; p.a_out = a_out; p.wp = wp; p.bias = bias; p.y = y; p.yb = yb;
  p.stats = stats; p.ez_x = ez_x; p.ez_mean = ez_mean; p.ez_invstd = ez_invstd; p.ez_gamma = ez_gamma; p.ez_beta = ez_beta;
  p.act_in = act_in; p.ep_mode = ep_mode; p.ez_act = ez_act; p.Ca = Ca;
  p.B = B; p.H = H; p.W = W; p.ldy = ldy; p.Nw = Nw; p.Cout = Cout;
  p.tiles_x = cdiv(W, CSM_TW); p.tiles_y = cdiv(H, CSM_TH); p.ntiles = B * p.tiles_x * p.tiles_y;
  p.grid = small_grid(p.ntiles, &p.acc_rows);
#ifdef VMTL_TUNING  // ablation switches (WRONG results): only in a -DVMTL_TUNING build, never in the shipped library
  static EnvInt e_dbg{"VMTL_SMALL_DBG", 0};
  p.dbg = env_int(e_dbg);
#else
  p.dbg = 0;
#endif
  hipStream_t st = (hipStream_t)stream;
  if (ldy > (Nw > 20 ? 36 : 20)) return VMTL_ERR_ARG;  // the output tile holds 16 / 32 MFMA columns + 4
  switch (Cs) {
    case 36: return vmtl_small_launch_cs36(p, st);
    case 20: return vmtl_small_launch_cs20(p, st);
    case 32: return vmtl_small_launch_cs32(p, st);
    default: return vmtl_small_launch_cs16(p, st);
  }
}
