// Per-step segmentation metrics from ONE pass over (argmax predictions, targets): a CxC confusion
// matrix built in LDS, then accuracy / Jaccard / F-beta derived on the device (no host sync).
//
// Replaces the four torchmetrics 0.7.3 updates of reference vision_mtl/lit_module.py:48-69,106-118
// (Accuracy average="micro"; FBetaScore beta=1 average="weighted" mdmc_average="global";
// JaccardIndex absent_score=0, mean over classes).  torchmetrics is not installable offline, so
// the formulas are restated from its documented definitions ("parity unpinned").
#include "common.h"

#define CM_MAX_C 64

__global__ __launch_bounds__(256) void confusion_kernel(const long long* __restrict__ pred,
                                                        const long long* __restrict__ tgt, int* __restrict__ cm,
                                                        long long P, int C) {
  extern __shared__ int hist[];
  for (int i = threadIdx.x; i < C * C; i += blockDim.x) hist[i] = 0;
  __syncthreads();
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (long long)gridDim.x * blockDim.x) {
    const long long t = tgt[i], p = pred[i];
    if (t >= 0 && t < C && p >= 0 && p < C) atomicAdd(&hist[(int)t * C + (int)p], 1);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < C * C; i += blockDim.x)
    if (hist[i]) atomicAdd(&cm[i], hist[i]);  // integer atomics: exact and order independent
}

// out[0] accuracy (micro), out[1] Jaccard (mean over classes, absent -> 0), out[2] F-beta (support-weighted)
__global__ void segm_metrics_kernel(const int* __restrict__ cm, int C, float beta, float* out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double total = 0.0, correct = 0.0, jac = 0.0, fb = 0.0, support_sum = 0.0;
  const double b2 = (double)beta * beta;
  for (int c = 0; c < C; ++c) {
    double tp = cm[c * C + c], row = 0.0, col = 0.0;
    for (int k = 0; k < C; ++k) {
      row += cm[c * C + k];  // targets of class c
      col += cm[k * C + c];  // predictions of class c
    }
    const double fn = row - tp, fp = col - tp;
    total += row;
    correct += tp;
    const double uni = tp + fp + fn;
    jac += uni > 0.0 ? tp / uni : 0.0;
    const double den = (1.0 + b2) * tp + b2 * fn + fp;
    fb += row * (den > 0.0 ? (1.0 + b2) * tp / den : 0.0);
    support_sum += row;
  }
  out[0] = (float)(total > 0.0 ? correct / total : 0.0);
  out[1] = (float)(jac / C);
  out[2] = (float)(support_sum > 0.0 ? fb / support_sum : 0.0);
}

extern "C" int vmtl_confusion_matrix(const long long* pred, const long long* target, int* cm, long long P, int C,
                                     void* stream) {
  VMTL_ENTER();
  if (!pred || !target || !cm || P <= 0 || C <= 0 || C > CM_MAX_C) return VMTL_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(cm, 0, (size_t)C * C * sizeof(int), st) != hipSuccess) return VMTL_ERR_LAUNCH;
  long long nb = cdivll(P, 256 * 8);
  if (nb > 1024) nb = 1024;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL(confusion_kernel, dim3((int)nb), dim3(256), (size_t)C * C * sizeof(int), st, pred, target, cm, P,
                     C);
  return vmtl_check_launch();
}

extern "C" int vmtl_segm_metrics(const int* cm, int C, float beta, float* out, void* stream) {
  VMTL_ENTER();
  if (!cm || !out || C <= 0 || C > CM_MAX_C) return VMTL_ERR_ARG;
  hipLaunchKernelGGL(segm_metrics_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, cm, C, beta, out);
  return vmtl_check_launch();
}
