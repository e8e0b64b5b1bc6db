// Library identity; also the place where the public header is compiled against the
// definitions so a signature drift fails the build instead of the loader.
#include "common.h"
#include "../../include/vmtl.h"

extern "C" const char* vmtl_version(void) { return "vmtl 0.1 (gfx950)"; }

thread_local int vmtl_last_hip_error = 0;

int vmtl_env_epoch = 1;
// re-read the VMTL_* tuning overrides on their next use (they are cached after the first read)
extern "C" int vmtl_reload_env(void) { return ++vmtl_env_epoch; }

// HIP's description of the last launch failure seen on this thread (status -2), for error messages
extern "C" const char* vmtl_last_error_string(void) { return hipGetErrorString((hipError_t)vmtl_last_hip_error); }

// Timeline probe for tuning multi-stream schedules: writes the 100 MHz wall clock when the stream reaches it
// (rocprofv3's kernel trace perturbs cross-queue overlap, a one-thread kernel barely does).
__global__ void timestamp_kernel(long long* out) { *out = (long long)wall_clock64(); }

extern "C" int vmtl_timestamp(long long* out, void* stream) {
  VMTL_ENTER();
  if (!out) return VMTL_ERR_ARG;
  hipLaunchKernelGGL(timestamp_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, out);
  return vmtl_check_launch();
}
