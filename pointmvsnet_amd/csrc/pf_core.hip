// Library-level entry points: version, errors, device info, sticky status word.
#include <string.h>

#include "pf_common.h"

namespace {
constexpr int kMaxDevices = 64;
unsigned* g_status[kMaxDevices] = {nullptr};
}  // namespace

unsigned* pf_status_ptr() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return nullptr;
  if (g_status[dev] == nullptr) {
    unsigned* p = nullptr;
    if (hipMalloc(&p, sizeof(unsigned)) != hipSuccess) return nullptr;
    if (hipMemset(p, 0, sizeof(unsigned)) != hipSuccess) return nullptr;
    g_status[dev] = p;
  }
  return g_status[dev];
}

namespace {
__global__ void timestamp_kernel(long long* out) { *out = (long long)wall_clock64(); }
}  // namespace

extern "C" {

int pf_debug_timestamp(long long* slot, void* stream) {
  PF_REQUIRE(slot != nullptr);
  hipLaunchKernelGGL(timestamp_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, slot);
  return pf_launch_status();
}

const char* pf_version(void) { return "pointflow_hip 0.1 (gfx950)"; }

const char* pf_error_string(int code) {
  if (code == PF_OK) return "ok";
  if (code == PF_ERR_INVALID_ARG) return "invalid argument (shape / pointer / limit check failed; nothing launched)";
  if (code == PF_ERR_UNSUPPORTED) return "unsupported configuration (see include/pointflow_hip.h limits)";
  if (code > 0) return hipGetErrorString((hipError_t)code);
  return "unknown pointflow error";
}

int pf_device_info(int* cu_count, int* lds_bytes_per_block, char* arch_host, int arch_len) {
  int dev = 0;
  PF_HIP(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  PF_HIP(hipGetDeviceProperties(&prop, dev));
  if (cu_count) *cu_count = prop.multiProcessorCount;
  if (lds_bytes_per_block) *lds_bytes_per_block = (int)prop.sharedMemPerBlock;
  if (arch_host && arch_len > 0) {
    strncpy(arch_host, prop.gcnArchName, (size_t)arch_len - 1);
    arch_host[arch_len - 1] = '\0';
  }
  return PF_OK;
}

int pf_check_status(unsigned* status_host, void* stream) {
  PF_REQUIRE(status_host != nullptr);
  unsigned* p = pf_status_ptr();
  if (p == nullptr) return (int)hipErrorNotInitialized;
  hipStream_t s = (hipStream_t)stream;
  PF_HIP(hipMemcpyAsync(status_host, p, sizeof(unsigned), hipMemcpyDeviceToHost, s));
  PF_HIP(hipStreamSynchronize(s));
  PF_HIP(hipMemsetAsync(p, 0, sizeof(unsigned), s));
  PF_HIP(hipStreamSynchronize(s));
  return PF_OK;
}

}  // extern "C"
