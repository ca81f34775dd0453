// Library-level pieces of libtdvc_hip.so: ABI version, thread-local error text, per-device scratch pages.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include <mutex>

#include "../../include/tdvc_hip.h"

static thread_local char g_err[512] = "";

void tdvc_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int tdvc_abi_version(void) { return 4; }   // 4: tdvc_conv_desc::chan_sum; 3: tdvc_prepare_device, SE-pool conv epilogue, deterministic col2im
extern "C" const char* tdvc_last_error(void) { return g_err; }

// ---- per-device scratch: a page of zeros nobody writes (DMA source of out-of-image halo pixels) and a dump page nobody
// reads (store target of lanes outside a strip).  One allocation per device, created under a mutex on the first launch that
// needs it on THAT device (a process driving several GPUs gets one page each); the allocation is synchronous, so a caller
// that captures launches into a graph calls tdvc_prepare_device() once before the capture.
namespace {
constexpr int kMaxDev = 64;
constexpr size_t kZeroBytes = 4096, kDumpBytes = 16384;
std::mutex g_scratch_mu;
unsigned char* g_scratch[kMaxDev] = {};
}  // namespace

// -> 0 and the two pointers, or a HIP error code (tdvc_last_error() says which)
int tdvc_scratch_pages(const void** zeros, void** dump) {
  int dev = 0;
  hipError_t err = hipGetDevice(&dev);
  if (err != hipSuccess || dev < 0 || dev >= kMaxDev) {
    tdvc_set_error("scratch pages: hipGetDevice failed or device index %d out of range: %s", dev, hipGetErrorString(err));
    return err != hipSuccess ? (int)err : TDVC_EINVAL;
  }
  std::lock_guard<std::mutex> lk(g_scratch_mu);
  if (!g_scratch[dev]) {
    unsigned char* p = nullptr;
    err = hipMalloc(reinterpret_cast<void**>(&p), kZeroBytes + kDumpBytes);
    if (err == hipSuccess) err = hipMemset(p, 0, kZeroBytes + kDumpBytes);
    if (err == hipSuccess) err = hipDeviceSynchronize();      // the fill runs on the null stream: complete before a launch on ANY stream reads the page
    if (err != hipSuccess) {
      if (p) (void)hipFree(p);
      tdvc_set_error("scratch pages: allocation on device %d failed: %s", dev, hipGetErrorString(err));
      return (int)err;
    }
    g_scratch[dev] = p;
  }
  if (zeros) *zeros = g_scratch[dev];
  if (dump) *dump = g_scratch[dev] + kZeroBytes;
  return TDVC_OK;
}

extern "C" int tdvc_prepare_device(void) { return tdvc_scratch_pages(nullptr, nullptr); }
