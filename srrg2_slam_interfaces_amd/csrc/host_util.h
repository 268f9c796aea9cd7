// host_util.h -- small host-side helpers shared by the drivers (error slot, HIP error check, device buffer)
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/srrg2_slam_amd.h"

namespace srrg2amd {

extern thread_local std::string g_err;
int fail(int code, const std::string& msg);

template <typename T>
struct DevBuf {
  T* p       = nullptr;
  size_t cap = 0;  // elements
  int reserve(size_t n) {
    if (n <= cap) return 0;
    if (p) (void) hipFree(p);
    p   = nullptr;
    cap = 0;
    size_t want = n + n / 8 + 16;
    hipError_t e = hipMalloc((void**) &p, want * sizeof(T));
    if (e != hipSuccess) return fail(SRRG2_E_HIP, std::string("hipMalloc: ") + hipGetErrorString(e));
    cap = want;
    return 0;
  }
  void release() {
    if (p) (void) hipFree(p);
    p   = nullptr;
    cap = 0;
  }
};

}  // namespace srrg2amd

#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess) {                                                                        \
      return srrg2amd::fail(SRRG2_E_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));       \
    }                                                                                              \
  } while (0)
