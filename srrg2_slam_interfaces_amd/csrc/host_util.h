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
  T* p          = nullptr;
  size_t cap    = 0;      // elements
  bool borrowed = false;  // p belongs to another DevBuf (a slice that shares the clouds of another slice)
  // view of another buffer: no ownership; the lender must outlive the use (re-borrowed at every compute())
  void borrow(const DevBuf<T>& o) {
    if (!borrowed) release();
    p        = o.p;
    cap      = o.cap;
    borrowed = o.p != nullptr;
  }
  int reserve(size_t n) {
    if (borrowed) {
      p        = nullptr;
      cap      = 0;
      borrowed = false;
    }
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
    if (p && !borrowed) (void) hipFree(p);
    p        = nullptr;
    cap      = 0;
    borrowed = false;
  }
};

// what scene.hip needs to see of an aligner slice after compute() (device pointers into the aligner's buffers)
struct AlignerSliceView {
  const float4* moving_sorted;  // .w = caller's index (= index in the cloud given to set_moving)
  const int* corr_fixed;        // per sorted moving point: matched fixed index or -1
  const float* corr_resp;
  const unsigned char* corr_stat;
  int nm, nf;
  bool prune;  // keep_only_inlier_correspondences && Success: only inlier correspondences count
  int device;
  void* ready_event;  // hipEvent_t recorded on the aligner's stream behind the kernels that wrote the arrays: wait for it on the
                      // consuming stream (hipStreamWaitEvent) before reading them
};
int aligner_slice_view(srrg2_aligner_s* a, int slice_idx, AlignerSliceView* v);

}  // namespace srrg2amd

#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess) {                                                                        \
      return srrg2amd::fail(SRRG2_E_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));       \
    }                                                                                              \
  } while (0)
