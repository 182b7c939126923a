// common.hpp — error plumbing and the pinned->device parameter staging ring.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>

#include "../../include/mirl.h"

namespace mirl {

std::string& last_error_ref();
inline int fail(int code, const std::string& msg) { last_error_ref() = msg; return code; }

#define MIRL_HIP(call)                                                         \
  do {                                                                         \
    hipError_t _e = (call);                                                    \
    if (_e != hipSuccess)                                                      \
      return ::mirl::fail(MIRL_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(_e)); \
  } while (0)

#define MIRL_LAUNCH_CHECK()                                                    \
  do {                                                                         \
    hipError_t _e = hipGetLastError();                                         \
    if (_e != hipSuccess)                                                      \
      return ::mirl::fail(MIRL_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(_e)); \
  } while (0)

// Small host->device parameter blocks (op lists, rng draws) travel through a
// ring of pinned host buffers with a matching device buffer each, copied with
// hipMemcpyAsync on the caller's stream; an event per block guards reuse so
// the host never waits unless it laps the ring.
class StagingRing {
 public:
  static const int kBlocks = 32;
  int init() {
    for (int i = 0; i < kBlocks; ++i) { host_[i] = nullptr; dev_[i] = nullptr; cap_[i] = 0; used_[i] = false; ev_[i] = nullptr; }
    for (int i = 0; i < kBlocks; ++i) MIRL_HIP(hipEventCreateWithFlags(&ev_[i], hipEventDisableTiming));
    return MIRL_OK;
  }
  void destroy() {
    for (int i = 0; i < kBlocks; ++i) {
      if (ev_[i]) { (void)hipEventSynchronize(ev_[i]); (void)hipEventDestroy(ev_[i]); }
      if (host_[i]) (void)hipHostFree(host_[i]);
      if (dev_[i]) (void)hipFree(dev_[i]);
    }
  }
  // Returns a block with at least `bytes` capacity; host pointer in *h, device in *d.
  int acquire(size_t bytes, char** h, char** d) {
    int i = next_; next_ = (next_ + 1) % kBlocks; cur_ = i;
    if (used_[i]) MIRL_HIP(hipEventSynchronize(ev_[i]));
    if (bytes > cap_[i]) {
      if (host_[i]) MIRL_HIP(hipHostFree(host_[i]));
      if (dev_[i]) MIRL_HIP(hipFree(dev_[i]));
      size_t cap = 4096; while (cap < bytes) cap *= 2;
      MIRL_HIP(hipHostMalloc((void**)&host_[i], cap, hipHostMallocDefault));
      MIRL_HIP(hipMalloc((void**)&dev_[i], cap));
      cap_[i] = cap;
    }
    *h = host_[i]; *d = dev_[i];
    return MIRL_OK;
  }
  // Upload the first `bytes` of the current block and mark it in flight.  Call
  // mark() again after the kernels that read the device block were enqueued.
  int upload(size_t bytes, hipStream_t s) {
    if (bytes) MIRL_HIP(hipMemcpyAsync(dev_[cur_], host_[cur_], bytes, hipMemcpyHostToDevice, s));
    return MIRL_OK;
  }
  int mark(hipStream_t s) {
    MIRL_HIP(hipEventRecord(ev_[cur_], s));
    used_[cur_] = true;
    return MIRL_OK;
  }
 private:
  char* host_[kBlocks]; char* dev_[kBlocks]; size_t cap_[kBlocks]; bool used_[kBlocks]; hipEvent_t ev_[kBlocks];
  int next_ = 0, cur_ = 0;
};

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace mirl
