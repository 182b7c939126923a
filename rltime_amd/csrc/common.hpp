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

// ---------------------------------------------------------------------------
// Per-kernel timing with HIP events ON THE LAUNCH STREAM (mirl_profile_*).
// level 0: off (one predictable branch per launch); level >= 2: every
// librltime_hip launch is bracketed by an event pair and accounted under its
// kernel name together with the ALGORITHMIC bytes the call site states for it
// (DESIGN.md section 3), so that bench.py can print achieved GB/s per kernel
// against the HBM roofline.  Event pairs are resolved by mirl_profile_collect()
// (which synchronises); nothing in the hot path waits.
struct ProfEntry { std::string name; int64_t calls = 0; double ms = 0.0, bytes = 0.0, flop = 0.0; };
class Profiler {
 public:
  int level = 0;
  std::vector<ProfEntry> entries;
  struct Pending { int idx; hipEvent_t a, b; };
  std::vector<Pending> pending;
  std::vector<hipEvent_t> pool;
  int index_of(const char* name) {
    for (size_t i = 0; i < entries.size(); ++i) if (entries[i].name == name) return (int)i;
    entries.push_back(ProfEntry{name}); return (int)entries.size() - 1;
  }
  hipEvent_t get() {
    if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
    hipEvent_t e = nullptr; (void)hipEventCreate(&e); return e;
  }
  void collect() {
    (void)hipDeviceSynchronize();
    for (Pending& p : pending) {
      float ms = 0.f;
      if (p.a && p.b && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) { entries[(size_t)p.idx].ms += ms; ++entries[(size_t)p.idx].calls; }
      if (p.a) pool.push_back(p.a);
      if (p.b) pool.push_back(p.b);
    }
    pending.clear();
  }
  void reset() { collect(); entries.clear(); }
};
Profiler& profiler();

struct ProfScope {
  int idx = -1; hipEvent_t a = nullptr; hipStream_t st;
  // bytes: ALGORITHMIC HBM bytes of the launch; flop: its matrix-pipe work (0 for streaming / latency kernels)
  ProfScope(const char* name, double bytes, hipStream_t s, double flop = 0.0) : st(s) {
    Profiler& p = profiler();
    if (p.level < 2) return;
    // never inside a stream capture: an event-record node on a pooled timing event would be baked into the graph and
    // re-recorded on every replay, and the elapsed-time query of the pair would read whatever replay came last
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return;
    idx = p.index_of(name);
    p.entries[(size_t)idx].bytes += bytes;
    p.entries[(size_t)idx].flop += flop;
    a = p.get();
    (void)hipEventRecord(a, st);
  }
  ~ProfScope() {
    if (idx < 0) return;
    Profiler& p = profiler();
    hipEvent_t b = p.get();
    (void)hipEventRecord(b, st);
    p.pending.push_back(Profiler::Pending{idx, a, b});
  }
};

}  // namespace mirl
