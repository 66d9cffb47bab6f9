// fvdb_internal.h — structures shared by the translation units of libfvdb_hip.so (not part of the C ABI).
#pragma once
#include "../../include/fvdb.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <mutex>
#include <string>
#include <vector>

struct DBuf {
  void* p = nullptr;
  size_t cap = 0;
  hipError_t ensure(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = std::max<size_t>(bytes + bytes / 4, 256);
    hipError_t e = hipMalloc(&p, want);
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
  template <typename T>
  T* as() const { return (T*)p; }
};

struct HBuf {  // pinned host staging
  void* p = nullptr;
  size_t cap = 0;
  hipError_t ensure(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
    size_t want = std::max<size_t>(bytes + bytes / 4, 4096);
    hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() {
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
  }
};

struct fvdb_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  int num_cus = 256;
  std::string err;   // last failure; written under err_mu (searches may fail on several host threads)
  std::mutex err_mu;
  void set_err(std::string m) {
    std::lock_guard<std::mutex> lk(err_mu);
    err = std::move(m);
  }
  HBuf h_stage;     // host->device staging for host-pointer entry points
  int profiling = 0;
};

#define HIPCHK(ctx, call)                                                                     \
  do {                                                                                        \
    hipError_t e_ = (call);                                                                   \
    if (e_ != hipSuccess) {                                                                   \
      (ctx)->set_err(std::string(#call) + ": " + hipGetErrorString(e_));                      \
      return e_ == hipErrorOutOfMemory ? FVDB_E_OOM : FVDB_E_HIP;                             \
    }                                                                                         \
  } while (0)

#define FAIL(ctx, code, msg) \
  do {                       \
    (ctx)->set_err(msg);     \
    return (code);           \
  } while (0)

static inline uint32_t cdiv(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

struct fvdb_store {
  fvdb_ctx* ctx = nullptr;
  uint32_t d = 0, dpad = 0;
  uint64_t rows = 0, cap = 0;
  float* data = nullptr;  // [cap][dpad]
  DBuf s_q, s_cand, s_out, s_in;
};

struct fvdb_scorer {
  fvdb_store* store = nullptr;
  hipStream_t stream = nullptr;  // private stream: scorers of different host threads run concurrently
  uint32_t max_B = 0, max_C = 0;
  float* d_q = nullptr;        // [max_B][dpad]
  uint32_t* h_cand = nullptr;  // pinned, mapped
  float* h_dist = nullptr;     // pinned, mapped
  uint32_t* d_cand = nullptr;  // device aliases of the mapped buffers
  float* d_dist = nullptr;
  DBuf s_rows, s_in;  // private scratch: scorers are driven from different host threads
};
