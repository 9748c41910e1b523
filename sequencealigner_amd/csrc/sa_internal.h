// Internal helpers shared by the HIP translation units of libsa_hip.so (not part of the C-ABI).
#ifndef SA_INTERNAL_H
#define SA_INTERNAL_H

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>

#include "sa_hip.h"

namespace sa {

// Per-thread message behind sa_last_error().
std::string& last_error();
int fail(int code, const char* fmt, ...);

#define SA_HIP_TRY(expr)                                                                         \
  do {                                                                                           \
    hipError_t _e = (expr);                                                                      \
    if (_e != hipSuccess)                                                                        \
      return sa::fail(SA_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                      __LINE__);                                                                 \
  } while (0)

// Direction codes, 2 bits per DP cell.  (The reference keeps an `edges` enum per cell,
// src/alignment.cpp:9-14; only these four states are needed to reproduce its traceback.)
enum : uint32_t { DIR_NONE = 0, DIR_DIAG = 1, DIR_LEFT = 2, DIR_UP = 3 };

inline size_t round_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Process-wide tuning and test options behind sa_set_option().  Each starts from its SA_* environment
// variable, read once when the first call needs it; later changes go through sa_set_option() only, so no
// call reads the environment while another thread may be changing it.
struct Options {
  std::atomic<long> force_i32{0};     // SA_FORCE_I32: never use the packed 16-bit kernels
  std::atomic<long> force_family{0};  // SA_FORCE_FAMILY: 0 automatic, 1 lane kernels, 2 wavefront kernels
  std::atomic<long> force_coop{-1};   // SA_FORCE_COOP: -1 automatic, 0 / 1 pin the cooperative wavefront kernel
  std::atomic<long> no_tables{0};     // SA_NO_TABLES: packed lane kernel without the score-table form
  std::atomic<long> coop_waves{8192}; // SA_COOP_WAVES: wavefronts of one pass of the cooperative Global /
                                      // Semiglobal kernel (64..8192)
  std::atomic<long> coop_c{0};        // SA_COOP_C: columns per lane of the cooperative kernel (0 automatic)
  std::atomic<long> coop_async{0};      // SA_COOP_ASYNC: 1: calls that run the cooperative kernels do not wait for them
  std::atomic<long> coop_seg_rows{-1};  // SA_COOP_SEG_ROWS: target rows per segment of the packed cooperative kernel's strips (-1 automatic, 0 none)
  std::atomic<long> coop_pk{-1};      // SA_COOP_PK: -1 automatic, 0 / 1 pin the packed form of the cooperative Global /
                                      // Semiglobal kernel (two pairs per wavefront) where it applies
  std::atomic<long> no_overlap{0};    // SA_NO_OVERLAP: never run a traceback on a second stream under the fill's last round
  std::atomic<long> pk16_k{0};        // SA_PK16_K: rows per strip of the packed lane kernel (32, 48, 64; 0 automatic)
  std::atomic<long> min_chunk{0};     // SA_MIN_CHUNK: windows per minimizer chunk (0 automatic, 8..4096)
  std::atomic<long> no_buckets{0};    // SA_NO_BUCKETS: single sequences take the device-wide radix sort
  std::atomic<long> no_segments{0};   // SA_NO_SEGMENTS: long pairs are traced back by one wavefront each
  std::atomic<long> seg_rows{0};      // SA_SEG_ROWS: target rows per traceback segment (0 automatic)
  std::atomic<long> spin_limit{1 << 24}; // SA_SPIN_LIMIT: polls a wait inside a cooperative kernel may take before
                                      // the kernels give up and the call fails (a small value: test hook)
  std::atomic<long> seg_spec{2};      // SA_SEG_SPEC: walks a traceback segment may make on speculative columns
  std::atomic<long> seg_cap{0};       // SA_SEG_CAP: runs a traceback segment may hold (0: its share of the slot);
                                      // a small value sends pairs to the one-wavefront walk (a test hook)
  std::atomic<long> fake_devices{0};  // SA_FAKE_DEVICES: the host batch calls deal to this many shards, all of
                                      // them on GPU 0 (exercises the multi-GPU path on one GPU)
};
Options& options();

}  // namespace sa

#endif
