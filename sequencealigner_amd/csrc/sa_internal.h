// Internal helpers shared by the HIP translation units of libsa_hip.so (not part of the C-ABI).
#ifndef SA_INTERNAL_H
#define SA_INTERNAL_H

#include <hip/hip_runtime.h>

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

}  // namespace sa

#endif
