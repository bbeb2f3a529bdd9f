// common.h -- error plumbing shared by the translation units of libabnet3_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/abnet3_hip.h"

namespace abn {

void set_error(const char* fmt, ...);

#define ABN_REQUIRE(cond, ...)                      \
    do {                                            \
        if (!(cond)) {                              \
            ::abn::set_error(__VA_ARGS__);          \
            return ABN_E_ARG;                       \
        }                                           \
    } while (0)

// Launch errors only: never synchronises (callers may be capturing a graph).
#define ABN_CHECK_LAUNCH(what)                                               \
    do {                                                                     \
        hipError_t e_ = hipGetLastError();                                   \
        if (e_ != hipSuccess) {                                              \
            ::abn::set_error("%s: %s", what, hipGetErrorString(e_));         \
            return ABN_E_LAUNCH;                                             \
        }                                                                    \
    } while (0)

static inline int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }
static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace abn
