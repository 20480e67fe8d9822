// common.h -- shared host-side helpers for liblic360_hip (error reporting, launch geometry).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdarg>
#include "../../include/lic360_hip.h"

#define LIC360_API extern "C" __attribute__((visibility("default")))

void lic360_set_error(const char *fmt, ...);

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            lic360_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return 1;                                                                        \
        }                                                                                    \
    } while (0)

#define LAUNCH_CHECK()                                                                       \
    do {                                                                                     \
        hipError_t _e = hipGetLastError();                                                   \
        if (_e != hipSuccess) {                                                              \
            lic360_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
            return 1;                                                                        \
        }                                                                                    \
    } while (0)

#define ARG_CHECK(cond)                                                                      \
    do {                                                                                     \
        if (!(cond)) {                                                                       \
            lic360_set_error("bad argument: %s (%s:%d)", #cond, __FILE__, __LINE__);         \
            return 2;                                                                        \
        }                                                                                    \
    } while (0)

// 1-D streaming launches: 256 threads/block; enough blocks to cover `n` items at `per_thread`
// each, capped so that grid-stride loops keep >= 8 waves per CU resident (256 CUs).
static inline unsigned lic360_blocks(long n, int per_thread = 1) {
    long b = (n + 256L * per_thread - 1) / (256L * per_thread);
    if (b < 1) b = 1;
    if (b > 256L * 32) b = 256L * 32;
    return (unsigned)b;
}
