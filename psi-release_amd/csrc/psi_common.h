// Shared host-side helpers for libpsi_hip.so (gfx950 only; no other backend exists in this build).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/psi_hip.h"

#define PSI_WAVE 64

void psi_set_error(const char *fmt, ...);

#define PSI_CHECK_HIP(expr)                                                              \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess) {                                                          \
            psi_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return (int)_e;                                                              \
        }                                                                                \
    } while (0)

#define PSI_CHECK_LAUNCH(name)                                                           \
    do {                                                                                 \
        hipError_t _e = hipGetLastError();                                               \
        if (_e != hipSuccess) {                                                          \
            psi_set_error("launch of %s failed: %s", name, hipGetErrorString(_e));       \
            return (int)_e;                                                              \
        }                                                                                \
    } while (0)

#define PSI_REQUIRE(cond, msg)                                                           \
    do {                                                                                 \
        if (!(cond)) {                                                                   \
            psi_set_error("invalid argument: %s (%s)", msg, #cond);                      \
            return PSI_EINVAL;                                                           \
        }                                                                                \
    } while (0)

static inline int psi_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Per-device scratch that grows on demand (used when the caller passes workspace == NULL).
void *psi_scratch(size_t bytes);
