// Shared host-side helpers for libpsi_hip.so (gfx950 only; no other backend exists in this build).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/psi_hip.h"

#define PSI_WAVE 64

// The Chamfer squared distance d = x2*x2 + y2*y2 + z2*z2 (chamfer.cu:32-35) in the library's two arithmetic modes:
//   default            ((x2*x2 + y2*y2) + z2*z2), three products and two sums, each rounded (the sources that use this macro
//                      are compiled with contraction off) — what the CUDA source says, i.e. nvcc --fmad=false;
//   -DPSI_CHAMFER_FMA  fma(z2, z2, fma(y2, y2, x2*x2)) — what nvcc's DEFAULT --fmad=true makes of the same expression (it
//                      contracts each `product + sum` pair left to right: mul, fma, fma).  Built as libpsi_hip_fma.so and selected
//                      with PSI_CHAMFER_FMA=1; the oracle has the matching mode (oracle/chamfer_oracle.c, same macro).
#ifdef PSI_CHAMFER_FMA
#define PSI_SQ3(x, y, z) __builtin_fmaf((z), (z), __builtin_fmaf((y), (y), (x) * (x)))
#else
#define PSI_SQ3(x, y, z) ((x) * (x) + (y) * (y) + (z) * (z))
#endif

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
void *psi_scratch(size_t bytes, hipStream_t stream);
