// libpsi_hip.so: error reporting, device facts, growable scratch.
#include "psi_internal.h"
#include <stdarg.h>
#include <mutex>

static thread_local char g_err[512] = "ok";

void psi_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *psi_last_error(void) { return g_err; }
extern "C" int psi_version(void) { return 100; }

extern "C" int psi_device_info(int *cu_count, int *wave_size, int *clock_khz, int *is_gfx950)
{
    int dev = 0;
    PSI_CHECK_HIP(hipGetDevice(&dev));
    hipDeviceProp_t p;
    PSI_CHECK_HIP(hipGetDeviceProperties(&p, dev));
    if (cu_count) *cu_count = p.multiProcessorCount;
    if (wave_size) *wave_size = p.warpSize;
    if (clock_khz) *clock_khz = p.clockRate;
    if (is_gfx950) {
        const char *a = p.gcnArchName;
        *is_gfx950 = (a[0] == 'g' && a[1] == 'f' && a[2] == 'x' && a[3] == '9' && a[4] == '5' && a[5] == '0') ? 1 : 0;
    }
    return 0;
}

static std::mutex g_scratch_mu;
static void *g_scratch[16] = {0};
static size_t g_scratch_sz[16] = {0};

void *psi_scratch(size_t bytes)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (g_scratch_sz[dev] < bytes) {
        if (g_scratch[dev]) {
            (void)hipDeviceSynchronize();
            (void)hipFree(g_scratch[dev]);
            g_scratch[dev] = nullptr;
            g_scratch_sz[dev] = 0;
        }
        size_t want = bytes + bytes / 4 + (1 << 20);
        if (hipMalloc(&g_scratch[dev], want) != hipSuccess) {
            psi_set_error("scratch hipMalloc(%zu) failed", want);
            return nullptr;
        }
        g_scratch_sz[dev] = want;
    }
    return g_scratch[dev];
}

thread_local PsiStageTimer *g_psi_timer = nullptr;
