// libpsi_hip.so: error reporting, device facts, growable scratch.
#include "psi_internal.h"
#include <stdarg.h>
#include <chrono>
#include <thread>
#include <map>
#include <mutex>
#include <utility>

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

// Internal scratch for entry points called with workspace = NULL: ONE buffer per (device, stream), so calls enqueued on different
// streams never share partial-result storage (a single per-device buffer raced between streams).  Growing a stream's buffer
// waits for THAT stream only before freeing the old one, and is refused while the stream is being captured into a hipGraph
// (allocation is illegal there): captured callers pass their own workspace.
static std::mutex g_scratch_mu;
struct PsiScratchEntry { void *ptr; size_t size; };
static std::map<std::pair<int, hipStream_t>, PsiScratchEntry> g_scratch;

void *psi_scratch(size_t bytes, hipStream_t stream)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    PsiScratchEntry &e = g_scratch[std::make_pair(dev, stream)];
    if (e.size < bytes) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (stream && hipStreamIsCapturing(stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) {
            psi_set_error("internal scratch cannot grow while the stream is being captured: pass a workspace");
            return nullptr;
        }
        if (e.ptr) {
            (void)hipStreamSynchronize(stream);
            (void)hipFree(e.ptr);
            e.ptr = nullptr;
            e.size = 0;
        }
        size_t want = bytes + bytes / 4 + (1 << 20);
        if (hipMalloc(&e.ptr, want) != hipSuccess) {
            psi_set_error("scratch hipMalloc(%zu) failed", want);
            e.ptr = nullptr;
            return nullptr;
        }
        e.size = want;
    }
    return e.ptr;
}

// Give back the internal scratch of one stream (NULL = the default stream) on the current device, or of EVERY stream of every device
// (all != 0).  Entries are otherwise kept for the life of the process: a caller that cycles through many short-lived streams (PyTorch's
// stream pool, one stream per engine in fitting_many) calls this before destroying a stream so that its buffer — and a stale map key a
// later stream handle could alias — does not stay behind.
extern "C" int psi_scratch_release(void *stream, int all)
{
    int dev = 0;
    PSI_CHECK_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    for (auto it = g_scratch.begin(); it != g_scratch.end();) {
        if (all || (it->first.first == dev && it->first.second == (hipStream_t)stream)) {
            if (it->second.ptr) {
                if (it->first.first == dev) (void)hipStreamSynchronize(it->first.second);
                else (void)hipDeviceSynchronize();
                (void)hipFree(it->second.ptr);
            }
            it = g_scratch.erase(it);
        } else {
            ++it;
        }
    }
    return 0;
}

thread_local PsiStageTimer *g_psi_timer = nullptr;

// Host-side wait for everything enqueued on `stream`, with a bound: polls hipStreamQuery (no blocking synchronise) and gives up after
// timeout_ms — a collective that never completes (a rank that died, ranks that disagree about how many collectives they issue) then
// becomes an error the caller can report instead of a process that hangs for ever.  0: done; PSI_ETIMEOUT: still running at the bound.
extern "C" int psi_stream_wait(void *stream, int timeout_ms)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (long polls = 0;; polls++) {
        hipError_t e = hipStreamQuery((hipStream_t)stream);
        if (e == hipSuccess) return 0;
        if (e != hipErrorNotReady) {
            psi_set_error("psi_stream_wait: %s", hipGetErrorString(e));
            return (int)e;
        }
        const long ms = (long)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
        if (timeout_ms >= 0 && ms >= timeout_ms) {
            psi_set_error("psi_stream_wait: the stream was still busy after %d ms (a collective that cannot complete?)", timeout_ms);
            return PSI_ETIMEOUT;
        }
        if (polls > 2000) std::this_thread::sleep_for(std::chrono::microseconds(200));    // the first ~ms spin, then back off
    }
}

