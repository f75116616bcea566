// Adam over ALL parameters of a model in one or two launches (train_s1.py:229 / train_s2.py:295-296: optim.Adam(model.parameters(), lr),
// stepped once per batch) — the optimiser step of the CVAE trainers.
//
// The update is 7 streams over the parameters (read p, g, m, v; write p, m, v: 28 bytes per parameter, 440 MB for HumanCVAES2's 15.7 M) and
// nothing else: an HBM pass.  The parameters are 122 separate tensors from 64 floats to 1 M; the tensors' addresses travel in the KERNEL
// ARGUMENTS (up to 80 per launch: captured by value in a hipGraph, no table in device memory to keep in step with the allocator), a
// workgroup takes 8192 consecutive elements of one tensor with 16-byte accesses.
//
// Arithmetic: the operation order of PyTorch's fused Adam (the optimiser this replaces; ATen fused_adam_utils.cuh adam_math — the scalar
// hyper-parameters are doubles there, so the moment updates are evaluated in double and rounded to fp32 once):
//     g'   = g + wd * p                               (weight_decay != 0)
//     m    = fl(beta1 * m + (1 - beta1) * g')          v = fl(beta2 * v + (1 - beta2) * g' * g')
//     p   -= fl(lr / bc1) * m / (fl(sqrt(v) / sqrt(bc2)) + eps),      bc1 = 1 - beta1^t, bc2 = 1 - beta2^t, t = step + 1
// The step counter is ONE fp32 device scalar shared by all parameters (they are always stepped together): every workgroup reads it at
// its start, and the workgroup that FINISHES LAST in the last launch (a ticket counter) adds 1 — by then every other workgroup has read it.
#include "psi_internal.h"

namespace {

constexpr int ADAM_MAXT = 80;
constexpr unsigned ADAM_CHUNK = 8192;
typedef float f4 __attribute__((ext_vector_type(4)));

struct AdamTensor {
    float *p;
    const float *g;
    float *m, *v;
    unsigned n, blk0;                                              // elements; first workgroup of this tensor in the launch
};
struct AdamArgs {
    AdamTensor t[ADAM_MAXT];
    int nt;
    unsigned nblk;
};

__device__ __forceinline__ void adam_one(float &p, float g, float &m, float &v, double beta1, double beta2, double wd, double eps, float step_size,
                                         float bc2s)
{
    if (wd != 0.0) g = (float)((double)g + (double)p * wd);
    m = (float)(beta1 * (double)m + (1.0 - beta1) * (double)g);
    v = (float)(beta2 * (double)v + (1.0 - beta2) * (double)g * (double)g);
    const float denom = (float)((double)(sqrtf(v) / bc2s) + eps);
    p -= step_size * m / denom;
}

__global__ __launch_bounds__(256) void adam_multi_kernel(const AdamArgs a, float *__restrict__ step, unsigned *__restrict__ ticket, int last,
                                                         double lr, double beta1, double beta2, double eps, double wd)
{
    const unsigned b = blockIdx.x;
    int lo = 0, hi = a.nt - 1;                                    // the tensor this workgroup belongs to: last i with blk0[i] <= b
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (a.t[mid].blk0 <= b) lo = mid;
        else hi = mid - 1;
    }
    const AdamTensor T = a.t[lo];
    const double t = (double)*step + 1.0;
    const float bc1 = (float)(1.0 - pow(beta1, t)), bc2s = (float)sqrt(1.0 - pow(beta2, t));
    const float step_size = (float)(lr / (double)bc1);
    const unsigned e0 = (b - T.blk0) * ADAM_CHUNK, e1 = min(T.n, e0 + ADAM_CHUNK);
    const bool vec = ((((size_t)T.p | (size_t)T.g | (size_t)T.m | (size_t)T.v) & 15) == 0);
    if (vec) {
        const unsigned q1 = e0 + ((e1 - e0) & ~3u);
        for (unsigned i = e0 + 4 * threadIdx.x; i < q1; i += 1024) {
            f4 p = *(const f4 *)(T.p + i), m = *(const f4 *)(T.m + i), v = *(const f4 *)(T.v + i);
            const f4 g = *(const f4 *)(T.g + i);
#pragma unroll
            for (int e = 0; e < 4; e++) {
                float pe = p[e], me = m[e], ve = v[e];
                adam_one(pe, g[e], me, ve, beta1, beta2, wd, eps, step_size, bc2s);
                p[e] = pe; m[e] = me; v[e] = ve;
            }
            *(f4 *)(T.p + i) = p;
            *(f4 *)(T.m + i) = m;
            *(f4 *)(T.v + i) = v;
        }
        for (unsigned i = q1 + threadIdx.x; i < e1; i += 256) adam_one(T.p[i], T.g[i], T.m[i], T.v[i], beta1, beta2, wd, eps, step_size, bc2s);
    } else {
        for (unsigned i = e0 + threadIdx.x; i < e1; i += 256) adam_one(T.p[i], T.g[i], T.m[i], T.v[i], beta1, beta2, wd, eps, step_size, bc2s);
    }
    if (last) {
        __syncthreads();                                           // every thread of this workgroup has read `step` (above) and is done
        if (threadIdx.x == 0) {
            __threadfence();
            if (atomicAdd(ticket, 1u) == a.nblk - 1) {             // all other workgroups of this launch have passed this point: they read `step` long ago
                *ticket = 0;
                *step = (float)t;
            }
        }
    }
}

}  // namespace

// One Adam step over `count` fp32 tensors: p / g / m / v [count] arrays of device pointers (host arrays), n [count] element counts;
// step: device fp32 scalar (the number of steps taken so far; incremented); ticket: device uint32, zero-initialised once.
extern "C" int psi_adam_step(void *const *p, const void *const *g, void *const *m, void *const *v, const long *n, int count, float *step,
                             unsigned *ticket, double lr, double beta1, double beta2, double eps, double weight_decay, void *stream)
{
    PSI_REQUIRE(p && g && m && v && n && step && ticket && count > 0, "bad arguments");
    hipStream_t st = (hipStream_t)stream;
    int i = 0;
    while (i < count) {
        AdamArgs a;
        a.nt = 0;
        unsigned blk = 0;
        for (; i < count && a.nt < ADAM_MAXT; i++) {
            if (n[i] <= 0) continue;
            PSI_REQUIRE(n[i] < (1l << 31), "tensor too large");
            PSI_REQUIRE(p[i] && g[i] && m[i] && v[i], "null tensor");
            a.t[a.nt] = AdamTensor{(float *)p[i], (const float *)g[i], (float *)m[i], (float *)v[i], (unsigned)n[i], blk};
            blk += (unsigned)((n[i] + ADAM_CHUNK - 1) / ADAM_CHUNK);
            a.nt++;
        }
        if (a.nt == 0) break;
        a.nblk = blk;
        // (trailing empty tensors do not make a launch of their own: `last` is decided on what is left)
        int rest = 0;
        for (int j = i; j < count; j++) rest += n[j] > 0;
        hipLaunchKernelGGL(adam_multi_kernel, dim3(blk), dim3(256), 0, st, a, step, ticket, rest == 0 ? 1 : 0, lr, beta1, beta2, eps, weight_decay);
        PSI_CHECK_LAUNCH("adam_multi_kernel");
    }
    return 0;
}
