// dev: cost of a kernel boundary inside a replayed hipGraph on gfx950: chains of N dependent launches of (a) an empty kernel,
// (b) a kernel that writes then reads 4 MB through global memory (forces the inter-kernel release/acquire to matter).
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void empty_k(float *p) { if (p == nullptr && threadIdx.x == 12345) p[0] = 0; }
__global__ __launch_bounds__(256) void rw_k(const float *__restrict__ in, float *__restrict__ out, int n)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = in[i] + 1.0f;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main()
{
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const int n = 1 << 20;
    float *a, *b; CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMemset(a, 0, n * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 3; mode++) {
        for (int N : {1, 8, 32}) {
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
            for (int i = 0; i < N; i++) {
                if (mode == 0) hipLaunchKernelGGL(empty_k, dim3(1), dim3(64), 0, st, a);
                else if (mode == 1) hipLaunchKernelGGL(empty_k, dim3(1024), dim3(256), 0, st, a);
                else hipLaunchKernelGGL(rw_k, dim3(n / 256), dim3(256), 0, st, (i & 1) ? b : a, (i & 1) ? a : b, n);
            }
            CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            for (int w = 0; w < 20; w++) CK(hipGraphLaunch(ge, st));
            CK(hipStreamSynchronize(st));
            const int reps = 200;
            CK(hipEventRecord(e0, st));
            for (int r = 0; r < reps; r++) CK(hipGraphLaunch(ge, st));
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%-26s N=%2d: %.2f us per graph, %.2f us per kernel\n",
                   mode == 0 ? "empty 1x64" : (mode == 1 ? "empty 1024x256" : "4MB read+write 4096x256"), N, ms * 1e3 / reps, ms * 1e3 / reps / N);
            (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
        }
    }
    return 0;
}
