// queue_fairness -- how does the GPU arbitrate between two streams of one process?  Creates K non-blocking streams in order and,
// for every pair (i, j), launches the same busy kernel on both at once; prints each kernel's duration.  Fair (round-robin)
// pairs finish together in ~2x the lone time; under strict priority one finishes in ~1x and the other in ~2x.
//   hipcc --offload-arch=gfx950 -O2 tools/queue_fairness.hip -o /tmp/queue_fairness && GPU_MAX_HW_QUEUES=8 /tmp/queue_fairness 10
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ __launch_bounds__(256) void busy(float *out, int iters)
{
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-4f;
    for (int i = 0; i < iters; ++i) { a = a * 1.0001f + b; b = b * 0.9999f + a; }
    if (a == 12345.678f) out[0] = a + b;
}
int main(int argc, char **argv)
{
    const int K = argc > 1 ? std::atoi(argv[1]) : 10;
    const int wgs = argc > 2 ? std::atoi(argv[2]) : 4096, iters = argc > 3 ? std::atoi(argv[3]) : 2000, chain = argc > 4 ? std::atoi(argv[4]) : 1;
    float *out; CK(hipMalloc(&out, 4));
    std::vector<hipStream_t> st(K);
    for (int i = 0; i < K; ++i) { CK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking)); hipLaunchKernelGGL(busy, dim3(1), dim3(256), 0, st[i], out, 1); }
    CK(hipDeviceSynchronize());
    hipEvent_t e[4];
    for (auto &x : e) CK(hipEventCreate(&x));
    float lone = 0;
    CK(hipEventRecord(e[0], st[0]));
    for (int c = 0; c < chain; ++c) hipLaunchKernelGGL(busy, dim3(wgs), dim3(256), 0, st[0], out, iters);
    CK(hipEventRecord(e[1], st[0]));
    CK(hipDeviceSynchronize()); CK(hipEventElapsedTime(&lone, e[0], e[1]));
    std::printf("lone chain of %d launches: %.1f us (%d workgroups each); cells = %% of that for stream i / stream j\n      ", chain, lone * 1e3f, wgs);
    for (int j = 0; j < K; ++j) std::printf("   j=%-2d     ", j);
    std::printf("\n");
    for (int i = 0; i < K; ++i) {
        std::printf("i=%-2d  ", i);
        for (int j = 0; j < K; ++j) {
            if (i == j) { std::printf("     -      "); continue; }
            float a = 0, b = 0;
            CK(hipEventRecord(e[0], st[i])); CK(hipEventRecord(e[2], st[j]));
            for (int c = 0; c < chain; ++c) {                      // chains of dependent launches, like the passes of two pose groups
                hipLaunchKernelGGL(busy, dim3(wgs), dim3(256), 0, st[i], out, iters);
                hipLaunchKernelGGL(busy, dim3(wgs), dim3(256), 0, st[j], out, iters);
            }
            CK(hipEventRecord(e[1], st[i])); CK(hipEventRecord(e[3], st[j]));
            CK(hipDeviceSynchronize());
            CK(hipEventElapsedTime(&a, e[0], e[1])); CK(hipEventElapsedTime(&b, e[2], e[3]));
            std::printf(" %4.0f/%-4.0f  ", 100.0f * a / lone, 100.0f * b / lone);
        }
        std::printf("\n");
    }
    return 0;
}
