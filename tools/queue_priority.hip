// queue_priority -- does a low-priority long launch (a raster) fill the gaps two high-priority chains of short dependent launches
// (the passes of two pose groups) leave, without slowing them?   GPU_MAX_HW_QUEUES=8 ./queue_priority
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ __launch_bounds__(256) void busy(float *out, int iters)
{
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-4f;
    for (int i = 0; i < iters; ++i) { a = a * 1.0001f + b; b = b * 0.9999f + a; }
    if (a == 12345.678f) out[0] = a + b;
}
int main()
{
    float *out; CK(hipMalloc(&out, 4));
    int least = 0, greatest = 0;
    CK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    std::printf("priority range: least %d greatest %d\n", least, greatest);
    const int chain = 21, wgs = 1664, iters = 700, rwgs = 31488, riters = 500;
    hipEvent_t e[6];
    for (auto &x : e) CK(hipEventCreate(&x));
    const char *names[4] = { "all default", "chains HIGH, long LOW", "chains LOW, long HIGH", "chains HIGH, long default" };
    for (int cfg = 0; cfg < 4; ++cfg) {
        hipStream_t a, b, r;
        const int pc = (cfg == 1 || cfg == 3) ? greatest : (cfg == 2 ? least : 0), pr = cfg == 1 ? least : (cfg == 2 ? greatest : 0);
        if (cfg == 0) { CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&r, hipStreamNonBlocking)); }
        else {
            CK(hipStreamCreateWithPriority(&a, hipStreamNonBlocking, pc)); CK(hipStreamCreateWithPriority(&b, hipStreamNonBlocking, pc));
            if (cfg == 3) CK(hipStreamCreateWithFlags(&r, hipStreamNonBlocking)); else CK(hipStreamCreateWithPriority(&r, hipStreamNonBlocking, pr));
        }
        hipLaunchKernelGGL(busy, dim3(1), dim3(256), 0, a, out, 1); hipLaunchKernelGGL(busy, dim3(1), dim3(256), 0, b, out, 1); hipLaunchKernelGGL(busy, dim3(1), dim3(256), 0, r, out, 1);
        CK(hipDeviceSynchronize());
        float lone = 0, lone_r = 0, pair = 0;
        CK(hipEventRecord(e[0], a)); for (int c = 0; c < chain; ++c) hipLaunchKernelGGL(busy, dim3(wgs), dim3(256), 0, a, out, iters); CK(hipEventRecord(e[1], a));
        CK(hipDeviceSynchronize()); CK(hipEventElapsedTime(&lone, e[0], e[1]));
        CK(hipEventRecord(e[0], r)); hipLaunchKernelGGL(busy, dim3(rwgs), dim3(256), 0, r, out, riters); CK(hipEventRecord(e[1], r));
        CK(hipDeviceSynchronize()); CK(hipEventElapsedTime(&lone_r, e[0], e[1]));
        CK(hipEventRecord(e[0], a)); CK(hipEventRecord(e[2], b));
        for (int c = 0; c < chain; ++c) { hipLaunchKernelGGL(busy, dim3(wgs), dim3(256), 0, a, out, iters); hipLaunchKernelGGL(busy, dim3(wgs), dim3(256), 0, b, out, iters); }
        CK(hipEventRecord(e[1], a)); CK(hipEventRecord(e[3], b));
        CK(hipDeviceSynchronize()); CK(hipEventElapsedTime(&pair, e[0], e[1]));
        for (int rep = 0; rep < 3; ++rep) {
            float ta, tb, tr;
            CK(hipEventRecord(e[0], a)); CK(hipEventRecord(e[2], b)); CK(hipEventRecord(e[4], r));
            hipLaunchKernelGGL(busy, dim3(rwgs), dim3(256), 0, r, out, riters);
            for (int c = 0; c < chain; ++c) { hipLaunchKernelGGL(busy, dim3(wgs), dim3(256), 0, a, out, iters); hipLaunchKernelGGL(busy, dim3(wgs), dim3(256), 0, b, out, iters); }
            CK(hipEventRecord(e[1], a)); CK(hipEventRecord(e[3], b)); CK(hipEventRecord(e[5], r));
            CK(hipDeviceSynchronize());
            CK(hipEventElapsedTime(&ta, e[0], e[1])); CK(hipEventElapsedTime(&tb, e[2], e[3])); CK(hipEventElapsedTime(&tr, e[4], e[5]));
            std::printf("%-26s lone chain %.0f us, two chains %.0f us, lone long %.0f us | together: chain a %.0f us, chain b %.0f us, long %.0f us\n",
                        names[cfg], lone * 1e3f, pair * 1e3f, lone_r * 1e3f, ta * 1e3f, tb * 1e3f, tr * 1e3f);
        }
    }
    return 0;
}
