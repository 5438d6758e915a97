// queue_fairness3 -- three streams at once, as in a pipelined step: one long launch (the other slot's raster) on stream r and two
// chains of short dependent launches (the two pose groups' passes) on streams a and b.  Prints, for every (r, a, b) out of K
// streams, how long each chain takes relative to a lone chain.  GPU_MAX_HW_QUEUES=8 ./queue_fairness3 8
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
    const int K = argc > 1 ? std::atoi(argv[1]) : 8;
    float *out; CK(hipMalloc(&out, 4));
    std::vector<hipStream_t> st(K);
    for (int i = 0; i < K; ++i) { CK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking)); hipLaunchKernelGGL(busy, dim3(1), dim3(256), 0, st[i], out, 1); }
    CK(hipDeviceSynchronize());
    hipEvent_t e[6];
    for (auto &x : e) CK(hipEventCreate(&x));
    const int chain = 21, wgs = 1664, iters = 700, rwgs = 31488, riters = 500;
    float lone = 0, lone_r = 0;
    CK(hipEventRecord(e[0], st[0]));
    for (int c = 0; c < chain; ++c) hipLaunchKernelGGL(busy, dim3(wgs), dim3(256), 0, st[0], out, iters);
    CK(hipEventRecord(e[1], st[0])); CK(hipDeviceSynchronize()); CK(hipEventElapsedTime(&lone, e[0], e[1]));
    CK(hipEventRecord(e[0], st[0])); hipLaunchKernelGGL(busy, dim3(rwgs), dim3(256), 0, st[0], out, riters); CK(hipEventRecord(e[1], st[0]));
    CK(hipDeviceSynchronize()); CK(hipEventElapsedTime(&lone_r, e[0], e[1]));
    std::printf("lone chain %.0f us, lone long launch %.0f us; rows: r a b -> chain a %%, chain b %%, long launch %%\n", lone * 1e3f, lone_r * 1e3f);
    for (int r = 0; r < K; ++r) for (int a = 0; a < K; ++a) for (int b = a + 1; b < K; ++b) {
        if (a == r || b == r) continue;
        float ta, tb, tr;
        CK(hipEventRecord(e[0], st[a])); CK(hipEventRecord(e[2], st[b])); CK(hipEventRecord(e[4], st[r]));
        hipLaunchKernelGGL(busy, dim3(rwgs), dim3(256), 0, st[r], out, riters);
        for (int c = 0; c < chain; ++c) {
            hipLaunchKernelGGL(busy, dim3(wgs), dim3(256), 0, st[a], out, iters);
            hipLaunchKernelGGL(busy, dim3(wgs), dim3(256), 0, st[b], out, iters);
        }
        CK(hipEventRecord(e[1], st[a])); CK(hipEventRecord(e[3], st[b])); CK(hipEventRecord(e[5], st[r]));
        CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&ta, e[0], e[1])); CK(hipEventElapsedTime(&tb, e[2], e[3])); CK(hipEventElapsedTime(&tr, e[4], e[5]));
        std::printf("%d %d %d -> %4.0f %4.0f %4.0f%s\n", r, a, b, 100 * ta / lone, 100 * tb / lone, 100 * tr / lone_r,
                    (ta > 1.25f * tb || tb > 1.25f * ta) ? "   <-- unfair" : "");
    }
    return 0;
}
