// VALU issue-rate calibration for gfx950 (VERDICT r05 "missing 2"): how many wave64 VALU instructions per second the chip retires,
// per instruction class, at 1 / 2 / 4 / 8 resident wavefronts per SIMD.  bench.py's VALU peak is taken from this table
// (profiles/r06/valu_peak.md), not from a guide sentence:  MI355X_MICROARCH.md:52-54 says a wave64 VALU instruction issues over 2 cycles
// (SIMD-32); rounds 4-5 assumed 4 (SIMD-16 as on gfx90a/gfx942).
//
//   hipcc --offload-arch=gfx950 -O3 tools/valu_peak.hip -o tools/scratch/valu_peak && tools/scratch/valu_peak
//
// Each lane keeps 8 independent registers (register pairs for the packed forms); one loop trip issues 64 instructions of the class
// round-robin over them, so a wavefront's own dependency distance is 8 instructions and the other resident wavefronts fill the rest.
// Inline asm volatile: the compiler neither folds nor reorders the instruction stream.  Loop overhead is scalar (s_add / s_cmp / s_cbranch).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float float2_t __attribute__((ext_vector_type(2)));

#define REP8(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7)
#define REP64(M) REP8(M) REP8(M) REP8(M) REP8(M) REP8(M) REP8(M) REP8(M) REP8(M)

// ---- one kernel per instruction class; all share the frame: 8 accumulators a[i], two loop-invariant operands k, c -----------------
#define KERNEL_SCALAR(NAME, ASM)                                                                                          \
    __global__ __launch_bounds__(256) void NAME(float* out, int trips, float k, float c, long long* clk) {                 \
        float a[8];                                                                                                        \
        for (int i = 0; i < 8; ++i) a[i] = 1.0f + 0.001f * float(threadIdx.x + i);                                        \
        long long t0 = clock64();                                                                                          \
        for (int t = 0; t < trips; ++t) {                                                                                  \
            REP64(ASM)                                                                                                     \
        }                                                                                                                  \
        long long t1 = clock64();                                                                                          \
        float s = 0.f;                                                                                                     \
        for (int i = 0; i < 8; ++i) s += a[i];                                                                             \
        if (s == 123.456f) out[0] = s;                                                                                     \
        if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;                                                         \
    }

#define A_MUL(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(k));
#define A_ADD(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
#define A_FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(k), "v"(c));
#define A_MULADD(i) asm volatile("v_mul_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2" : "+v"(a[i]) : "v"(k), "v"(c));   // 2 instructions per slot
#define A_RCP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
#define A_SQRT(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
#define A_MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(k));
#define A_ADDU(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(k));
#define A_MIN3(i) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(k), "v"(c));
#define A_DPP(i) asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]));
#define A_CMPCND(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(a[i]) : "v"(k), "v"(c) : "vcc");  // 2 per slot
#define A_CVT(i) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a[i]));
#define A_F64FMA(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(kd), "v"(cd));
#define A_PKMUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(kp));
#define A_PKADD(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(cp));
#define A_PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(kp), "v"(cp));

KERNEL_SCALAR(k_mul, A_MUL)
KERNEL_SCALAR(k_add, A_ADD)
KERNEL_SCALAR(k_fma, A_FMA)
KERNEL_SCALAR(k_muladd, A_MULADD)
KERNEL_SCALAR(k_rcp, A_RCP)
KERNEL_SCALAR(k_sqrt, A_SQRT)
KERNEL_SCALAR(k_mullo, A_MULLO)
KERNEL_SCALAR(k_addu, A_ADDU)
KERNEL_SCALAR(k_min3, A_MIN3)
KERNEL_SCALAR(k_dpp, A_DPP)
KERNEL_SCALAR(k_cmpcnd, A_CMPCND)
KERNEL_SCALAR(k_cvt, A_CVT)

#define KERNEL_PACKED(NAME, ASM)                                                                                          \
    __global__ __launch_bounds__(256) void NAME(float* out, int trips, float k, float c, long long* clk) {                 \
        float2_t p[8];                                                                                                     \
        float2_t kp = {k, k}, cp = {c, c};                                                                                 \
        for (int i = 0; i < 8; ++i) { p[i].x = 1.0f + 0.001f * float(threadIdx.x + i); p[i].y = p[i].x + 0.5f; }          \
        long long t0 = clock64();                                                                                          \
        for (int t = 0; t < trips; ++t) {                                                                                  \
            REP64(ASM)                                                                                                     \
        }                                                                                                                  \
        long long t1 = clock64();                                                                                          \
        float s = 0.f;                                                                                                     \
        for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y;                                                                  \
        if (s == 123.456f) out[0] = s;                                                                                     \
        if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;                                                         \
    }
KERNEL_PACKED(k_pkmul, A_PKMUL)
KERNEL_PACKED(k_pkadd, A_PKADD)
KERNEL_PACKED(k_pkfma, A_PKFMA)

__global__ __launch_bounds__(256) void k_f64fma(float* out, int trips, float k, float c, long long* clk) {
    double d[8];
    double kd = k, cd = c;
    for (int i = 0; i < 8; ++i) d[i] = 1.0 + 0.001 * double(threadIdx.x + i);
    long long t0 = clock64();
    for (int t = 0; t < trips; ++t) {
        REP64(A_F64FMA)
    }
    long long t1 = clock64();
    double s = 0.;
    for (int i = 0; i < 8; ++i) s += d[i];
    if (s == 123.456) out[0] = float(s);
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

typedef void (*kern_t)(float*, int, float, float, long long*);
struct Row { const char* name; kern_t fn; int insts_per_slot; const char* note; };

int main(int argc, char** argv) {
    int dev = 0;
    CK(hipSetDevice(dev));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, dev));
    const int cus = prop.multiProcessorCount;
    int clock_khz = 0;
    CK(hipDeviceGetAttribute(&clock_khz, hipDeviceAttributeClockRate, dev));
    printf("# VALU issue calibration — %s, %d CUs, reported max clock %.0f MHz\n\n", prop.gcnArchName, cus, clock_khz / 1000.0);
    float* out; long long* clk;
    CK(hipMalloc(&out, 64));
    CK(hipMalloc(&clk, 64));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const Row rows[] = {
        {"v_mul_f32", k_mul, 1, ""}, {"v_add_f32", k_add, 1, ""}, {"v_fma_f32", k_fma, 1, ""},
        {"v_mul_f32 ; v_add_f32 (dependent pair)", k_muladd, 2, "the uncontracted product-sum of the correspondence pass"},
        {"v_pk_mul_f32", k_pkmul, 1, "2 f32 per lane"}, {"v_pk_add_f32", k_pkadd, 1, "2 f32 per lane"}, {"v_pk_fma_f32", k_pkfma, 1, "2 f32 per lane"},
        {"v_fma_f64", k_f64fma, 1, ""},
        {"v_add_u32", k_addu, 1, ""}, {"v_min3_u32", k_min3, 1, ""}, {"v_mul_lo_u32", k_mullo, 1, ""}, {"v_cvt_i32_f32", k_cvt, 1, ""},
        {"v_cmp_lt_f32 ; v_cndmask_b32", k_cmpcnd, 2, "VOPC writes vcc"},
        {"v_add_f32 row_shr:1 (DPP)", k_dpp, 1, ""},
        {"v_rcp_f32", k_rcp, 1, "transcendental"}, {"v_sqrt_f32", k_sqrt, 1, "transcendental"},
    };
    const int waves_per_simd[] = {1, 2, 4, 8};
    printf("wave-instructions per second over the whole chip (1e12/s); in brackets: shader-clock cycles per wave-instruction per SIMD "
           "(from clock64 of one wavefront, s_memtime) — `cycles x resident waves` would be flat if issue were the only limit\n\n");
    printf("| instruction | 1 wave/SIMD | 2 waves/SIMD | 4 waves/SIMD | 8 waves/SIMD | note |\n|---|---|---|---|---|---|\n");
    double best_mul = 0, best_pk = 0;
    for (const Row& r : rows) {
        printf("| `%s` |", r.name);
        for (int w : waves_per_simd) {
            // w waves per SIMD = w workgroups of 256 lanes per CU
            const int blocks = cus * w;
            int trips = 4000;
            r.fn<<<blocks, 256>>>(out, 50, 1.0000001f, 1e-9f, clk);      // warm-up (clock ramp)
            CK(hipDeviceSynchronize());
            double best = 0, cyc = 0;
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipEventRecord(e0));
                r.fn<<<blocks, 256>>>(out, trips, 1.0000001f, 1e-9f, clk);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                long long c; CK(hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost));
                const double insts = double(blocks) * 4.0 * double(trips) * 64.0 * r.insts_per_slot;     // wave-instructions
                const double rate = insts / (ms * 1e-3);
                if (rate > best) { best = rate; cyc = double(c) / (double(trips) * 64.0 * r.insts_per_slot) / w; }
            }
            printf(" %.3f (%.2f) |", best / 1e12, cyc);
            if (r.fn == k_mul && best > best_mul) best_mul = best;
            if (r.fn == k_pkmul && best > best_pk) best_pk = best;
        }
        printf(" %s |\n", r.note);
        fflush(stdout);
    }
    const double simds = double(cus) * 4.0;
    printf("\n**Result.**  Non-packed f32: %.3fe12 wave-instructions/s = %.2f GHz x %d SIMDs / %.2f cycles per wave64 instruction "
           "(at the reported %.2f GHz).  Packed f32: %.3fe12 wave-instructions/s (= %.3fe12 lane-pairs... two f32 per lane each).\n",
           best_mul / 1e12, clock_khz / 1e6, int(simds), simds * clock_khz * 1e3 / best_mul, clock_khz / 1e6, best_pk / 1e12, 2 * best_pk / 1e12);
    printf("VALU_PEAK_WAVE_INSTR_PER_S=%.4e\n", best_mul);
    return 0;
}
