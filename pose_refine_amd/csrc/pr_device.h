// pr_device.h -- device-side helpers shared by every kernel file: x86 float->int semantics, offset loads, DPP lane access
// gfx950 (CDNA4, wave64); compiled with -ffp-contract=off: every per-element value is bit-identical to the CPU restatement.
#pragma once
#include <hip/hip_runtime.h>
#include <limits.h>
#include <float.h>
#include <stdlib.h>
#include <atomic>

#include "pr_internal.h"
#include "pr_tuning.h"

namespace prk {

// ------------------------------------------------------------------------------------------------
// conversions with the semantics the reference CPU build has on x86-64 (cvttss2si): out-of-range
// and NaN inputs give INT_MIN.  v_cvt_i32_f32 saturates instead, so the range test is explicit.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int f2i_x86(float v)
{
    return (v > -2147483904.0f && v < 2147483648.0f) ? (int)v : INT_MIN;
}
// size_t(float) for loop starts that are later compared against a bound < 2^24: anything
// outside (-1, 2^24) can never satisfy "(float)p <= bound", so it maps to "no iterations".
__device__ __forceinline__ int loop_start(float v)
{
    return (v > -1.0f && v < 16777216.0f) ? (int)v : INT_MAX;
}

__device__ __forceinline__ float sel_max(float a, float b) { return (a > b) ? a : b; }
__device__ __forceinline__ float sel_min(float a, float b) { return (a < b) ? a : b; }

// One-instruction square root / reciprocal (v_sqrt_f32 / v_rcp_f32: 1 ulp) for SAFETY MARGINS only -- step lengths, window radii, slack: every
// use multiplies the result by a factor at least 1e-6 away from 1 on the safe side.  Never for a value the oracle is compared with bit by bit
// (sqrtf and '/' expand to the correctly rounded sequences there: ten instructions and more each).
__device__ __forceinline__ float margin_sqrt(float v) { return __builtin_amdgcn_sqrtf(v); }
__device__ __forceinline__ float margin_rcp(float v) { return __builtin_amdgcn_rcpf(v); }

typedef float float2v __attribute__((ext_vector_type(2)));
// base + 32-bit byte offset: lets the compiler address with a scalar base and one 32-bit VGPR (global_load ... v_off, s[base])
// instead of building a 64-bit address per lane (v_lshl_add_u64 / v_mad_u64_u32 plus the copies that come with 64-bit values)
template <class T> __device__ __forceinline__ T ld_off(const void *base, uint32_t byte_off)
{ return *reinterpret_cast<const T *>(static_cast<const char *>(base) + byte_off); }
template <class T> __device__ __forceinline__ void st_off(void *base, uint32_t byte_off, const T &v)
{ *reinterpret_cast<T *>(static_cast<char *>(base) + byte_off) = v; }
struct Corr { float dx, dy, dz, nx, ny, nz; };   // destination point and its normal

// ------------------------------------------------------------------------------------------------
// wave64 sum with a fixed balanced pairwise tree in lane order, result in lane 63:
//   row_shr:1,2,4,8 inside each 16-lane row, row_bcast15 into rows 1 and 3, row_bcast31 into rows 2,3.
// ------------------------------------------------------------------------------------------------
template <int kCtrl, int kRowMask>
__device__ __forceinline__ float dpp_get(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), kCtrl, kRowMask, 0xf, true));
}

// sampled fingerprint of up to three caller-owned arrays (scene_fingerprint_kernel, batch_check_kernel): this lane's share of the hash of a
// 256-lane workgroup -- 16 of the 4096 samples per array; the workgroup's fingerprint is the SUM of its lanes' values (order-free)
__device__ __forceinline__ uint32_t fingerprint_lane(const uint32_t *__restrict__ a, unsigned long long na, const uint32_t *__restrict__ b, unsigned long long nb,
                                                     const uint32_t *__restrict__ c, unsigned long long nc)
{
    uint32_t h = 0;
    const uint32_t *arr[3] = { a, b, c };
    const unsigned long long len[3] = { na, nb, nc };
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (!arr[k] || len[k] == 0) continue;
        // sample s of 4096 sits in stripe s of the array, at a hashed offset inside it (32-bit arithmetic: __umulhi maps a hash onto a range)
        const uint32_t n = len[k] > 0xffffffffull ? 0xffffffffu : (uint32_t)len[k];
        const uint32_t stripe = n / 4096u;                            // 0 for short arrays: every word is visited, wrapping around
        uint32_t w[16];
#pragma unroll
        for (uint32_t i = 0; i < 16u; ++i) {
            const uint32_t s = threadIdx.x * 16u + i;
            const uint32_t pos = stripe ? s * stripe + __umulhi(s * 2654435761u + 0x9e3779b9u, stripe) : (n >= 4096u ? s : s % n);
            w[i] = arr[k][pos] ^ pos;
        }
#pragma unroll
        for (uint32_t i = 0; i < 16u; ++i) h += w[i] * 2654435761u + (uint32_t)k;      // a sum: the order of the lanes does not matter
    }
    return h;
}

constexpr uint32_t kBoxRowsPerBlock = 16;                        // rows of a pixel box per workgroup: 4 wavefronts x 4 rows (few, fatter workgroups: dispatch-bound otherwise)

}  // namespace prk
