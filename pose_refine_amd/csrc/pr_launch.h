// pr_launch.h -- host-side helpers of the launchers (one copy per translation unit)
// gfx950 (CDNA4, wave64); compiled with -ffp-contract=off: every per-element value is bit-identical to the CPU restatement.
#pragma once
#include "pr_device.h"

namespace prk {

// Dynamic LDS beyond 64 KiB is an opt-in PER DEVICE (hipFuncSetAttribute applies to the current device): one flag per kernel slot and
// device, so that a process driving several GPUs (one host thread each) opts in on every one of them.
static bool lds_opt_in(const void *fn, int slot, uint32_t bytes)
{
    static std::atomic<uint64_t> done[4];                         // bit = device ordinal (<= 64 devices), one word per kernel slot
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    if (done[slot].load(std::memory_order_acquire) & (1ull << dev)) return true;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return false;
    done[slot].fetch_or(1ull << dev, std::memory_order_release);
    return true;
}
static inline uint32_t cap_grid(size_t want) { return (uint32_t)(want < 1 ? 1 : (want > 8192 ? 8192 : want)); }

// row scans of the depth -> cloud stage (d2c.hip), called from the render and scene launchers of other files
hipError_t launch_d2c_scan(const uint32_t *row_count, uint32_t gh, uint32_t *row_off, uint32_t *counts, uint32_t n_img, hipStream_t s);
hipError_t launch_d2c_scan_init(const uint32_t *row_count, uint32_t gh, uint32_t *row_off, uint32_t *counts, uint32_t n_img,
                                PoseMeta *meta, DevIcpState *st, uint32_t *arrive, uint32_t cloud_stride, hipStream_t s);

}  // namespace prk
