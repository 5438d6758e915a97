// pr_runtime.h -- the host-side runtime behind the C ABI, shared by its translation units: contexts and their registry, workspaces, the write
// log of caller-owned memory, options, slots, profiling spans.  pr_context.cpp (contexts, memory, options, profiling entry points),
// pr_scene.cpp (scene caches and device-side scene preparation), pr_icp.cpp (the batched ICP driver: device and host solve loops),
// pr_refine.cpp (render, cloud, the fused batch path with its two asynchronous slots and helper threads), pr_comm.cpp (RCCL gather).
#pragma once
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>          // types and prototypes only: librccl is opened on first use (pr_comm_*), never linked
#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "pr_internal.h"
#include "pr_tuning.h"

#include "pr_internal.h"

namespace prh {
extern thread_local std::string g_err;
void set_error(const char *fmt, ...);
}  // namespace prh

namespace prr {

using prh::set_error;

// PR_TRACE=<file>: host-side time stamps of the batch path (who waits for whom when a step stalls), dumped at process exit.  Off: one load and a branch.
void trace_mark(const char *tag);


#define HIP_TRY(expr)                                                                           \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            (void)hipGetLastError();          /* the runtime's sticky copy of this error must not be found by the next launch's check */ \
            return PR_ERR_HIP;                                                                  \
        }                                                                                       \
    } while (0)
#define PR_TRY(expr) do { int rc_ = (expr); if (rc_ != PR_OK) return rc_; } while (0)

struct DevBuf {                      // grow-only device workspace
    void *p = nullptr; size_t cap = 0;
    int ensure(size_t bytes)
    {
        if (bytes <= cap) return PR_OK;
        if (p) { hipFree(p); p = nullptr; cap = 0; }
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) { set_error("hipMalloc(%zu) failed: %s", want, hipGetErrorString(e)); p = nullptr; return PR_ERR_NOMEM; }
        cap = want;
        return PR_OK;
    }
    void release() { if (p) hipFree(p); p = nullptr; cap = 0; }
    template <class T> T *as() const { return static_cast<T *>(p); }
};
struct PinBuf {                      // grow-only pinned host staging
    void *p = nullptr; size_t cap = 0;
    int ensure(size_t bytes)
    {
        if (bytes <= cap) return PR_OK;
        if (p) { hipHostFree(p); p = nullptr; cap = 0; }
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
        if (e != hipSuccess) { set_error("hipHostMalloc(%zu) failed: %s", want, hipGetErrorString(e)); p = nullptr; return PR_ERR_NOMEM; }
        cap = want;
        return PR_OK;
    }
    void release() { if (p) hipHostFree(p); p = nullptr; cap = 0; }
    template <class T> T *as() const { return static_cast<T *>(p); }
};

// ---- which device ranges were written through this library, and when --------------------------------
// Derived data (the packed projective scene, the kd traversal records, the host copy of a model box) is cached by the address
// of the caller's buffers.  A cache entry remembers the write generation it was built at and is dropped as soon as a later
// write through the library (pr_memcpy_*, pr_fill_i32, pr_free, the *_prepare_dev / *_build_dev functions, pr_render) or a
// pr_invalidate() call overlaps one of its source ranges.  Writes the library cannot see (raw HIP calls, the caller's own
// kernels) must be announced with pr_invalidate -- include/pose_refine.h states that contract.
struct WriteLog {
    std::mutex mu;
    struct W { uintptr_t lo, hi; };
    static constexpr uint64_t kRing = 128;
    W ring[kRing];
    uint64_t gen = 0;                                            // writes recorded so far; write k (1-based) sits in ring[k % kRing]
    void note(const void *p, size_t bytes)
    {
        if (!p) return;
        uintptr_t lo = reinterpret_cast<uintptr_t>(p), hi = lo + bytes;
        if (bytes == 0) {                                        // unknown extent: the whole allocation that contains p
            void *base = nullptr; size_t size = 0;
            if (hipMemGetAddressRange(reinterpret_cast<hipDeviceptr_t *>(&base), &size, const_cast<void *>(p)) == hipSuccess && base) { lo = reinterpret_cast<uintptr_t>(base); hi = lo + size; }
            else { (void)hipGetLastError(); lo = 0; hi = ~(uintptr_t)0; }
        }
        std::lock_guard<std::mutex> lk(mu);
        ++gen;
        ring[gen % kRing] = W{ lo, hi };
    }
    uint64_t now() { std::lock_guard<std::mutex> lk(mu); return gen; }
    // may [p, p+bytes) have been written after generation `since`?
    bool written_since(uint64_t since, const void *p, size_t bytes)
    {
        const uintptr_t lo = reinterpret_cast<uintptr_t>(p), hi = lo + bytes;
        std::lock_guard<std::mutex> lk(mu);
        if (gen - since >= kRing) return true;                   // older writes have left the ring: assume the worst
        for (uint64_t k = since + 1; k <= gen; ++k) { const W &w = ring[k % kRing]; if (w.lo < hi && lo < w.hi) return true; }
        return false;
    }
};
extern WriteLog g_writes;
extern std::atomic<uint64_t> g_flag_overtook;     // PR_SOLVE_HOST: group flags that reached the host before all of their rows (process-wide; expected 0; read-only option "stat_flag_overtook")

// ---- process-wide options (pr_set_option): plain ints, shared by every context ---------------------
constexpr int kSlots = PR_SLOTS;            // asynchronous slots per context (pr_tuning.h)
struct Options {
    int pose_groups = 0;             // split the batch over this many streams (1..4); 0 = two, for either scene kind (pose_groups_for below has the measurements);
                                     // launches of different groups overlap, so timed calls fall back to one group
    int solve_mode = PR_SOLVE_HOST;
    int host_worker = 1;             // PR_SOLVE_HOST batches of pr_refine_submit run on the slot's helper thread (0: on the caller's thread, inside the call)
    int steps = 3;                   // 1024-point steps per workgroup -> 3072 points per workgroup (round 3: 3-5 % faster than 2048 / 4096).  Round 6, with the write-back issued behind the gathers, option points_per_block = 4096 pipelines 2 % better (286.5 / 287.3 k against 280.6 / 281.4 k, 100 steps, same box; 273.6 against 271.6 k at 20 steps; 5120: 282 k, 6144: 277 k, 8192: 242-262 k; kd-tree and host solve within noise) but a LONE 256-hypothesis launch takes 39.6 instead of 37.7 us (six longer chains per hypothesis instead of eight): the default stays, the option is there (it selects another summation tree: a quarter of the bench's hypotheses end in a different basin, the oracle follows)
    int profile = 0;
    int sample_period = 32;          // profile 2: one timed (synchronous, single-group) call in this many
    int nn_lds_nodes = 1024;
    int nn_lds_records = 0;          // stack traversal: leading 64-byte node records staged in LDS; measured 0/64/128/256/512 -> 42.2/41.6/45.2/45.4/58.2 ms per step (occupancy lost to the extra LDS outweighs the saved L1 lookups)
    int nn_seed = 1;                 // compact kd records: start every search from the previous pass' winner distance
    int nn_compact = 1;              // stack traversal: 32-byte node records with 16-bit outward-rounded child boxes (half the L1 traffic)
    int blocking_wait = 0;           // pr_refine_wait sleeps on the slot's event instead of spinning (set before the first asynchronous batch)
    int nn_wide = 1;                 // queued tree searches: order-free walk over 128-byte lines (eight subtree boxes per wide node, one line per leaf); ties go to the binary walk
    int nn_stack = 1;                // kd-tree query: per-lane LDS stack (1) or the reference's stackless walk (0)
    int nn_split = 1;                // kd-tree scenes on compact records: search kernel (runs of consecutive points, grid window) + winners pass
    int nn_run = 8;                  // 256-point chunks a workgroup of the search kernel walks (1..8).  Since the window scans left that kernel (round 4) it is a light
                                     // streaming pass and fewer, longer workgroups are a little better: 38.9 k (2) against 39.3 k (8) poses/s over five runs each
    int nn_grid = 1;                 // fused path: pixel grid of the scene points (seeds + window search); 0 = tree only
    int host_poll = 1;               // PR_SOLVE_HOST with the sums stored straight into pinned memory: 1 = the host polls per-hypothesis tags and solves each hypothesis as its sums arrive; 0 = it waits for the stream, then solves the group (rounds 2-5)
    int nn_count = 0;                // instrumented runs: the search kernel counts its work per pass (pr_nn_counters)
    int start_overlap = -1;          // asynchronous path, a batch submitted while the other slot is idle: the pass after which the next batch's render may start.
                                     // -1 = the per-batch rule of a running pipeline.  Rounds 2-3 released the next render at pass 0 here; with the roofline
                                     // sample out of the timed region that reads 239 / 240 / 251 k against 253 / 253 / 256 k poses/s for the rule in three
                                     // alternating 20-step runs on one box: batches 1 and 2 then finish together and the third finds an empty chip
    int overlap_pass = -1;           // asynchronous path: the other slot's render may start once this slot has issued this pass of its loop; -1 = chosen per batch
                                     // (refine_submit_async: with 13 passes to go while both slots' clouds fit the Infinity Cache, later by the render's weight when
                                     // they do not; kd-tree scenes: pass 0)
    int sub_batch = 512;             // asynchronous fused path: hypotheses per sub-batch (cache residency of the clouds)
    int fused_solve = 1;             // PR_SOLVE_DEVICE: the workgroup delivering a hypothesis' last partial sum also runs its finalize + solve (no second launch per iteration)
    int use_graph = 1;               // PR_SOLVE_DEVICE, single pose group only (the runtime serialises the branches of a captured multi-stream
                                     // graph, which forfeits the overlap): capture the whole iteration loop in a hipGraph and replay it
    int eager_streams = 1;           // asynchronous path: create the streams of both slots in one run (see slot_streams)
    int raster_mode = 0;             // fused path: 0 = global atomicMin inside the per-pose pixel box (reference scheme), 1 = LDS depth bands (int32); pr_set_option refuses anything else
    int scene_cache = 1;             // keep the packed projective scene / kd traversal records of the latest scene between calls (pr_scene_invalidate)
};
extern Options opt;
// streams a batch is split over: the option, or two.  (Round 3 ran kd-tree scenes as three groups: with 8 workgroups per hypothesis and five
// waves per SIMD the search kernels left gaps a third group filled, +1.5 %.  With round 4's 12 workgroups and six waves two groups are
// enough and the third only costs launches: 34.4 / 34.9 k against 32.2 / 31.9 k poses/s on configs[2], same box; one group 29.8 k, four 31.2 k.
// Projective scenes with 16 hardware queues: three groups 174 k against 247 k.)
inline uint32_t pose_groups_for(int scene_kind) { (void)scene_kind; return opt.pose_groups > 0 ? (uint32_t)opt.pose_groups : 2u; }

// packed projective scene (one 16-byte record per pixel + the two back-projection tables) of the latest scene it was built for
struct PackedCache {
    DevBuf rec;                      // [n] float4, colf[w], rowf[h], then five words: exact flag, sampled fingerprint of the source arrays, its verdict, full fingerprint, a call's full fingerprint
    const void *pcd = nullptr, *normal = nullptr;
    uint64_t w = 0, h = 0; float k[4] = { 0, 0, 0, 0 }; uint32_t tl[2] = { 0, 0 };
    uint64_t gen = 0;
    bool valid = false, exact = false;
};
// everything needed to run a submitted batch again (refine_wait does so when the model box the batch assumed turns out stale)
struct Resubmit {
    const pr_triangle *tris = nullptr; size_t n_tris = 0; uint32_t W = 0, H = 0; pr_mat4 proj{}; float K[9] = { 0 };
    int scene_kind = 0; pr_scene_proj_crop sp{}; pr_scene_nn sn{}; pr_criteria crit{}; pr_roi roi{ 0, 0, 0, 0 };
    pr_result *results_dev = nullptr;
};
// A slot's helper thread (PR_SOLVE_HOST): the reference solves on the host (icp.cu:207), which makes a batch a chain of
// launch -> wait -> solve -> launch that only a host thread can drive; the reference's answer is "many host threads, each refining its own
// hypothesis" (README.md:15).  A caller that pipelines batches through pr_refine_submit / pr_refine_wait from ONE thread gets the same
// overlap from the library: each slot owns a thread with a private context (its own streams, workspaces and caches, as pr_thread_context
// gives a caller's thread) that runs the synchronous host-solve path for the batch while the caller goes on to submit the next one.
struct SlotWorker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    bool has_job = false, done = false, quit = false, alive = false, started = false;
    int device = 0;
    // the job: every input by value (the caller's arrays of poses may go away after pr_refine_submit returns)
    Resubmit in;
    std::vector<pr_mat4> poses;
    pr_result *results_host = nullptr;
    uint32_t *sizes_host = nullptr;
    int rc = PR_OK;
    std::string err;
};
struct Slot {
    std::unique_ptr<SlotWorker> worker;
    bool worker_job = false;         // the batch in flight runs on the helper thread
    DevBuf poses_bbox, depth, row_count, row_off, counts, cloud, meta, partial, dstate, dresults, arrive, aabb_keys, nn_prev;
    PackedCache packed;
    PinBuf h_in, h_out;
    Resubmit again;
    size_t flag_off = 0;             // offset in h_out of the word the device-side model-box check writes (1 = the assumed box was stale)
    hipStream_t stream = nullptr, side[3] = { nullptr, nullptr, nullptr };
    hipEvent_t fork = nullptr, join[3] = { nullptr, nullptr, nullptr }, done = nullptr, scene_ready = nullptr, progress = nullptr;
    bool progress_valid = false;
    bool pending = false, delivered = false;
    int nn_set = -1;                 // the set of derived kd-tree records the batch in flight reads (Ctx::nn_sets), -1: none
    uint32_t P = 0;
    pr_result *user_results_host = nullptr;
    uint32_t *user_sizes = nullptr;
    // a TIMED asynchronous batch (option profile = 3): HIP events on the slot's stream around its render, its cloud emit and every
    // correspondence pass (start / stop pairs in enqueue order), read by pr_refine_wait
    bool timed = false;
    std::vector<hipEvent_t> t_events;            // pool, reused from batch to batch
    size_t t_used = 0;
    struct TSpan { size_t e0, e1; int kind; uint32_t q0, nq; bool edge; bool marks; size_t m[3]; };   // edge: first or last pass (36 B / point instead of 48); marks: events between the kernels of a kd-tree pass
    std::vector<TSpan> t_spans;
};

// ---- hipGraph cache for the device-solve iteration loop ------------------------------------------
struct GraphKey {                    // every value a captured launch depends on, byte for byte (grows as needed)
    std::vector<unsigned char> bytes;
    template <class T> void add(const T &v) { const unsigned char *p = reinterpret_cast<const unsigned char *>(&v); bytes.insert(bytes.end(), p, p + sizeof(T)); }
    bool operator==(const GraphKey &o) const { return bytes == o.bytes; }
};
struct CachedGraph {
    GraphKey key; hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr;
    std::vector<hipEvent_t> events;      // pairs around every correspondence launch when captured with profiling on
    uint64_t stamp = 0;
};
inline void destroy_graph(CachedGraph &c)
{
    if (c.exec) (void)hipGraphExecDestroy(c.exec);
    if (c.graph) (void)hipGraphDestroy(c.graph);
    for (hipEvent_t e : c.events) (void)hipEventDestroy(e);
    c = CachedGraph();
}

constexpr int kNNSets = 2;
struct NNDerived {
    DevBuf topo, bmin, bmax, pts, nnrec, nnrec32, nndesc, nnwide, nnwq, nndepth, nn_cells, nn_grid;
    const void *pcd = nullptr, *normal = nullptr, *nodes = nullptr; uint32_t n_points = 0, n_nodes = 0; uint64_t gen = 0; bool valid = false;
    uint32_t info[24] = { 0 };
    float frame_margin = 0.0f;                                   // the acceptance radius (x 1.01) the wide records' frame was built for: a larger radius rebuilds
    bool grid_valid = false, grid_usable = false; uint32_t gw = 0, gh = 0; float gk[4] = { 0, 0, 0, 0 };
    uint64_t used = 0;                                           // Ctx::nn_clock of the latest call that chose this set
    void release() { for (DevBuf *b : { &topo, &bmin, &bmax, &pts, &nnrec, &nnrec32, &nndesc, &nnwide, &nnwq, &nndepth, &nn_cells, &nn_grid }) b->release(); valid = false; grid_valid = false; }
};
struct Ctx {
    std::mutex mu;                   // one call at a time per context; contexts of different devices / threads run side by side
    std::atomic<int> pins{ 0 };      // pr_free holds a private context alive through this while it drains it WITHOUT g_private_mu (private_unregister_and_wait)
    bool ready = false;
    bool is_private = false;         // pr_thread_context(1): owned by one host thread
    int device = -1;
    hipStream_t stream = nullptr;
    hipStream_t side[3] = { nullptr, nullptr, nullptr };   // extra lanes of the device-solve loop (pose groups overlap one group's solve tail with another group's pass)
    hipEvent_t ev_fork = nullptr, ev_join[3] = { nullptr, nullptr, nullptr };
    int n_cus = 256;
    // host copy of the model box the asynchronous path derives its pixel boxes from: keyed by (pointer, size) and VERIFIED on
    // the device by every batch that uses it (refine_submit / refine_wait), so a rewritten triangle buffer cannot go unnoticed
    float aabb_host[6] = { 0, 0, 0, 0, 0, 0 }; bool aabb_host_valid = false;
    const void *mesh_key = nullptr; size_t mesh_n = 0;   // triangle buffer aabb_host belongs to
    uint32_t cloud_hint = 0;          // largest cloud of the latest finished asynchronous batch: sizes the next batch's grid
    // workspaces
    DevBuf aabb, aabb_keys, bbox, poses, depth, row_count, row_off, counts, cloud, meta, partial, sums, nn_prev, dstate, dresults, arrive, conv16, conv8, kd_scratch, kd_tmp, nn_full;
    PinBuf h_sums, h_meta, h_counts, h_results, h_dstate, h_poses, h_flags;
    PackedCache packed;              // synchronous paths (the asynchronous slots keep their own)
    // kd-tree scenes: what the library derives from a scene (traversal records, wide records, pixel grid) -- kNNSets sets, so that with a NEW scene per
    // frame the batch of one asynchronous slot keeps the set it runs on while the next frame's records are derived into the other (round 5: one set
    // meant draining both slots before every new scene)
    NNDerived nn_sets[kNNSets];
    uint64_t nn_clock = 0;
    DevBuf nn_counters;
    // profiling
    std::vector<hipEvent_t> ev_pool; size_t ev_used = 0;
    struct Span { size_t e0, e1; int kind; };
    std::vector<Span> spans;
    double icp_ms = 0, render_ms = 0, cloud_ms = 0; uint64_t icp_launches = 0, icp_points = 0, icp_bytes = 0, sample_clock = 0;
    std::vector<float> icp_launch_us;                            // the timed launches one by one (pr_profile_launches), bounded
    uint64_t stat_repeated = 0, stat_timing_dropped = 0;         // pr_stats: batches run twice by the stale-cache safety net; timed spans lost to a failed event call
    double nn_part_ms[4] = { 0, 0, 0, 0 }; uint64_t nn_part_n = 0;   // kd-tree pass by kernel: search, bound, task walk, winners pass (pr_profile_nn)
    Slot slots[kSlots];
    std::vector<CachedGraph> graphs;
    uint64_t graph_clock = 0;
    // RCCL communicator this context is a rank of (pr_comm_init_rank / pr_comm_init_all), or null
    ncclComm_t comm = nullptr; int comm_rank = 0, comm_world = 1;
    DevBuf gather_tmp;
    // HIP events around the gathers issued while option "profile" is on (pr_gather_profile)
    std::vector<std::pair<hipEvent_t, hipEvent_t>> gather_ev; size_t gather_ev_used = 0; double gather_ms = 0; uint64_t gather_n = 0;
};

// ---- context registry --------------------------------------------------------------------------------
// One shared context per device (created by pr_init / pr_set_device / first use) plus optional private contexts of single
// host threads (pr_thread_context).  `g` is the context the calling thread is bound to; every entry point binds a thread
// that never chose one to the process default (the device of the first pr_init, else device 0).
extern std::mutex g_reg_mu;
extern std::vector<Ctx *> g_shared;         // index = device ordinal
extern int g_default_device;
extern thread_local Ctx *g;
struct PrivateCtx { Ctx *c = nullptr; ~PrivateCtx(); };   // destructor below, once the teardown helpers exist
extern thread_local PrivateCtx tl_private;
// every private context alive, so that pr_free can drain the device's contexts one by one.  g_private_mu guards the LIST only and is never
// held while a context's mu is taken or waited for (a slot's helper thread registers / unregisters its private context while its
// caller holds the caller's context mutex and waits for it; pr_free on a third thread used to hold g_private_mu while it locked every context
// of the device -- a three-way wait).  pr_free pins the contexts it found (under g_private_mu), drops g_private_mu, then locks them one by
// one; a context that is being released is unregistered first (no new pins), torn down, and deleted once its pins are gone.
extern std::mutex g_private_mu;
extern std::vector<Ctx *> g_private;
inline void private_register(Ctx *c) { std::lock_guard<std::mutex> lk(g_private_mu); g_private.push_back(c); }
inline void private_unregister(Ctx *c) { std::lock_guard<std::mutex> lk(g_private_mu); for (size_t i = 0; i < g_private.size(); ++i) if (g_private[i] == c) { g_private.erase(g_private.begin() + (long)i); break; } }
inline void private_wait_unpinned(Ctx *c) { while (c->pins.load(std::memory_order_acquire) != 0) std::this_thread::yield(); }   // call with NO lock held, after private_unregister

inline void drop_graphs() { for (auto &c : g->graphs) destroy_graph(c); g->graphs.clear(); }
void comm_teardown(Ctx *c);

inline int device_count_checked(int *n_out)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_error("no usable HIP device (hipGetDeviceCount: %s, count %d) -- this library has no CPU fallback",
                  hipGetErrorString(e), n);
        return PR_ERR_NO_DEVICE;
    }
    *n_out = n;
    return PR_OK;
}

// bind the calling thread to the shared context of `device` (-1: the process default), creating it if needed
inline int bind_shared(int device)
{
    std::lock_guard<std::mutex> lk(g_reg_mu);
    if (device < 0) device = g_default_device >= 0 ? g_default_device : 0;
    int n = 0;
    PR_TRY(device_count_checked(&n));
    if (device >= n) { set_error("device %d out of range (%d visible)", device, n); return PR_ERR_NO_DEVICE; }
    if ((int)g_shared.size() < n) g_shared.resize((size_t)n, nullptr);
    if (!g_shared[(size_t)device]) { g_shared[(size_t)device] = new Ctx(); g_shared[(size_t)device]->device = device; }
    if (g_default_device < 0) g_default_device = device;
    g = g_shared[(size_t)device];
    return PR_OK;
}
inline int bind_default() { return g ? PR_OK : bind_shared(-1); }

// with g->mu held: make the context's device current for this thread and create its stream on first use
inline int require_ctx()
{
    HIP_TRY(hipSetDevice(g->device));                            // the current device is per host thread
    if (g->ready) return PR_OK;
    HIP_TRY(hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&g->ev_fork, hipEventDisableTiming));
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, g->device) == hipSuccess && prop.multiProcessorCount > 0) g->n_cus = prop.multiProcessorCount;
    g->ready = true;
    return PR_OK;
}
// every device entry point: bind, lock, make current
#define PR_ENTER()                                   \
    PR_TRY(bind_default());                          \
    std::lock_guard<std::mutex> lk(g->mu);           \
    PR_TRY(require_ctx())

// ---- profiling spans (HIP events on the library stream) ------------------------------------------
enum { kSpanIcp = 0, kSpanRender = 1, kSpanCloud = 2 };
// (a failed event call must not vanish: HIP_TRY clears the sticky error, so the span is dropped HERE and counted -- pr_stats)
constexpr size_t kNoEvent = (size_t)-1;
inline size_t take_event()
{
    if (g->ev_used == g->ev_pool.size()) {
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess || !e) { (void)hipGetLastError(); g->stat_timing_dropped++; return kNoEvent; }
        g->ev_pool.push_back(e);
    }
    return g->ev_used++;
}
inline bool record_event(size_t e, hipStream_t st)
{
    if (e == kNoEvent) return false;
    if (hipEventRecord(g->ev_pool[e], st) != hipSuccess) { (void)hipGetLastError(); g->stat_timing_dropped++; return false; }
    return true;
}
struct SpanGuard {
    bool on; size_t e0 = 0; int kind;
    explicit SpanGuard(int k) : on(opt.profile != 0), kind(k) { if (on) { e0 = take_event(); on = record_event(e0, g->stream); } }
    ~SpanGuard() { if (on) { const size_t e1 = take_event(); if (record_event(e1, g->stream)) g->spans.push_back({ e0, e1, kind }); } }
};
constexpr size_t kLaunchSamples = 8192;
inline void note_launch_us(float ms) { if (g->icp_launch_us.size() < kLaunchSamples) g->icp_launch_us.push_back(ms * 1e3f); }
inline void drain_spans()                   // call after the stream has been synchronised
{
    for (const auto &s : g->spans) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, g->ev_pool[s.e0], g->ev_pool[s.e1]) != hipSuccess) continue;
        if (s.kind == kSpanIcp) { g->icp_ms += ms; g->icp_launches++; note_launch_us(ms); }
        else if (s.kind == kSpanRender) g->render_ms += ms;
        else g->cloud_ms += ms;
    }
    g->spans.clear();
    g->ev_used = 0;
}

// A few words from device memory to the host WITHOUT a copy command: a kernel stores them into the context's pinned flag array, the stream is
// waited for, the host copies them out.  Round 6: copy commands to pageable host memory go through the runtime's bounce buffers and its copy-engine
// path -- tens of microseconds each, and the place where one-off multi-millisecond stalls came from (profiles/r06/README.md).  n_words <= 64.
inline int read_back_words(const void *dev_src, void *host_dst, uint32_t n_words, hipStream_t st)
{
    if (n_words == 0 || n_words > 64) { set_error("read_back_words: 1..64 words"); return PR_ERR_INVALID; }
    PR_TRY(g->h_flags.ensure(256));
    void *vd = nullptr;
    HIP_TRY(hipHostGetDevicePointer(&vd, g->h_flags.p, 0));
    HIP_TRY(prk::launch_copy_words32(dev_src, vd, n_words, st));
    HIP_TRY(hipStreamSynchronize(st));
    std::memcpy(host_dst, g->h_flags.p, sizeof(uint32_t) * n_words);
    return PR_OK;
}

inline void identity16(float *T) { for (int i = 0; i < 16; ++i) T[i] = (i % 5 == 0) ? 1.0f : 0.0f; }

// ---- scene variants -----------------------------------------------------------------------------
struct SceneSel {
    int kind = PR_SCENE_PROJ;
    bool packed = false;
    prk::SceneProjAoS aos{};
    prk::SceneProjPacked pk{};
    prk::SceneNNDev nn{};
    uint32_t nn_split = 0;           // kd-tree scene: search kernel + winners pass instead of the fused search pass
    uint32_t nn_max_points = 0;      // largest cloud of the batch (grid of the search kernel)
    int nn_set = -1;                 // kd-tree scene: which of Ctx::nn_sets the records are in
};
constexpr uint32_t kCounterPasses = 64;
// camera of the hypotheses (fused paths): lets a kd-tree scene be indexed by pixel as well
struct Camera { uint32_t w = 0, h = 0; float fx = 0, fy = 0, cx = 0, cy = 0; };

// ---- functions of the other translation units (pr_scene / pr_icp / pr_refine / pr_comm / pr_context) ---------------------------------
int fingerprint_differs(const void *a, size_t ab, const void *b, size_t bb, const void *c, size_t cb, uint32_t *slot, hipStream_t st, bool &differs);
int ensure_packed(PackedCache &pc, const pr_scene_proj &s, uint32_t tl_x, uint32_t tl_y, hipStream_t st, bool verify_now = false);
int make_scene(int kind, const void *scene, bool want_packed, SceneSel &out, PackedCache *pc_in = nullptr, hipStream_t st = nullptr, const Camera *cam = nullptr,
               bool verify_now = true);
int kd_build_dev(pr_vec3 *pcd, pr_vec3 *nrm, uint32_t n, int max_leaf, pr_kdnode *nodes, size_t cap, uint32_t *n_nodes);
hipError_t launch_pass(const prk::IcpBatch &b, const SceneSel &sc, uint32_t P, hipStream_t st = nullptr, hipEvent_t *nn_marks = nullptr);
int ensure_stream(hipStream_t &st, hipEvent_t *ev = nullptr);
int icp_drive(pr_vec3 *cloud_base, const uint32_t *start_h, const uint32_t *count_h, uint32_t P, const SceneSel &sc_in,
              pr_criteria crit, pr_result *results_host, pr_result *results_dev);
void drain_all_slots();
// a write through the library into caller-owned device memory: logged for the caches (WriteLog) and ordered BEHIND every batch of this context that is still
// in flight and reads the range as part of its scene, mesh or result block (a scene object re-initialised per frame keeps its arrays, so the
// next frame's preparation would otherwise overwrite them under the previous frame's batch).  Call with g->mu held.
void drain_slots_reading(const void *p, size_t bytes);
inline void note_write(const void *p, size_t bytes) { g_writes.note(p, bytes); drain_slots_reading(p, bytes); }
void slot_release(Slot &sl);
void slot_drain(Slot &sl);
void comm_teardown(Ctx *c);
void ctx_teardown(Ctx *c);
int rccl_load();

}  // namespace prr
