// pr_api.cpp -- the C ABI (include/pose_refine.h): device context, workspaces, and the batched
// ICP driver that replaces the per-pose host loop of cuda_icp/icp.cu:156-217.
//
// One library-owned HIP stream; workspaces are grow-only and reused between calls (the reference
// cudaMallocs per call: common.cu:27, renderer.cu:39).  No CPU fallback: every device entry point
// returns PR_ERR_NO_DEVICE when no GPU is usable.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>          // types and prototypes only: librccl is opened on first use (pr_comm_*), never linked
#include <dlfcn.h>

#include <algorithm>
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

namespace prh {
static thread_local std::string g_err;
void set_error(const char *fmt, ...)
{
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    g_err = buf;
}
}  // namespace prh

namespace {

using prh::set_error;

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
WriteLog g_writes;

// ---- process-wide options (pr_set_option): plain ints, shared by every context ---------------------
constexpr int kSlots = 2;
struct Options {
    int pose_groups = 0;             // split the batch over this many streams (1..4); 0 = by scene: 2 for projective scenes (1.31 vs 1.45 ms/step at
                                     // 256 poses), 3 for kd-tree scenes; launches of different groups overlap, so timed calls fall back to one group
    int solve_mode = PR_SOLVE_HOST;
    int host_worker = 1;             // PR_SOLVE_HOST batches of pr_refine_submit run on the slot's helper thread (0: on the caller's thread, inside the call)
    int steps = 3;                   // 1024-point steps per workgroup -> 3072 points per workgroup (9 workgroups per 26 k-point cloud: measured 3-5 % faster than 2048 / 4096)
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
    int nn_count = 0;                // instrumented runs: the search kernel counts its work per pass (pr_nn_counters)
    int start_overlap = -1;          // asynchronous path, a batch submitted while the other slot is idle: the pass after which the next batch's render may start.
                                     // -1 = the per-batch rule of a running pipeline.  Rounds 2-3 released the next render at pass 0 here; with the roofline
                                     // sample out of the timed region that reads 239 / 240 / 251 k against 253 / 253 / 256 k poses/s for the rule in three
                                     // alternating 20-step runs on one box: batches 1 and 2 then finish together and the third finds an empty chip
    int overlap_pass = -1;           // asynchronous path: the other slot's render may start once this slot has issued this pass of its loop (-1: chosen per batch, see refine_submit_async)
                                     // (-1: 70 % of the passes -- measured best of 6/10/14/17 at 256 and 512 poses per batch)
    int sub_batch = 512;             // asynchronous fused path: hypotheses per sub-batch (cache residency of the clouds)
    int fused_solve = 1;             // PR_SOLVE_DEVICE: the workgroup delivering a hypothesis' last partial sum also runs its finalize + solve (no second launch per iteration)
    int icp_flow = 0;                // PR_SOLVE_DEVICE: 1 = one persistent dataflow launch for all iterations (bit-identical; measured equal at
                                     // 256 poses and 35 % slower at 1024), 0 = one launch per pass + per solve
    int use_graph = 1;               // PR_SOLVE_DEVICE, single pose group only (the runtime serialises the branches of a captured multi-stream
                                     // graph, which forfeits the overlap): capture the whole iteration loop in a hipGraph and replay it
    int eager_streams = 1;           // asynchronous path: create the streams of both slots in one run (see slot_streams)
    int raster_mode = 0;             // fused path: 0 = global atomicMin inside the per-pose pixel box (reference scheme), 1 = LDS depth bands (int32),
                                     // 2 = one workgroup per hypothesis with its whole box in LDS as 16-bit depth offsets (global path for boxes that do not fit)
    int scene_cache = 1;             // keep the packed projective scene / kd traversal records of the latest scene between calls (pr_scene_invalidate)
};
Options opt;
// streams a batch is split over: the option, or two.  (Round 3 ran kd-tree scenes as three groups: with 8 workgroups per hypothesis and five
// waves per SIMD the search kernels left gaps a third group filled, +1.5 %.  With round 4's 12 workgroups and six waves two groups are
// enough and the third only costs launches: 34.4 / 34.9 k against 32.2 / 31.9 k poses/s on configs[2], same box; one group 29.8 k, four 31.2 k.
// Projective scenes with 16 hardware queues: three groups 174 k against 247 k.)
inline uint32_t pose_groups_for(int scene_kind) { (void)scene_kind; return opt.pose_groups > 0 ? (uint32_t)opt.pose_groups : 2u; }

// packed projective scene (one 16-byte record per pixel + the two back-projection tables) of the latest scene it was built for
struct PackedCache {
    DevBuf rec;                      // [n] float4, colf[w], rowf[h], exact flag (uint32), sampled fingerprint of the source arrays (uint32)
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
    bool has_job = false, done = false, quit = false, alive = false;
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
void destroy_graph(CachedGraph &c)
{
    if (c.exec) (void)hipGraphExecDestroy(c.exec);
    if (c.graph) (void)hipGraphDestroy(c.graph);
    for (hipEvent_t e : c.events) (void)hipEventDestroy(e);
    c = CachedGraph();
}

struct Ctx {
    std::mutex mu;                   // one call at a time per context; contexts of different devices / threads run side by side
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
    DevBuf aabb, aabb_keys, bbox, poses, depth, row_count, row_off, counts, cloud, meta, partial, sums, topo, bmin, bmax, pts, nnrec, nnrec32, nndesc, nnwide, nnwq, nn_prev, nndepth, dstate, dresults, vbdesc, flowsync, arrive, conv16, conv8, kd_idx, kd_scratch, kd_child, kd_ctrl, kd_tmp, nn_full;
    PinBuf h_sums, h_meta, h_counts, h_results, h_dstate, h_flow;
    PackedCache packed;              // synchronous paths (the asynchronous slots keep their own)
    struct { const void *pcd = nullptr, *normal = nullptr, *nodes = nullptr; uint32_t n_points = 0, n_nodes = 0; uint64_t gen = 0; bool valid = false;
             uint32_t info[24] = { 0 };
             bool grid_valid = false, grid_usable = false; uint32_t gw = 0, gh = 0; float gk[4] = { 0, 0, 0, 0 }; } nn_cache;   // kd traversal records (topo ... nndesc) + pixel grid of the latest kd-tree scene
    DevBuf nn_cells, nn_grid, nn_counters;
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
std::mutex g_reg_mu;
std::vector<Ctx *> g_shared;         // index = device ordinal
int g_default_device = -1;
thread_local Ctx *g = nullptr;
struct PrivateCtx { Ctx *c = nullptr; ~PrivateCtx(); };   // destructor below, once the teardown helpers exist
thread_local PrivateCtx tl_private;
// every private context alive, so that pr_free can drain the device's contexts one by one (lock order: g_private_mu, then a context's mu;
// nothing takes them the other way round: contexts register before and unregister after they are used)
std::mutex g_private_mu;
std::vector<Ctx *> g_private;
void private_register(Ctx *c) { std::lock_guard<std::mutex> lk(g_private_mu); g_private.push_back(c); }
void private_unregister(Ctx *c) { std::lock_guard<std::mutex> lk(g_private_mu); for (size_t i = 0; i < g_private.size(); ++i) if (g_private[i] == c) { g_private.erase(g_private.begin() + (long)i); break; } }

void drop_graphs() { for (auto &c : g->graphs) destroy_graph(c); g->graphs.clear(); }
void comm_teardown(Ctx *c);

int device_count_checked(int *n_out)
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
int bind_shared(int device)
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
int require_ctx()
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
size_t take_event()
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
void drain_spans()                   // call after the stream has been synchronised
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
};
constexpr uint32_t kCounterPasses = 64;
// camera of the hypotheses (fused paths): lets a kd-tree scene be indexed by pixel as well
struct Camera { uint32_t w = 0, h = 0; float fx = 0, fy = 0, cx = 0, cy = 0; };

// Packed copy of a projective scene in `pc`: reused while the caller's arrays are unchanged as far as the library can tell
// (option scene_cache, WriteLog above), rebuilt otherwise.  A rebuild also learns (one 4-byte read-back) whether the pcd array
// is exactly what dep2pcd produces -- only then may the packed form stand in for it.
// Synchronous callers (pr_icp_*, the synchronous fused path) check a cache hit on the spot: the sampled fingerprint of the source arrays
// against the one stored with the cache (`slot[0]`), `slot[1]` receives the verdict.  ~15 us; the asynchronous path checks on its idle stream.
int fingerprint_differs(const void *a, size_t ab, const void *b, size_t bb, const void *c, size_t cb, uint32_t *slot, hipStream_t st, bool &differs)
{
    HIP_TRY(hipMemsetAsync(slot + 1, 0, sizeof(uint32_t), st));
    HIP_TRY(prk::launch_scene_fingerprint(a, ab, b, bb, c, cb, slot, slot + 1, true, st));
    uint32_t flag = 0;
    HIP_TRY(hipMemcpyAsync(&flag, slot + 1, sizeof flag, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    differs = flag != 0u;
    return PR_OK;
}
int ensure_packed(PackedCache &pc, const pr_scene_proj &s, uint32_t tl_x, uint32_t tl_y, hipStream_t st, bool verify_now = false)
{
    const size_t n = (size_t)s.width * s.height;
    const float k[4] = { s.K[0], s.K[4], s.K[2], s.K[5] };
    const bool same = pc.valid && pc.pcd == s.pcd && pc.normal == s.normal && pc.w == s.width && pc.h == s.height &&
                      std::memcmp(pc.k, k, sizeof k) == 0 && pc.tl[0] == tl_x && pc.tl[1] == tl_y;
    if (same && opt.scene_cache && !g_writes.written_since(pc.gen, s.pcd, n * sizeof(pr_vec3)) &&
        !g_writes.written_since(pc.gen, s.normal, n * sizeof(pr_vec3))) {
        bool stale = false;
        if (verify_now) {
            const size_t tb = ((s.width + s.height) * sizeof(float) + 15) & ~(size_t)15;
            uint32_t *ex = reinterpret_cast<uint32_t *>(reinterpret_cast<unsigned char *>(pc.rec.as<float4>() + n) + tb);
            PR_TRY(fingerprint_differs(s.pcd, n * sizeof(pr_vec3), s.normal, n * sizeof(pr_vec3), nullptr, 0, ex + 1, st, stale));
        }
        if (!stale) { pc.gen = g_writes.now(); return PR_OK; }
    }
    pc.valid = false;
    const uint64_t gen = g_writes.now();
    const size_t tables = ((s.width + s.height) * sizeof(float) + 15) & ~(size_t)15;
    PR_TRY(pc.rec.ensure(n * sizeof(float4) + tables + 16));
    float *colf = reinterpret_cast<float *>(pc.rec.as<float4>() + n);
    float *rowf = colf + s.width;
    uint32_t *exact_dev = reinterpret_cast<uint32_t *>(reinterpret_cast<unsigned char *>(colf) + tables);
    HIP_TRY(prk::launch_pack_proj_scene(s.pcd, s.normal, pc.rec.as<float4>(), n, colf, rowf, (uint32_t)s.width, (uint32_t)s.height,
                                        k[0], k[1], k[2], k[3], tl_x, tl_y, exact_dev, st));
    HIP_TRY(prk::launch_scene_fingerprint(s.pcd, n * sizeof(pr_vec3), s.normal, n * sizeof(pr_vec3), nullptr, 0, exact_dev + 1, nullptr, false, st));
    uint32_t exact = 0;
    HIP_TRY(hipMemcpyAsync(&exact, exact_dev, sizeof exact, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    pc.pcd = s.pcd; pc.normal = s.normal; pc.w = s.width; pc.h = s.height; std::memcpy(pc.k, k, sizeof k); pc.tl[0] = tl_x; pc.tl[1] = tl_y;
    pc.gen = gen; pc.exact = exact != 0; pc.valid = true;
    return PR_OK;
}

void drain_all_slots();
int make_scene(int kind, const void *scene, bool want_packed, SceneSel &out, PackedCache *pc_in = nullptr, hipStream_t st = nullptr, const Camera *cam = nullptr,
               bool verify_now = true)
{
    PackedCache &pc = pc_in ? *pc_in : g->packed;
    if (!st) st = g->stream;
    out.kind = kind;
    if (kind == PR_SCENE_PROJ || kind == PR_SCENE_PROJ_CROP) {
        const pr_scene_proj *s = static_cast<const pr_scene_proj *>(scene);       // pr_scene_proj_crop starts with the plain view
        if (!s || !s->pcd || !s->normal || s->width == 0 || s->height == 0) { set_error("invalid pr_scene_proj"); return PR_ERR_INVALID; }
        if (s->width > 0x7fffffffull || s->height > 0x7fffffffull) { set_error("pr_scene_proj: frame too large"); return PR_ERR_INVALID; }
        uint32_t tl_x = 0, tl_y = 0;
        if (kind == PR_SCENE_PROJ_CROP) { const pr_scene_proj_crop *c = static_cast<const pr_scene_proj_crop *>(scene); tl_x = c->tl_x; tl_y = c->tl_y; }
        out.kind = PR_SCENE_PROJ;
        out.aos = prk::SceneProjAoS{ (uint32_t)s->width, (uint32_t)s->height, s->max_dist_diff, s->K[0], s->K[4], s->K[2], s->K[5],
                                     (float)tl_x, (float)tl_y, s->pcd, s->normal };
        out.packed = false;
        if (want_packed) {
            PR_TRY(ensure_packed(pc, *s, tl_x, tl_y, st, verify_now));
            if (pc.exact) {                                         // else: the caller's pcd is not dep2pcd's -- use the arrays as they are
                const size_t n = (size_t)s->width * s->height;
                float *colf = reinterpret_cast<float *>(pc.rec.as<float4>() + n);
                out.packed = true;
                out.pk = prk::SceneProjPacked{ (uint32_t)s->width, (uint32_t)s->height, s->max_dist_diff, s->K[0], s->K[4], s->K[2], s->K[5],
                                               (float)tl_x, (float)tl_y, pc.rec.as<float4>(), colf, colf + s->width };
            }
        }
        return PR_OK;
    }
    if (kind == PR_SCENE_NN) {
        const pr_scene_nn *s = static_cast<const pr_scene_nn *>(scene);
        if (!s || !s->pcd || !s->normal || !s->nodes || s->n_nodes == 0 || s->n_points == 0) { set_error("invalid pr_scene_nn"); return PR_ERR_INVALID; }
        auto &nc = g->nn_cache;
        const bool same = nc.valid && nc.pcd == s->pcd && nc.normal == s->normal && nc.nodes == s->nodes && nc.n_points == s->n_points && nc.n_nodes == s->n_nodes;
        bool hit = same && opt.scene_cache && !g_writes.written_since(nc.gen, s->pcd, (size_t)s->n_points * sizeof(pr_vec3)) &&
                   !g_writes.written_since(nc.gen, s->nodes, (size_t)s->n_nodes * sizeof(pr_kdnode));
        if (hit && verify_now) {
            bool stale = false;
            PR_TRY(fingerprint_differs(s->pcd, (size_t)s->n_points * sizeof(pr_vec3), s->nodes, (size_t)s->n_nodes * sizeof(pr_kdnode), s->normal,
                                       (size_t)s->n_points * sizeof(pr_vec3), g->nndepth.as<uint32_t>() + 12, g->stream, stale));
            hit = !stale;
        }
        if (hit) {
            nc.gen = g_writes.now();                                // (the normals are read through the caller's pointer, never copied)
        } else {
            drain_all_slots();                                      // the records below are shared by both slots' batches
            nc.valid = false; nc.grid_valid = false;
            const uint64_t gen = g_writes.now();
            PR_TRY(g->topo.ensure((size_t)s->n_nodes * sizeof(int4)));
            PR_TRY(g->bmin.ensure((size_t)s->n_nodes * sizeof(float4)));
            PR_TRY(g->bmax.ensure((size_t)s->n_nodes * sizeof(float4)));
            PR_TRY(g->pts.ensure((size_t)s->n_points * sizeof(float4)));
            PR_TRY(g->nnrec.ensure((size_t)s->n_nodes * 4 * sizeof(float4)));
            PR_TRY(g->nndepth.ensure(24 * sizeof(uint32_t)));      // [0] depth [1] rec32 valid [2..7] rec32 frame [8] wide valid [9] wide nodes [12] fingerprint [16..19] wide frame
            PR_TRY(g->nnrec32.ensure((size_t)s->n_nodes * 2 * sizeof(uint4)));
            PR_TRY(g->nndesc.ensure((size_t)s->n_nodes * sizeof(uint2)));
            // wide records: one 128-byte line per wide node
            PR_TRY(g->nnwide.ensure(prk::nn_wide_capacity(s->n_nodes) * 128));
            PR_TRY(g->nnwq.ensure(prk::nn_wide_capacity(s->n_nodes) * 2 * sizeof(uint32_t)));
            HIP_TRY(prk::launch_build_nn_accel(s->nodes, s->n_nodes, s->pcd, s->n_points, g->topo.as<int4>(), g->bmin.as<float4>(),
                                               g->bmax.as<float4>(), g->pts.as<float4>(), g->nnrec.as<float4>(), g->nnrec32.as<uint4>(),
                                               g->nndesc.as<uint2>(), g->nndepth.as<uint32_t>(), g->stream,
                                               g->nnwide.as<uint4>(), g->nnwq.as<uint32_t>(), s->max_dist_diff * 1.01f));
            HIP_TRY(prk::launch_scene_fingerprint(s->pcd, (size_t)s->n_points * sizeof(pr_vec3), s->nodes, (size_t)s->n_nodes * sizeof(pr_kdnode), s->normal,
                                                  (size_t)s->n_points * sizeof(pr_vec3), g->nndepth.as<uint32_t>() + 12, nullptr, false, g->stream));
            HIP_TRY(hipMemcpyAsync(nc.info, g->nndepth.p, sizeof nc.info, hipMemcpyDeviceToHost, g->stream));
            HIP_TRY(hipStreamSynchronize(g->stream));
            nc.pcd = s->pcd; nc.normal = s->normal; nc.nodes = s->nodes; nc.n_points = s->n_points; nc.n_nodes = s->n_nodes; nc.gen = gen; nc.valid = true;
        }
        const uint32_t *info = nc.info;
        const uint32_t depth = info[0];
        if (depth >= 0x7fffffffu) { nc.valid = false; set_error("pr_scene_nn: the nodes do not form a consistent tree (child / parent links disagree or point outside the array)"); return PR_ERR_INVALID; }
        uint32_t lds = (uint32_t)std::max(0, opt.nn_lds_nodes);
        lds = std::min(lds, s->n_nodes);
        lds = std::min<uint32_t>(lds, 8192);                      // <= 128 KiB of LDS
        // pending far children on the stack never exceed the tree depth; deeper trees use the stackless walk
        uint32_t stack = 0;
        if (opt.nn_stack) stack = (depth <= 16) ? 16u : ((depth <= 24) ? 24u : 0u);
        if (stack) lds = std::min<uint32_t>({ (uint32_t)std::max(0, opt.nn_lds_records), s->n_nodes, (65536u - stack * 2048u) / 64u });   // stacks + records <= 64 KiB
        out.nn = prk::SceneNNDev{ s->max_dist_diff, g->topo.as<int4>(), g->bmin.as<float4>(), g->bmax.as<float4>(), g->pts.as<float4>(),
                                  s->pcd, s->normal, s->n_nodes, lds, g->nnrec.as<float4>(), stack, nullptr, { 0, 0, 0 }, { 0, 0, 0 }, g->nndesc.as<uint2>() };
        if (stack && opt.nn_compact && info[1] == 1u) {
            out.nn.rec32 = g->nnrec32.as<uint4>();
            for (int a = 0; a < 3; ++a) { std::memcpy(&out.nn.qmin[a], &info[2 + a], 4); std::memcpy(&out.nn.qscale[a], &info[5 + a], 4); }
            if (opt.nn_wide && info[8] == 1u && info[9] > 0u) {
                out.nn.wide = g->nnwide.as<uint4>(); out.nn.n_wide = info[9];
                for (int a = 0; a < 3; ++a) std::memcpy(&out.nn.wmin[a], &info[16 + a], 4);
                std::memcpy(&out.nn.wscale, &info[19], 4);
            }
        }
        // a bare ICP call has no camera of its own: the scene's hint, if it carries one (pose_refine.h)
        Camera hinted;
        if (!cam && s->cam_magic == PR_SCENE_NN_CAM_MAGIC && s->cam_w && s->cam_h && s->cam_fx > 0.0f && s->cam_fy > 0.0f) { hinted = Camera{ s->cam_w, s->cam_h, s->cam_fx, s->cam_fy, s->cam_cx, s->cam_cy }; cam = &hinted; }
        // pixel grid of the scene points under the hypotheses' camera (fused paths), or under the camera the scene says it was made with.  Usable
        // when every scene point owns a cell -- a Scene_nn made from a depth image with these intrinsics -- else the tree alone.
        if (cam && opt.nn_grid && out.nn.rec32 && (size_t)cam->w * cam->h <= ((size_t)1 << 24)) {
            const float gk[4] = { cam->fx, cam->fy, cam->cx, cam->cy };
            if (!(nc.grid_valid && nc.gw == cam->w && nc.gh == cam->h && std::memcmp(nc.gk, gk, sizeof gk) == 0)) {
                drain_all_slots();
                const size_t cells = (size_t)cam->w * cam->h;
                PR_TRY(g->nn_cells.ensure(cells * sizeof(int32_t) + 16));
                PR_TRY(g->nn_grid.ensure(prk::nn_grid_cells(cam->w, cam->h) * sizeof(float4)));
                uint32_t *flag = reinterpret_cast<uint32_t *>(g->nn_cells.as<int32_t>() + cells);
                HIP_TRY(prk::launch_build_nn_grid(s->pcd, s->n_points, cam->w, cam->h, gk[0], gk[1], gk[2], gk[3], g->nn_cells.as<int32_t>(),
                                                  g->nn_grid.as<float4>(), flag, g->stream));
                uint32_t usable = 0;
                HIP_TRY(hipMemcpyAsync(&usable, flag, sizeof usable, hipMemcpyDeviceToHost, g->stream));
                HIP_TRY(hipStreamSynchronize(g->stream));
                nc.grid_valid = true; nc.grid_usable = usable != 0; nc.gw = cam->w; nc.gh = cam->h; std::memcpy(nc.gk, gk, sizeof gk);
            }
            if (nc.grid_usable) {
                out.nn.grid = g->nn_grid.as<float4>(); out.nn.gw = cam->w; out.nn.gh = cam->h; out.nn.gfx = gk[0]; out.nn.gfy = gk[1]; out.nn.gcx = gk[2]; out.nn.gcy = gk[3];
                const size_t w4 = (cam->w + 3) / 4, h4 = (cam->h + 3) / 4, w16 = (w4 + 3) / 4, h16 = (h4 + 3) / 4;
                out.nn.pyr4 = out.nn.grid + (size_t)cam->w * cam->h; out.nn.pyr16 = out.nn.pyr4 + w4 * h4; out.nn.pyr64 = out.nn.pyr16 + w16 * h16;
            }
        }
        return PR_OK;
    }
    set_error("unknown scene kind %d", kind);
    return PR_ERR_INVALID;
}

hipError_t launch_pass(const prk::IcpBatch &b, const SceneSel &sc, uint32_t P, hipStream_t st = nullptr, hipEvent_t *nn_marks = nullptr)
{
    if (!st) st = g->stream;
    if (sc.kind == PR_SCENE_NN) {
        if (sc.nn_split && b.nn_prev) {                            // search (applies the pending update, leaves the winners in nn_prev) ...
            hipError_t e = prk::launch_nn_search(b, sc.nn, P, sc.nn_max_points, (uint32_t)std::max(1, opt.nn_run), st, nn_marks);
            if (e != hipSuccess) return e;
            prk::IcpBatch bb = b;                                   // ... then the canonical-order pass over the winners
            bb.pre_transformed = 1;
            const prk::SceneNNWinners w{ sc.nn.max_dist_diff, sc.nn.pts, sc.nn.normal, b.nn_prev };
            return prk::launch_icp_pass_nn_winners(bb, w, P, st);
        }
        return prk::launch_icp_pass_nn(b, sc.nn, P, st);
    }
    if (sc.packed) return prk::launch_icp_pass_proj_packed(b, sc.pk, P, st);
    return prk::launch_icp_pass_proj_aos(b, sc.aos, P, st);
}

// Streams are created only when they are first needed: the runtime multiplexes all streams of a process onto a handful of
// hardware queues (4 by default) in creation order, and two streams that share a queue serialise -- with every possible
// stream created up front, a slot's render stream ended up behind the other slot's 21 queued passes.
int ensure_stream(hipStream_t &st, hipEvent_t *ev = nullptr)
{
    if (!st) HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    if (ev && !*ev) HIP_TRY(hipEventCreateWithFlags(ev, hipEventDisableTiming));
    return PR_OK;
}

// ---- the batched ICP driver -----------------------------------------------------------------------
// clouds: cloud i = cloud_base[start_h[i] .. start_h[i]+count_h[i]).  start/count must already be in
// g->start / g->counts on the device when dev_meta_ready, otherwise they are uploaded here.
int icp_drive(pr_vec3 *cloud_base, const uint32_t *start_h, const uint32_t *count_h, uint32_t P, const SceneSel &sc_in,
              pr_criteria crit, pr_result *results_host, pr_result *results_dev)
{
    if (P == 0) return PR_OK;
    SceneSel sc = sc_in;
    if (crit.max_iteration < 0) { set_error("max_iteration must be >= 0"); return PR_ERR_INVALID; }
    const uint32_t steps = (uint32_t)std::max(1, opt.steps);
    const uint32_t ppb = steps * prk::kPointsPerStep;
    uint32_t max_n = 0; uint64_t sum_n = 0;
    for (uint32_t i = 0; i < P; ++i) { max_n = std::max(max_n, count_h[i]); sum_n += count_h[i]; }
    const uint32_t nblk = (max_n + ppb - 1) / ppb;

    PR_TRY(g->meta.ensure(sizeof(prk::PoseMeta) * P));
    PR_TRY(g->partial.ensure(sizeof(float) * prk::kAccStride * (size_t)std::max(1u, nblk) * P));
    PR_TRY(g->sums.ensure(sizeof(float) * prk::kAccStride * P));
    PR_TRY(g->h_meta.ensure(sizeof(prk::PoseMeta) * P));
    PR_TRY(g->h_sums.ensure(sizeof(float) * prk::kAccStride * P));
    PR_TRY(g->h_results.ensure(sizeof(pr_result) * P));

    prk::IcpBatch b{};
    b.cloud = cloud_base; b.meta = g->meta.as<prk::PoseMeta>(); b.partial = g->partial.as<float>();
    b.nblk = nblk; b.steps = steps;
    sc.nn_split = (sc.kind == PR_SCENE_NN && sc.nn.rec32 && opt.nn_split && !opt.icp_flow) ? 1u : 0u;
    sc.nn_max_points = max_n;
    if (sc.nn_split && opt.nn_count) {                            // instrumented run: kCounterPasses x 8 counters, accumulated until read
        const bool fresh = g->nn_counters.p == nullptr;
        PR_TRY(g->nn_counters.ensure(sizeof(unsigned long long) * 8 * kCounterPasses));
        if (fresh) HIP_TRY(hipMemsetAsync(g->nn_counters.p, 0, sizeof(unsigned long long) * 8 * kCounterPasses, g->stream));
        if ((uint32_t)crit.max_iteration + 1 <= kCounterPasses) sc.nn.counters = g->nn_counters.as<unsigned long long>();
    }
    if (sc.kind == PR_SCENE_NN && sc.nn.rec32 && (opt.nn_seed || sc.nn_split)) {     // winners, indexed like the cloud points
        size_t span = 1;
        for (uint32_t i = 0; i < P; ++i) span = std::max(span, (size_t)start_h[i] + count_h[i]);
        // winners | slack of the keep-the-winner test | queue 1 and queue 2 of unsettled queries (8 B per entry) | queue counters per hypothesis
        PR_TRY(g->nn_prev.ensure(sizeof(uint32_t) * (span * prk::kNNWordsPerPoint + prk::kQCountStride * (size_t)P) + 64));
        b.nn_prev = g->nn_prev.as<uint32_t>();
        b.nn_slack = reinterpret_cast<float *>(b.nn_prev + span);
        b.nn_queue = reinterpret_cast<uint2 *>(b.nn_prev + 2 * span);
        b.nn_queue2 = reinterpret_cast<uint2 *>(b.nn_prev + 4 * span);
        b.nn_qcount = b.nn_prev + 6 * span;
        HIP_TRY(hipMemsetAsync(b.nn_qcount, 0, sizeof(uint32_t) * prk::kQCountStride * P, g->stream));
    }

    prk::PoseMeta *h_meta = g->h_meta.as<prk::PoseMeta>();
    float *h_sums = g->h_sums.as<float>();
    pr_result *res = g->h_results.as<pr_result>();
    for (uint32_t i = 0; i < P; ++i) {
        identity16(res[i].T); res[i].inlier_rmse = 0.0f; res[i].fitness = 0.0f;   // icp.h:29-31
        std::memset(&h_meta[i], 0, sizeof(prk::PoseMeta));
        h_meta[i].start = start_h[i]; h_meta[i].count = count_h[i];
        h_meta[i].state = (count_h[i] > 0) ? prk::kRun : prk::kSkip;    // empty cloud: count==0 -> identity result (icp.cu:183)
    }

    if (opt.solve_mode == PR_SOLVE_DEVICE) {
        PR_TRY(g->dstate.ensure(sizeof(prk::DevIcpState) * P));
        PR_TRY(g->h_dstate.ensure(sizeof(prk::DevIcpState) * P));
        prk::DevIcpState *init = g->h_dstate.as<prk::DevIcpState>();
        for (uint32_t i = 0; i < P; ++i) { identity16(init[i].T); init[i].fitness = 0; init[i].rmse = 0; init[i].done = (h_meta[i].state == prk::kSkip); init[i].passes = 0; }
        pr_result *dres = results_dev;
        if (!dres) { PR_TRY(g->dresults.ensure(sizeof(pr_result) * P)); dres = g->dresults.as<pr_result>(); }
        const bool may_exit_early = (crit.relative_fitness > 0.0f && crit.relative_rmse > 0.0f);
        const bool fused = opt.fused_solve != 0;
        if (fused) PR_TRY(g->arrive.ensure(sizeof(uint32_t) * P));

        if (opt.icp_flow) {
            // ---- dataflow path: one persistent launch runs every iteration of every hypothesis ------------------
            std::vector<uint2> desc;
            desc.reserve((size_t)P * std::max(1u, nblk));
            for (uint32_t i = 0; i < P; ++i) {
                const uint32_t nb = (count_h[i] + ppb - 1) / ppb;
                for (uint32_t gi = 0; gi < nb; ++gi) desc.push_back(make_uint2(i, gi));
            }
            const uint32_t n_vbs = (uint32_t)desc.size();
            const size_t sync_words = (size_t)2 * P + 1;
            PR_TRY(g->vbdesc.ensure(sizeof(uint2) * std::max<size_t>(1, n_vbs)));
            PR_TRY(g->flowsync.ensure(sizeof(uint32_t) * sync_words));
            PR_TRY(g->h_flow.ensure(sizeof(uint2) * std::max<size_t>(1, n_vbs) + sizeof(uint32_t) * (sync_words + 1)));
            uint32_t *h_sync = g->h_flow.as<uint32_t>();
            uint2 *h_desc = reinterpret_cast<uint2 *>(h_sync + ((sync_words + 2) & ~(size_t)1));
            for (uint32_t i = 0; i < P; ++i) { h_sync[i] = 0; h_sync[P + i] = (h_meta[i].state == prk::kSkip) ? 0xffffffffu : 0u; }
            h_sync[2 * P] = 0;
            if (n_vbs) std::memcpy(h_desc, desc.data(), sizeof(uint2) * n_vbs);
            HIP_TRY(hipMemcpyAsync(g->dstate.p, init, sizeof(prk::DevIcpState) * P, hipMemcpyHostToDevice, g->stream));
            HIP_TRY(hipMemcpyAsync(g->meta.p, h_meta, sizeof(prk::PoseMeta) * P, hipMemcpyHostToDevice, g->stream));
            HIP_TRY(hipMemcpyAsync(g->flowsync.p, h_sync, sizeof(uint32_t) * sync_words, hipMemcpyHostToDevice, g->stream));
            if (n_vbs) {
                HIP_TRY(hipMemcpyAsync(g->vbdesc.p, h_desc, sizeof(uint2) * n_vbs, hipMemcpyHostToDevice, g->stream));
                prk::FlowArgs fa{};
                fa.cloud = cloud_base; fa.meta = g->meta.as<prk::PoseMeta>(); fa.partial = g->partial.as<float>();
                fa.st = g->dstate.as<prk::DevIcpState>(); fa.vb_desc = g->vbdesc.as<uint2>();
                fa.arrive = g->flowsync.as<uint32_t>(); fa.ready = fa.arrive + P; fa.abort_flag = fa.arrive + 2 * P;
                fa.n_vbs = n_vbs; fa.nblk = nblk; fa.steps = steps; fa.crit = crit;
                uint32_t grid = 0;
                SpanGuard sp(kSpanIcp);
                if (sc.kind == PR_SCENE_NN) HIP_TRY(prk::launch_icp_flow_nn(fa, sc.nn, (uint32_t)g->n_cus, g->stream, &grid));
                else if (sc.packed) HIP_TRY(prk::launch_icp_flow_proj_packed(fa, sc.pk, (uint32_t)g->n_cus, g->stream, &grid));
                else HIP_TRY(prk::launch_icp_flow_proj_aos(fa, sc.aos, (uint32_t)g->n_cus, g->stream, &grid));
                if (opt.profile) {                                  // one launch = all passes: 36 B/point on pass 0, 48 B/point afterwards
                    g->icp_points += sum_n * (uint64_t)(crit.max_iteration + 1);
                    g->icp_bytes += sum_n * (36ull + 48ull * (uint64_t)crit.max_iteration);
                }
            }
            HIP_TRY(prk::launch_pack_results(g->dstate.as<prk::DevIcpState>(), dres, P, g->stream));
            if (results_host) HIP_TRY(hipMemcpyAsync(res, dres, sizeof(pr_result) * P, hipMemcpyDeviceToHost, g->stream));
            HIP_TRY(hipMemcpyAsync(h_sync + 2 * P, g->flowsync.as<uint32_t>() + 2 * P, sizeof(uint32_t), hipMemcpyDeviceToHost, g->stream));
            HIP_TRY(hipStreamSynchronize(g->stream));
            drain_spans();
            if (h_sync[2 * P] != 0) { set_error("dataflow ICP kernel timed out waiting on a hypothesis (workgroups not co-resident?)"); return PR_ERR_HIP; }
            if (results_host) std::memcpy(results_host, res, sizeof(pr_result) * P);
            return PR_OK;
        }

        // profile==2: every sample_period-th call is a timed call -- it runs synchronously, as one pose group, with the other slot
        // drained, and times every correspondence launch of its loop
        const uint64_t tick = g->sample_clock++;
        const bool sample_call = (opt.profile == 2) && (tick % (uint64_t)std::max(1, opt.sample_period) == 0);
        // Pose groups: the batch is split over up to four streams so that one group's serial solve tail (and the ragged end
        // of its pass) overlaps another group's pass.  Timed launches (profile 1, the sampled call of profile 2) run as a
        // single group so that the measured kernel has the chip to itself.
        const bool timed_call = (opt.profile == 1 || opt.profile == 3) || sample_call;   // (profile 3 only keeps SUBMITTED batches asynchronous: whatever runs here is a synchronous timed call)
        const uint32_t n_groups = timed_call ? 1u : std::max(1u, std::min({ pose_groups_for(sc.kind), 4u, P / 32u }));
        // the fixed sequence: state upload, (max_iteration+1) x [correspondence pass, finalize+solve], result pack
        auto enqueue_all = [&](std::vector<hipEvent_t> *, bool host_checks) -> int {
            HIP_TRY(hipMemcpyAsync(g->dstate.p, init, sizeof(prk::DevIcpState) * P, hipMemcpyHostToDevice, g->stream));
            HIP_TRY(hipMemcpyAsync(g->meta.p, h_meta, sizeof(prk::PoseMeta) * P, hipMemcpyHostToDevice, g->stream));
            if (fused) HIP_TRY(hipMemsetAsync(g->arrive.p, 0, sizeof(uint32_t) * P, g->stream));
            // Two pose groups on two streams: while the (latency-bound, one wavefront per pose) finalize+solve
            // of one group runs, the correspondence pass of the other group keeps the chip busy.
            auto group_begin = [&](uint32_t grp) { return (uint32_t)(((uint64_t)P * grp) / n_groups); };
            if (n_groups > 1) {
                for (uint32_t k = 1; k < n_groups; ++k) PR_TRY(ensure_stream(g->side[k - 1], &g->ev_join[k - 1]));
                HIP_TRY(hipEventRecord(g->ev_fork, g->stream));
                for (uint32_t k = 1; k < n_groups; ++k) HIP_TRY(hipStreamWaitEvent(g->side[k - 1], g->ev_fork, 0));
            }
            for (uint32_t it = 0; it <= (uint32_t)crit.max_iteration; ++it) {
                for (uint32_t grp = 0; grp < n_groups; ++grp) {
                    const uint32_t p0 = group_begin(grp), np = group_begin(grp + 1) - p0;
                    hipStream_t st = grp ? g->side[grp - 1] : g->stream;
                    prk::IcpBatch bb = b;
                    bb.meta += p0; bb.partial += (size_t)p0 * nblk * prk::kAccStride; if (bb.nn_qcount) bb.nn_qcount += prk::kQCountStride * (size_t)p0;
                    bb.iter = it;
                    if (fused) { bb.fused = 1; bb.crit = crit; bb.st = g->dstate.as<prk::DevIcpState>() + p0; bb.arrive = g->arrive.as<uint32_t>() + p0; }
                    bb.score_only = (it == (uint32_t)crit.max_iteration) ? 1u : 0u;
                    if (grp == 0 && timed_call) {           // a timed call times every launch of its loop
                        SpanGuard sp(kSpanIcp); HIP_TRY(launch_pass(bb, sc, np, st));
                        uint64_t pts = 0; for (uint32_t i = p0; i < p0 + np; ++i) pts += count_h[i];
                        // algorithmic bytes per point: 12 read + 24 gathered, + 12 written back once a transform is
                        // pending; the score-only last pass needs the scene point but not its normal (12 + 12 + 12)
                        const bool first = (it == 0), last = (it == (uint32_t)crit.max_iteration);
                        g->icp_points += pts; g->icp_bytes += pts * ((first || last) ? 36u : 48u);
                    } else HIP_TRY(launch_pass(bb, sc, np, st));
                    if (!fused) HIP_TRY(prk::launch_icp_finalize_solve(bb.partial, g->meta.as<prk::PoseMeta>() + p0, nblk, steps,
                                                                       g->dstate.as<prk::DevIcpState>() + p0, crit, it, np, st));
                }
                if (host_checks && may_exit_early && (it & 3) == 3 && it < (uint32_t)crit.max_iteration) {
                    for (uint32_t k = 1; k < n_groups; ++k) HIP_TRY(hipStreamSynchronize(g->side[k - 1]));
                    HIP_TRY(hipMemcpyAsync(h_meta, g->meta.p, sizeof(prk::PoseMeta) * P, hipMemcpyDeviceToHost, g->stream));
                    HIP_TRY(hipStreamSynchronize(g->stream));
                    bool any = false;
                    for (uint32_t i = 0; i < P; ++i) any |= (h_meta[i].state != prk::kSkip);
                    if (!any) break;
                }
            }
            for (uint32_t k = 1; k < n_groups; ++k) { HIP_TRY(hipEventRecord(g->ev_join[k - 1], g->side[k - 1])); HIP_TRY(hipStreamWaitEvent(g->stream, g->ev_join[k - 1], 0)); }
            HIP_TRY(prk::launch_pack_results(g->dstate.as<prk::DevIcpState>(), dres, P, g->stream));
            // results go to the pinned staging buffer (a pageable destination is not capturable)
            if (results_host) HIP_TRY(hipMemcpyAsync(res, dres, sizeof(pr_result) * P, hipMemcpyDeviceToHost, g->stream));
            return PR_OK;
        };

        if (opt.use_graph && n_groups == 1 && (opt.profile == 0 || (opt.profile == 2 && !sample_call))) {   // HIP events recorded inside a captured graph cannot be timed
            GraphKey key;
            key.add(P); key.add(nblk); key.add(steps); key.add(crit); key.add(sc); key.add(cloud_base); key.add(g->meta.p); key.add(g->partial.p);
            key.add(g->dstate.p); key.add(dres); key.add(results_host != nullptr); key.add(res); key.add(h_meta); key.add(init); key.add(opt.profile); key.add(pose_groups_for(sc.kind)); key.add(opt.fused_solve); key.add(g->arrive.p); key.add(b.nn_prev);
            key.add(b.nn_slack); key.add(b.nn_queue); key.add(b.nn_queue2); key.add(b.nn_qcount);    // laid out behind nn_prev at multiples of `span` (max start+count): same P / max_n, other offsets => other addresses
            CachedGraph *hit = nullptr;
            for (auto &c : g->graphs) if (c.exec && c.key == key) { hit = &c; break; }
            if (!hit) {
                if (g->graphs.size() >= 8) {                         // evict the least recently used entry
                    size_t lru = 0;
                    for (size_t i = 1; i < g->graphs.size(); ++i) if (g->graphs[i].stamp < g->graphs[lru].stamp) lru = i;
                    destroy_graph(g->graphs[lru]);
                    g->graphs.erase(g->graphs.begin() + lru);
                }
                CachedGraph c; c.key = key;
                HIP_TRY(hipStreamSynchronize(g->stream));
                HIP_TRY(hipStreamBeginCapture(g->stream, hipStreamCaptureModeThreadLocal));
                int rc = enqueue_all(nullptr, /*host_checks=*/false);
                hipError_t ce = hipStreamEndCapture(g->stream, &c.graph);
                if (rc != PR_OK || ce != hipSuccess) { destroy_graph(c); if (rc == PR_OK) { set_error("hipStreamEndCapture failed: %s", hipGetErrorString(ce)); rc = PR_ERR_HIP; } return rc; }
                HIP_TRY(hipGraphInstantiate(&c.exec, c.graph, nullptr, nullptr, 0));
                g->graphs.push_back(std::move(c));
                hit = &g->graphs.back();
            }
            hit->stamp = ++g->graph_clock;
            HIP_TRY(hipGraphLaunch(hit->exec, g->stream));
            HIP_TRY(hipStreamSynchronize(g->stream));
            if (results_host) std::memcpy(results_host, res, sizeof(pr_result) * P);
            drain_spans();
            return PR_OK;
        }

        PR_TRY(enqueue_all(nullptr, /*host_checks=*/true));
        HIP_TRY(hipStreamSynchronize(g->stream));
        if (results_host) std::memcpy(results_host, res, sizeof(pr_result) * P);
        drain_spans();
        return PR_OK;
    }

    // PR_SOLVE_HOST: per iteration one launch + one small D2H, then the per-pose logic of icp.cu:178-212 on the host.  The batch runs
    // as up to four pose groups, each on a stream of its own, software-pipelined: while the host waits for, solves and re-arms one
    // group, the correspondence pass of the next group is on the chip -- the host's share of an iteration (copy latency, wake-up,
    // 160 ns per 6x6 solve) is as long as the pass itself, and a single group leaves the GPU idle for all of it.  Results per
    // hypothesis do not depend on the grouping (sums, solve and state are per hypothesis).  A timed call (profile 1) is one group.
    // profile 2: one call in sample_period times ONE of its passes (a different iteration from call to call); that call runs as one pose
    // group like the calls of profile 1 / 3, so that the timed launch has the chip to itself -- the other calls keep their pipeline
    const uint64_t host_tick = g->sample_clock++;
    const bool host_sample_call = (opt.profile == 2) && (host_tick % (uint64_t)std::max(1, opt.sample_period) == 0);
    const uint32_t host_sample_it = (uint32_t)(((host_tick / (uint64_t)std::max(1, opt.sample_period)) * 5 + 1) % (uint64_t)(crit.max_iteration + 1));
    const uint32_t n_groups = (opt.profile == 1 || opt.profile == 3 || host_sample_call) ? 1u : std::max(1u, std::min({ pose_groups_for(sc.kind), 4u, P / 32u }));
    auto group_begin = [&](uint32_t grp) { return (uint32_t)(((uint64_t)P * grp) / n_groups); };
    if (opt.fused_solve) { PR_TRY(g->arrive.ensure(sizeof(uint32_t) * P)); HIP_TRY(hipMemsetAsync(g->arrive.p, 0, sizeof(uint32_t) * P, g->stream)); }
    if (n_groups > 1) {
        for (uint32_t k = 1; k < n_groups; ++k) PR_TRY(ensure_stream(g->side[k - 1], &g->ev_join[k - 1]));
        HIP_TRY(hipEventRecord(g->ev_fork, g->stream));               // the groups start behind the render / cloud work of this call
        for (uint32_t k = 1; k < n_groups; ++k) HIP_TRY(hipStreamWaitEvent(g->side[k - 1], g->ev_fork, 0));
    }
    auto group_stream = [&](uint32_t grp) { return grp ? g->side[grp - 1] : g->stream; };
    // With option fused_solve the pass also adds up the workgroup sums of a hypothesis (the workgroup that arrives last does, as in the
    // device-solve loop) and stores the 29 totals straight into the pinned host array: no finalize launch, no copy command -- two
    // API calls and one copy-engine round trip less per group and iteration on a path that is bound by exactly those.
    const bool host_fused = opt.fused_solve != 0;
    float *sums_dev = nullptr;
    if (host_fused) {
        PR_TRY(g->arrive.ensure(sizeof(uint32_t) * P));
        HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void **>(&sums_dev), h_sums, 0));
    }
    // Projective scenes also read the per-hypothesis state (64 bytes: cloud span, pending update) from the pinned host array instead
    // of receiving it through a copy command: one uniform load per workgroup over the host link, 1.62 -> 1.54 ms per 256-hypothesis
    // batch.  The four kernels of a kd-tree pass have ten times the workgroups; there the copy is cheaper (8.3 against 8.8 ms).
    const prk::PoseMeta *meta_dev = nullptr;
    if (host_fused && sc.kind != PR_SCENE_NN) HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void **>(const_cast<prk::PoseMeta **>(&meta_dev)), h_meta, 0));
    // one iteration of one group goes onto its stream: state upload, pass, block sums -> pose sums, download
    auto enqueue_group = [&](uint32_t grp, uint32_t it) -> int {
        const uint32_t p0 = group_begin(grp), np = group_begin(grp + 1) - p0;
        hipStream_t st = group_stream(grp);
        prk::IcpBatch bb = b;
        if (meta_dev) bb.meta = meta_dev;                            // the pass reads the 64-byte state of its hypothesis from the pinned host array
        else HIP_TRY(hipMemcpyAsync(g->meta.as<prk::PoseMeta>() + p0, h_meta + p0, sizeof(prk::PoseMeta) * np, hipMemcpyHostToDevice, st));
        bb.meta += p0; bb.partial += (size_t)p0 * nblk * prk::kAccStride; if (bb.nn_qcount) bb.nn_qcount += prk::kQCountStride * (size_t)p0;
        bb.iter = it;
        if (host_fused) { bb.fused = 2; bb.arrive = g->arrive.as<uint32_t>() + p0; bb.sums_out = sums_dev + (size_t)p0 * prk::kAccStride; }
        if (opt.profile == 1 || opt.profile == 3 || (host_sample_call && it == host_sample_it)) {
            SpanGuard sp(kSpanIcp); HIP_TRY(launch_pass(bb, sc, np, st));
            for (uint32_t i = p0; i < p0 + np; ++i) if (h_meta[i].state != prk::kSkip) { g->icp_points += count_h[i]; g->icp_bytes += (uint64_t)count_h[i] * (it == 0 ? 36u : 48u); }
        } else HIP_TRY(launch_pass(bb, sc, np, st));
        if (!host_fused) {
            HIP_TRY(prk::launch_icp_finalize(bb.partial, g->meta.as<prk::PoseMeta>() + p0, nblk, steps, g->sums.as<float>() + (size_t)p0 * prk::kAccStride, np, st));
            HIP_TRY(hipMemcpyAsync(h_sums + (size_t)p0 * prk::kAccStride, g->sums.as<float>() + (size_t)p0 * prk::kAccStride, sizeof(float) * prk::kAccStride * np,
                                   hipMemcpyDeviceToHost, st));
        }
        return PR_OK;
    };
    // the host's part of an iteration for one group (its download has completed); returns how many of its hypotheses go on
    auto solve_group = [&](uint32_t grp, uint32_t it) -> uint32_t {
        uint32_t active = 0;
        for (uint32_t i = group_begin(grp); i < group_begin(grp + 1); ++i) {
            if (h_meta[i].state == prk::kSkip) continue;
            const float *Ab = h_sums + (size_t)i * prk::kAccStride;
            pr_result &r = res[i];
            const float prev_fit = r.fitness, prev_rmse = r.inlier_rmse;
            const float cnt = Ab[28], err = Ab[27];
            if (cnt == 0) { h_meta[i].state = prk::kSkip; continue; }                        // icp.cu:183
            r.fitness = cnt / (float)count_h[i];                                          // icp.cu:185
            r.inlier_rmse = std::sqrt(err / cnt);                                         // icp.cu:186
            if (it == (uint32_t)crit.max_iteration) { h_meta[i].state = prk::kSkip; continue; }   // icp.cu:189
            if (std::fabs(r.fitness - prev_fit) < crit.relative_fitness &&
                std::fabs(r.inlier_rmse - prev_rmse) < crit.relative_rmse) { h_meta[i].state = prk::kSkip; continue; }   // icp.cu:191-194
            float A[36], bb[6], E[16];
            for (int k = 0; k < 6; ++k) bb[k] = Ab[21 + k];
            int sh = 0;
            for (int y = 0; y < 6; ++y) for (int x = y; x < 6; ++x) { A[x + y * 6] = Ab[sh]; A[y + x * 6] = Ab[sh]; ++sh; }   // icp.cu:196-205
            prh::solve_666(A, bb, E);
            std::memcpy(h_meta[i].xform, E, sizeof(float) * 12);
            prh::mat4_mul(E, r.T, r.T);                                                   // icp.cu:212
            h_meta[i].state = prk::kRunWithTransform;
            ++active;
        }
        return active;
    };
    bool live[4] = { false, false, false, false };
    int rc_loop = PR_OK;
    for (uint32_t grp = 0; grp < n_groups && rc_loop == PR_OK; ++grp) {
        for (uint32_t i = group_begin(grp); i < group_begin(grp + 1); ++i) live[grp] |= (h_meta[i].state != prk::kSkip);
        if (live[grp]) rc_loop = enqueue_group(grp, 0);
    }
    for (uint32_t it = 0; it <= (uint32_t)crit.max_iteration && rc_loop == PR_OK && (live[0] || live[1] || live[2] || live[3]); ++it) {
        for (uint32_t grp = 0; grp < n_groups && rc_loop == PR_OK; ++grp) {
            if (!live[grp]) continue;
            if (hipError_t e = hipStreamSynchronize(group_stream(grp)); e != hipSuccess) { set_error("HIP error: %s", hipGetErrorString(e)); rc_loop = PR_ERR_HIP; break; }
            live[grp] = solve_group(grp, it) > 0;
            if (live[grp]) rc_loop = enqueue_group(grp, it + 1);                          // it + 1 <= max_iteration: the last iteration leaves nobody active
        }
    }
    for (uint32_t k = 1; k < n_groups; ++k) (void)hipStreamSynchronize(g->side[k - 1]);  // nothing of this call is left on a side stream, error or not
    if (rc_loop != PR_OK) { (void)hipStreamSynchronize(g->stream); drain_spans(); return rc_loop; }
    if (results_host) std::memcpy(results_host, res, sizeof(pr_result) * P);
    if (results_dev) {
        HIP_TRY(hipMemcpyAsync(results_dev, res, sizeof(pr_result) * P, hipMemcpyHostToDevice, g->stream));
        HIP_TRY(hipStreamSynchronize(g->stream));
    }
    drain_spans();
    return PR_OK;
}

// the raster packs pixel coordinates into 13 bits each and enumerates a triangle's candidate pixels with 24-bit arithmetic
bool frame_size_ok(size_t W, size_t H)
{
    if (W > 8192 || H > 8192 || W * H > ((size_t)1 << 24)) { set_error("frames larger than 8192 on a side or 2^24 pixels are not supported (got %zux%zu)", W, H); return false; }
    return true;
}

int render_impl(const pr_triangle *tris_dev, size_t n_tris, const pr_mat4 *poses_host, size_t P, size_t W, size_t H,
                const pr_mat4 *proj, pr_roi roi, int32_t *depth_dev, bool zero_empty)
{
    if ((!tris_dev && n_tris > 0) || (P && (!poses_host || !depth_dev)) || !proj || W == 0 || H == 0) { set_error("pr_render: bad arguments"); return PR_ERR_INVALID; }   // (an empty model has no array, no hypotheses need none)
    if (!frame_size_ok(W, H)) return PR_ERR_INVALID;
    size_t rw = W, rh = H;
    if (roi.width > 0 && roi.height > 0) {
        if (roi.x < 0 || roi.y < 0 || (size_t)(roi.x + roi.width) > W || (size_t)(roi.y + roi.height) > H) {
            set_error("pr_render: roi out of image");        // renderer.cu:202-203 asserts
            return PR_ERR_INVALID;
        }
        rw = (size_t)roi.width; rh = (size_t)roi.height;
    }
    if (P == 0) return PR_OK;
    PR_TRY(g->poses.ensure(sizeof(pr_mat4) * P));
    SpanGuard sp(kSpanRender);
    HIP_TRY(hipMemcpyAsync(g->poses.p, poses_host, sizeof(pr_mat4) * P, hipMemcpyHostToDevice, g->stream));
    HIP_TRY(prk::launch_fill_i32(depth_dev, P * rw * rh, INT32_MAX, g->stream));
    HIP_TRY(prk::launch_raster(tris_dev, (uint32_t)n_tris, g->poses.as<pr_mat4>(), (uint32_t)P, depth_dev, (uint32_t)W, (uint32_t)H,
                               *proj, roi, (uint32_t)rw, (uint32_t)rh, g->stream));
    if (zero_empty) HIP_TRY(prk::launch_max2zero(depth_dev, P * rw * rh, g->stream));
    return PR_OK;
}

template <typename T>
int depth2cloud_impl(const T *depth_dev, uint32_t W, uint32_t H, const float K[9], uint32_t stride, uint32_t tl_x, uint32_t tl_y,
                     pr_vec3 **cloud_out, uint32_t *n_out)
{
    if (!depth_dev || !K || !cloud_out || !n_out || W == 0 || H == 0 || stride == 0) { set_error("pr_depth2cloud: bad arguments"); return PR_ERR_INVALID; }
    const uint32_t gh = H / stride;
    PR_TRY(g->row_count.ensure(sizeof(uint32_t) * std::max(1u, gh)));
    PR_TRY(g->row_off.ensure(sizeof(uint32_t) * std::max(1u, gh)));
    PR_TRY(g->counts.ensure(sizeof(uint32_t)));
    HIP_TRY(prk::launch_depth2cloud<T>(depth_dev, 1, 0, W, H, stride, tl_x, tl_y, K[0], K[4], K[2], K[5], false, g->row_count.as<uint32_t>(),
                                       g->row_off.as<uint32_t>(), g->counts.as<uint32_t>(), nullptr, 0, false, g->stream));
    uint32_t n = 0;
    HIP_TRY(hipMemcpyAsync(&n, g->counts.p, sizeof n, hipMemcpyDeviceToHost, g->stream));
    HIP_TRY(hipStreamSynchronize(g->stream));
    pr_vec3 *cloud = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&cloud), sizeof(pr_vec3) * std::max(1u, n)));
    if (n > 0) {
        hipError_t e = prk::launch_depth2cloud<T>(depth_dev, 1, 0, W, H, stride, tl_x, tl_y, K[0], K[4], K[2], K[5], false, g->row_count.as<uint32_t>(),
                                                  g->row_off.as<uint32_t>(), g->counts.as<uint32_t>(), cloud, 0, true, g->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(g->stream);
        if (e != hipSuccess) { hipFree(cloud); set_error("depth2cloud emit failed: %s", hipGetErrorString(e)); return PR_ERR_HIP; }
    }
    *cloud_out = cloud; *n_out = n;
    return PR_OK;
}

bool roi_ok(pr_roi roi, uint32_t W, uint32_t H)
{
    if (roi.width <= 0 || roi.height <= 0) return true;            // no ROI
    if (roi.x < 0 || roi.y < 0 || (size_t)roi.x + (size_t)roi.width > W || (size_t)roi.y + (size_t)roi.height > H) {
        set_error("roi out of image");                               // renderer.cu:202-203 asserts
        return false;
    }
    return true;
}

int refine_impl(const pr_triangle *tris_dev, size_t n_tris, const pr_mat4 *poses_host, uint32_t P, uint32_t W, uint32_t H,
                const pr_mat4 *proj, const float K[9], int scene_kind, const void *scene, pr_criteria crit, pr_roi roi,
                pr_result *results_host, pr_result *results_dev, uint32_t *sizes_host)
{
    if (!K || W == 0 || H == 0) { set_error("pr_refine_batch: bad arguments"); return PR_ERR_INVALID; }
    if (!frame_size_ok(W, H) || !roi_ok(roi, W, H)) return PR_ERR_INVALID;
    if (P == 0) return PR_OK;
    SceneSel sc;
    std::memset(static_cast<void *>(&sc), 0, sizeof sc);           // also zeroes padding: sc is part of the graph-cache key
    const Camera cam{ W, H, K[0], K[4], K[2], K[5] };
    PR_TRY(make_scene(scene_kind, scene, /*want_packed=*/true, sc, nullptr, nullptr, &cam));
    // bound the depth workspace to ~4 GiB per chunk (288 GB of HBM would allow far more; this keeps
    // first-touch cost and the 2^32 element index space comfortable)
    const size_t img = (size_t)W * H;
    uint32_t chunk = (uint32_t)std::max<size_t>(1, std::min<size_t>(P, ((size_t)4 << 30) / (img * sizeof(int32_t))));
    std::vector<uint32_t> start(chunk), count(chunk);
    // model box: recomputed from the triangle buffer on every call (one pass over the mesh; nothing is cached by address here)
    PR_TRY(g->aabb.ensure(6 * sizeof(float)));
    PR_TRY(g->aabb_keys.ensure(6 * sizeof(uint32_t)));
    HIP_TRY(prk::launch_model_aabb(tris_dev, (uint32_t)n_tris, g->aabb_keys.as<uint32_t>(), g->aabb.as<float>(), nullptr, nullptr, g->stream));
    for (uint32_t p0 = 0; p0 < P; p0 += chunk) {
        const uint32_t np = std::min(chunk, P - p0);
        PR_TRY(g->depth.ensure(sizeof(int32_t) * img * np));
        PR_TRY(g->row_count.ensure(sizeof(uint32_t) * (size_t)H * np));
        PR_TRY(g->row_off.ensure(sizeof(uint32_t) * (size_t)H * np));
        PR_TRY(g->counts.ensure(sizeof(uint32_t) * np));
        PR_TRY(g->h_counts.ensure(sizeof(uint32_t) * np));
        uint32_t *h_counts = g->h_counts.as<uint32_t>();
        // per-pose pixel boxes; raster + row counts + row scan
        PR_TRY(g->bbox.ensure(sizeof(int4) * np));
        PR_TRY(g->poses.ensure(sizeof(pr_mat4) * np));
        {
            SpanGuard sp(kSpanRender);
            HIP_TRY(hipMemcpyAsync(g->poses.p, poses_host + p0, sizeof(pr_mat4) * np, hipMemcpyHostToDevice, g->stream));
            if (opt.raster_mode == 1)
                HIP_TRY(prk::launch_render_bands(tris_dev, (uint32_t)n_tris, g->poses.as<pr_mat4>(), np, g->aabb.as<float>(), g->bbox.as<int4>(),
                                                 g->depth.as<int32_t>(), g->row_count.as<uint32_t>(), g->row_off.as<uint32_t>(),
                                                 g->counts.as<uint32_t>(), W, H, *proj, roi, (uint32_t)g->n_cus, g->stream));
            else
                HIP_TRY(prk::launch_render_boxes(tris_dev, (uint32_t)n_tris, g->poses.as<pr_mat4>(), np, g->aabb.as<float>(), g->bbox.as<int4>(),
                                                 g->depth.as<int32_t>(), g->row_count.as<uint32_t>(), g->row_off.as<uint32_t>(),
                                                 g->counts.as<uint32_t>(), W, H, *proj, roi, g->stream));
        }
        HIP_TRY(hipMemcpyAsync(h_counts, g->counts.p, sizeof(uint32_t) * np, hipMemcpyDeviceToHost, g->stream));
        HIP_TRY(hipStreamSynchronize(g->stream));
        uint32_t max_n = 0;
        for (uint32_t i = 0; i < np; ++i) max_n = std::max(max_n, h_counts[i]);
        const size_t cstride = ((size_t)max_n + 3) & ~(size_t)3;           // keeps every cloud 16-byte aligned
        PR_TRY(g->cloud.ensure(sizeof(pr_vec3) * std::max<size_t>(4, cstride) * np));
        if (max_n > 0) {
            SpanGuard sp(kSpanCloud);
            HIP_TRY(prk::launch_emit_box(g->depth.as<int32_t>(), np, W, H, g->bbox.as<int4>(), K[0], K[4], K[2], K[5],
                                         g->row_count.as<uint32_t>(), g->row_off.as<uint32_t>(), g->cloud.as<pr_vec3>(), cstride, g->stream));
        }
        for (uint32_t i = 0; i < np; ++i) { start[i] = (uint32_t)(i * cstride); count[i] = h_counts[i]; }
        if (sizes_host) std::memcpy(sizes_host + p0, count.data(), sizeof(uint32_t) * np);
        PR_TRY(icp_drive(g->cloud.as<pr_vec3>(), start.data(), count.data(), np, sc, crit,
                         results_host ? results_host + p0 : nullptr, results_dev ? results_dev + p0 : nullptr));
    }
    return PR_OK;
}

// ---- asynchronous fused refinement: two slots, everything of a batch enqueued without a host round trip -------------------
// The synchronous path above reads the cloud sizes back in the middle of a step (they size the cloud stride and the grid),
// uploads the start state and returns only after the results are in: ≈150 µs of a 1.3 ms step during which the GPU idles.
// Here the host computes the per-pose pixel boxes itself (same arithmetic as pose_bbox_kernel, 8 corners per pose), which
// bounds every cloud by its box area, the start state is written by a kernel from the device-side counts, and a batch is
// only waited for when its results are wanted -- so the next batch can be enqueued while this one runs.
// The streams of BOTH slots are created together, in one run: main and first side stream of slot 0, then of slot 1.  Which
// hardware queue -- and with it which of the command processor's four pipes -- a stream lands on follows from the order in which
// the process created its queues (queue k -> pipe k mod 4), and two of these four streams on one pipe cost up to half the
// throughput (tools/queue_fairness.hip: two chains of dependent launches on one pipe take 2.1x a lone chain each, on two pipes
// 1.25x; bench.py with PR_STREAM_PADS-style gaps: 250 k -> 118-213 k poses/s).  Created lazily, slot by slot, anything the host does
// in between (its first synchronous copy creates the null stream's queue) shifts slot 1 onto slot 0's pipes; four queues
// created back to back always sit on four different pipes.  Further side streams (3-4 pose groups) come lazily, ensure_stream.
int slot_streams(Slot &sl)
{
    if (sl.stream) return PR_OK;
    for (Slot &o : g->slots) {
        if (o.stream || (!opt.eager_streams && &o != &sl)) continue;
        HIP_TRY(hipStreamCreateWithFlags(&o.stream, hipStreamNonBlocking));
        if (opt.eager_streams) PR_TRY(ensure_stream(o.side[0], &o.join[0]));
    }
    for (Slot &o : g->slots) {
        if (o.done) continue;
        HIP_TRY(hipEventCreateWithFlags(&o.scene_ready, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&o.fork, hipEventDisableTiming));
        // `done` is the one event a host thread waits on (pr_refine_wait): option "blocking_wait" makes that wait sleep instead of spin --
        // eight ranks spinning on a node whose cgroup grants sixteen CPUs leave nothing for RCCL's proxy threads
        HIP_TRY(hipEventCreateWithFlags(&o.done, hipEventDisableTiming | (opt.blocking_wait ? hipEventBlockingSync : 0u)));
        HIP_TRY(hipEventCreateWithFlags(&o.progress, hipEventDisableTiming));
    }
    return PR_OK;
}
// wait for everything a slot has in flight (its stream and side streams)
void slot_worker_wait(Slot &sl)      // the helper thread's batch, if one is running, has finished when this returns
{
    if (!sl.worker || !sl.worker_job) return;
    std::unique_lock<std::mutex> lk(sl.worker->mu);
    sl.worker->cv.wait(lk, [&] { return sl.worker->done || !sl.worker->alive; });
}
void slot_worker_stop(Slot &sl)
{
    if (!sl.worker) return;
    { std::lock_guard<std::mutex> lk(sl.worker->mu); sl.worker->quit = true; }
    sl.worker->cv.notify_all();
    if (sl.worker->th.joinable()) sl.worker->th.join();
    sl.worker.reset();
    sl.worker_job = false;
}
void slot_drain(Slot &sl)
{
    slot_worker_wait(sl);
    for (int i = 0; i < 3; ++i) if (sl.side[i]) (void)hipStreamSynchronize(sl.side[i]);
    if (sl.stream) (void)hipStreamSynchronize(sl.stream);
}
void drain_all_slots()
{
    for (Slot &o : g->slots) if (o.pending && !o.delivered) slot_drain(o);
}
void slot_release(Slot &sl)
{
    slot_drain(sl);
    slot_worker_stop(sl);
    for (DevBuf *b : { &sl.poses_bbox, &sl.depth, &sl.row_count, &sl.row_off, &sl.counts, &sl.cloud, &sl.meta, &sl.partial, &sl.dstate,
                       &sl.dresults, &sl.arrive, &sl.aabb_keys, &sl.nn_prev, &sl.packed.rec }) b->release();
    sl.packed = PackedCache();
    sl.h_in.release(); sl.h_out.release();
    for (int i = 0; i < 3; ++i) {
        if (sl.side[i]) hipStreamDestroy(sl.side[i]);
        if (sl.join[i]) hipEventDestroy(sl.join[i]);
        sl.side[i] = nullptr; sl.join[i] = nullptr;
    }
    if (sl.fork) hipEventDestroy(sl.fork);
    if (sl.done) hipEventDestroy(sl.done);
    if (sl.scene_ready) hipEventDestroy(sl.scene_ready);
    if (sl.progress) hipEventDestroy(sl.progress);
    sl.progress = nullptr; sl.progress_valid = false;
    if (sl.stream) hipStreamDestroy(sl.stream);
    sl.fork = sl.done = sl.scene_ready = nullptr; sl.stream = nullptr; sl.pending = false;
    for (hipEvent_t e : sl.t_events) (void)hipEventDestroy(e);
    sl.t_events.clear(); sl.t_used = 0; sl.t_spans.clear(); sl.timed = false;
}

// Host-side copy of a triangle buffer's box, once per (pointer, size): it feeds the per-pose pixel boxes computed on the host,
// which size a batch before anything of it has run.  The copy is an ASSUMPTION about memory the caller owns: every batch that
// uses it re-derives the box on the device (one pass over the mesh, on a stream that is idle at that point) and refine_wait
// re-runs the batch through the synchronous path if the two differ -- a rewritten mesh costs one repeated batch, never a wrong one.
// (An indexed form of the soup with a per-pose vertex stage was built and measured as well: three divergent 16-byte gathers
// per triangle cost the texture addresser more than the 170 saved VALU instructions give back -- 1.27 -> 1.32 ms per step.)
int ensure_model_box(const pr_triangle *tris_dev, size_t n_tris)
{
    if (g->mesh_key == tris_dev && g->mesh_n == n_tris && g->aabb_host_valid) return PR_OK;
    std::vector<pr_triangle> h(n_tris);
    if (n_tris) HIP_TRY(hipMemcpy(h.data(), tris_dev, sizeof(pr_triangle) * n_tris, hipMemcpyDeviceToHost));
    const float *f = reinterpret_cast<const float *>(h.data());
    float lo[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, hi[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
    for (size_t v = 0; v < n_tris * 3; ++v)
        for (int d = 0; d < 3; ++d) { lo[d] = fminf(lo[d], f[3 * v + d]); hi[d] = fmaxf(hi[d], f[3 * v + d]); }
    for (int d = 0; d < 3; ++d) { g->aabb_host[d] = lo[d]; g->aabb_host[3 + d] = hi[d]; }
    g->mesh_key = tris_dev; g->mesh_n = n_tris; g->aabb_host_valid = true;
    return PR_OK;
}

// pose_bbox_kernel on the host (same operations in the same order; any conservative box gives the same images and clouds)
void pose_bbox_host(const float *aabb, const pr_mat4 &pose, const pr_mat4 &proj, uint32_t width, uint32_t height, pr_roi roi, int32_t out[4])
{
    const float *M = pose.m;
    float mnx = FLT_MAX, mny = FLT_MAX, mxx = -FLT_MAX, mxy = -FLT_MAX;
    bool all_front = true;
    for (int c = 0; c < 8; ++c) {
        const float x = aabb[(c & 1) ? 3 : 0], y = aabb[(c & 2) ? 4 : 1], z = aabb[(c & 4) ? 5 : 2];
        const float lx = M[0] * x + M[1] * y + M[2] * z + M[3];
        const float ly = M[4] * x + M[5] * y + M[6] * z + M[7];
        const float lz = M[8] * x + M[9] * y + M[10] * z + M[11];
        if (!(lz > 1e-3f)) all_front = false;
        const float cxp = proj.m[0] * lx + proj.m[1] * ly + proj.m[2] * lz + proj.m[3];
        const float cyp = proj.m[4] * lx + proj.m[5] * ly + proj.m[6] * lz + proj.m[7];
        const float sx = cxp / lz * (float)width / 2.0f + (float)width / 2.0f;
        const float sy = cyp / lz * (float)height / 2.0f + (float)height / 2.0f;
        mnx = fminf(mnx, sx); mxx = fmaxf(mxx, sx); mny = fminf(mny, sy); mxy = fmaxf(mxy, sy);
    }
    int x0 = 0, y0 = 0, x1 = (int)width - 1, y1 = (int)height - 1;
    const bool finite = (mnx > -1e8f) && (mxx < 1e8f) && (mny > -1e8f) && (mxy < 1e8f);
    if (all_front && finite) {
        x0 = std::max(0, (int)floorf(mnx) - 2);  x1 = std::min((int)width - 1, (int)ceilf(mxx) + 2);
        y0 = std::max(0, (int)floorf(mny) - 2);  y1 = std::min((int)height - 1, (int)ceilf(mxy) + 2);
    }
    if (roi.width > 0 && roi.height > 0) {
        x0 = std::max(x0, roi.x);  x1 = std::min(x1, roi.x + roi.width - 1);
        y0 = std::max(y0, (int)height - 1 - (roi.y + roi.height - 1));  y1 = std::min(y1, (int)height - 1 - roi.y);
    }
    out[0] = x0; out[1] = y0; out[2] = x1; out[3] = y1;
}

// the synchronous entry points run on whichever slot holds no unfinished batch of the caller's (-1: none)
int free_slot() { for (int i = 0; i < kSlots; ++i) if (!g->slots[i].pending) return i; return -1; }

int refine_wait(int slot)
{
    if (slot < 0 || slot >= kSlots) { set_error("slot must be 0..%d", kSlots - 1); return PR_ERR_INVALID; }
    Slot &sl = g->slots[slot];
    if (!sl.pending) { set_error("pr_refine_wait: nothing was submitted on slot %d", slot); return PR_ERR_INVALID; }
    if (sl.delivered) { sl.pending = false; return PR_OK; }
    if (sl.worker_job) {                                          // the batch ran on the slot's helper thread: collect its verdict
        slot_worker_wait(sl);
        sl.worker_job = false; sl.pending = false;
        std::lock_guard<std::mutex> lk(sl.worker->mu);
        if (!sl.worker->done) { set_error("pr_refine_wait: the slot's helper thread ended before its batch did"); return PR_ERR_HIP; }
        if (sl.worker->rc != PR_OK) set_error("%s", sl.worker->err.c_str());
        return sl.worker->rc;
    }
    {
        // the slot stays occupied until its batch has really finished: if the wait itself fails, everything the slot has in flight is
        // drained before the error goes back (a later submit must never reuse the slot's buffers under a running batch)
        const hipError_t we = hipEventSynchronize(sl.done);
        if (we != hipSuccess) { slot_drain(sl); sl.pending = false; set_error("pr_refine_wait: hipEventSynchronize failed: %s", hipGetErrorString(we)); return PR_ERR_HIP; }
    }
    sl.pending = false;
    const unsigned char *h_out = sl.h_out.as<unsigned char>();
    if (*reinterpret_cast<const volatile uint32_t *>(h_out + sl.flag_off) != 0u) {
        // the triangle buffer no longer has the box this batch was sized with: forget the host copy and run the batch again,
        // synchronously (that path derives every box on the device); outputs are overwritten in full
        g->aabb_host_valid = false; g->mesh_key = nullptr;
        g->stat_repeated++;                                         // (pr_stats: the safety net is not free -- a caller that sees this count grow should call pr_invalidate)
        // (or a scene array no longer has the content its cached form was derived from: same cure)
        drain_all_slots();
        for (Slot &o : g->slots) o.packed.valid = false;
        g->packed.valid = false; g->nn_cache.valid = false; g->nn_cache.grid_valid = false;
        const Resubmit &r = sl.again;
        const void *scene = (r.scene_kind == PR_SCENE_NN) ? static_cast<const void *>(&r.sn) : static_cast<const void *>(&r.sp);
        return refine_impl(r.tris, r.n_tris, sl.h_in.as<pr_mat4>(), sl.P, r.W, r.H, &r.proj, r.K, r.scene_kind, scene, r.crit, r.roi,
                           sl.user_results_host, r.results_dev, sl.user_sizes);
    }
    const uint32_t *h_counts = sl.h_out.as<uint32_t>();
    uint32_t largest = 1;
    for (uint32_t i = 0; i < sl.P; ++i) largest = std::max(largest, h_counts[i]);
    g->cloud_hint = largest;
    if (sl.timed) {                                              // the batch carried timing events: the same accounts as the synchronous timed path keeps
        for (const Slot::TSpan &t : sl.t_spans) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, sl.t_events[t.e0], sl.t_events[t.e1]) != hipSuccess) continue;
            if (t.kind == kSpanRender) g->render_ms += ms;
            else if (t.kind == kSpanCloud) g->cloud_ms += ms;
            else {
                uint64_t pts = 0;
                for (uint32_t i = t.q0; i < t.q0 + t.nq; ++i) pts += h_counts[i];
                g->icp_ms += ms; g->icp_launches++; note_launch_us(ms);
                g->icp_points += pts; g->icp_bytes += pts * (t.edge ? 36u : 48u);
                if (t.marks) {                                      // kd-tree pass: search | bound | task walk | winners pass
                    const hipEvent_t ev[5] = { sl.t_events[t.e0], sl.t_events[t.m[0]], sl.t_events[t.m[1]], sl.t_events[t.m[2]], sl.t_events[t.e1] };
                    float part[4]; bool ok = true;
                    for (int k = 0; k < 4; ++k) ok = ok && hipEventElapsedTime(&part[k], ev[k], ev[k + 1]) == hipSuccess;
                    if (ok) { for (int k = 0; k < 4; ++k) g->nn_part_ms[k] += part[k]; g->nn_part_n++; }
                }
            }
        }
        sl.t_spans.clear(); sl.t_used = 0; sl.timed = false;
    }
    if (sl.user_sizes) std::memcpy(sl.user_sizes, h_counts, sizeof(uint32_t) * sl.P);
    if (sl.user_results_host) std::memcpy(sl.user_results_host, h_out + (((size_t)sl.P * 4 + 63) & ~(size_t)63), sizeof(pr_result) * sl.P);
    return PR_OK;
}

int refine_impl(const pr_triangle *tris_dev, size_t n_tris, const pr_mat4 *poses_host, uint32_t P, uint32_t W, uint32_t H,
                const pr_mat4 *proj, const float K[9], int scene_kind, const void *scene, pr_criteria crit, pr_roi roi,
                pr_result *results_host, pr_result *results_dev, uint32_t *sizes_host);
// the helper thread of a slot: a private context on the slot's device, then one synchronous batch per job
void slot_worker_main(SlotWorker *w)
{
    bool ready = bind_shared(w->device) == PR_OK && pr_thread_context(1) == PR_OK;
    for (;;) {
        std::unique_lock<std::mutex> lk(w->mu);
        w->cv.wait(lk, [&] { return w->has_job || w->quit; });
        if (w->quit) break;
        lk.unlock();
        int rc = PR_ERR_HIP;
        if (!ready) set_error("the slot's helper thread could not create its context: %s", std::string(prh::g_err).c_str());
        else {
            std::lock_guard<std::mutex> ck(g->mu);
            rc = require_ctx();
            if (rc == PR_OK) {
                const Resubmit &r = w->in;
                const void *scene = (r.scene_kind == PR_SCENE_NN) ? static_cast<const void *>(&r.sn)
                                    : (r.scene_kind == PR_SCENE_PROJ_CROP ? static_cast<const void *>(&r.sp) : static_cast<const void *>(&r.sp.view));
                rc = refine_impl(r.tris, r.n_tris, w->poses.data(), (uint32_t)w->poses.size(), r.W, r.H, &r.proj, r.K, r.scene_kind, scene, r.crit, r.roi,
                                 w->results_host, r.results_dev, w->sizes_host);
            }
        }
        lk.lock();
        w->rc = rc; w->err = (rc == PR_OK) ? std::string() : prh::g_err;
        w->has_job = false; w->done = true;
        lk.unlock();
        w->cv.notify_all();
    }
    if (ready) (void)pr_thread_context(0);
    { std::lock_guard<std::mutex> lk(w->mu); w->alive = false; }
    w->cv.notify_all();
}
int slot_worker_post(Slot &sl, const pr_triangle *tris_dev, size_t n_tris, const pr_mat4 *poses_host, uint32_t P, uint32_t W, uint32_t H,
                     const pr_mat4 *proj, const float K[9], int scene_kind, const void *scene, pr_criteria crit, pr_roi roi,
                     pr_result *results_host, pr_result *results_dev, uint32_t *sizes_host)
{
    if (scene_kind != PR_SCENE_NN && scene_kind != PR_SCENE_PROJ && scene_kind != PR_SCENE_PROJ_CROP) { set_error("unknown scene kind %d", scene_kind); return PR_ERR_INVALID; }
    if (!scene) { set_error("pr_refine_submit: null scene"); return PR_ERR_INVALID; }
    if (!sl.worker) {
        sl.worker.reset(new SlotWorker());
        sl.worker->device = g->device; sl.worker->alive = true;
        try { sl.worker->th = std::thread(slot_worker_main, sl.worker.get()); }
        catch (...) { sl.worker.reset(); set_error("pr_refine_submit: cannot start the slot's helper thread"); return PR_ERR_NOMEM; }
    }
    SlotWorker &w = *sl.worker;
    {
        std::lock_guard<std::mutex> lk(w.mu);
        Resubmit &r = w.in;
        r.tris = tris_dev; r.n_tris = n_tris; r.W = W; r.H = H; r.proj = *proj; std::memcpy(r.K, K, sizeof r.K);
        r.scene_kind = scene_kind; r.crit = crit; r.roi = roi; r.results_dev = results_dev;
        if (scene_kind == PR_SCENE_NN) r.sn = *static_cast<const pr_scene_nn *>(scene);
        else if (scene_kind == PR_SCENE_PROJ_CROP) r.sp = *static_cast<const pr_scene_proj_crop *>(scene);
        else { r.sp.view = *static_cast<const pr_scene_proj *>(scene); r.sp.tl_x = r.sp.tl_y = 0; }
        w.poses.assign(poses_host, poses_host + P);
        w.results_host = results_host; w.sizes_host = sizes_host;
        w.done = false; w.has_job = true;
    }
    w.cv.notify_all();
    return PR_OK;
}

int refine_submit_async(Slot &sl, const pr_triangle *tris_dev, size_t n_tris, uint32_t P, uint32_t W, uint32_t H,
                        const pr_mat4 *proj, const float K[9], int scene_kind, const void *scene, pr_criteria crit, pr_roi roi,
                        pr_result *results_host, pr_result *results_dev);

int refine_submit(int slot, const pr_triangle *tris_dev, size_t n_tris, const pr_mat4 *poses_host, uint32_t P, uint32_t W, uint32_t H,
                  const pr_mat4 *proj, const float K[9], int scene_kind, const void *scene, pr_criteria crit, pr_roi roi,
                  pr_result *results_host, pr_result *results_dev, uint32_t *sizes_host)
{
    if (slot < 0 || slot >= kSlots) { set_error("slot must be 0..%d", kSlots - 1); return PR_ERR_INVALID; }
    Slot &sl = g->slots[slot];
    if (sl.pending) { set_error("pr_refine_submit: slot %d still holds an unfinished batch (call pr_refine_wait)", slot); return PR_ERR_INVALID; }
    if ((!tris_dev && n_tris > 0) || !poses_host || !proj || !K || W == 0 || H == 0 || (!results_host && !results_dev)) { set_error("pr_refine_submit: bad arguments"); return PR_ERR_INVALID; }
    if (crit.max_iteration < 0) { set_error("max_iteration must be >= 0"); return PR_ERR_INVALID; }
    if (!frame_size_ok(W, H) || !roi_ok(roi, W, H)) return PR_ERR_INVALID;
    sl.P = P; sl.user_results_host = results_host; sl.user_sizes = sizes_host; sl.delivered = false;
    const size_t img = (size_t)W * H;
    const uint64_t period = (uint64_t)std::max(1, opt.sample_period);
    const bool sample_call = (opt.profile == 2) && (g->sample_clock % period == 0);
    const bool proj_scene = (scene_kind == PR_SCENE_PROJ || scene_kind == PR_SCENE_PROJ_CROP);
    const bool nn_scene = (scene_kind == PR_SCENE_NN) && !opt.nn_count;     // (an instrumented kd-tree run stays synchronous)
    (void)img;                                                   // (large frames: the asynchronous path sizes its sub-batches to its workspace bound)
    const bool async_ok = P > 0 && opt.solve_mode == PR_SOLVE_DEVICE && !opt.icp_flow && opt.raster_mode == 0 && (proj_scene || nn_scene)
                          && (opt.profile == 0 || opt.profile == 3 || (opt.profile == 2 && !sample_call));
    if (!async_ok && P > 0 && opt.solve_mode == PR_SOLVE_HOST && opt.host_worker && opt.profile == 0 && !opt.icp_flow && !opt.nn_count) {
        // host solve, nothing to time: the batch goes to the slot's helper thread (see SlotWorker) and this call returns
        PR_TRY(slot_worker_post(sl, tris_dev, n_tris, poses_host, P, W, H, proj, K, scene_kind, scene, crit, roi, results_host, results_dev, sizes_host));
        sl.pending = true; sl.delivered = false; sl.worker_job = true;
        return PR_OK;
    }
    if (!async_ok) {
        // the synchronous path (host solve, timed calls, oversized batches): let the other slot drain first so
        // that a timed launch has the chip to itself, then run to completion; pr_refine_wait has nothing left to do
        for (Slot &o : g->slots) if (o.pending && !o.delivered && o.done) HIP_TRY(hipEventSynchronize(o.done));
        PR_TRY(refine_impl(tris_dev, n_tris, poses_host, P, W, H, proj, K, scene_kind, scene, crit, roi, results_host, results_dev, sizes_host));
        sl.pending = true; sl.delivered = true;
        return PR_OK;
    }
    g->sample_clock++;
    // the host copies of this batch's inputs (the poses are staged from here; the rest is what a re-run needs)
    const size_t in_bytes = (sizeof(pr_mat4) + sizeof(int4)) * (size_t)P;
    PR_TRY(sl.h_in.ensure(in_bytes + 16));
    std::memcpy(sl.h_in.p, poses_host, sizeof(pr_mat4) * P);
    Resubmit &r = sl.again;
    r.tris = tris_dev; r.n_tris = n_tris; r.W = W; r.H = H; r.proj = *proj; std::memcpy(r.K, K, sizeof r.K);
    r.scene_kind = scene_kind; r.crit = crit; r.roi = roi; r.results_dev = results_dev;
    if (scene_kind == PR_SCENE_NN) r.sn = *static_cast<const pr_scene_nn *>(scene);
    else if (scene_kind == PR_SCENE_PROJ_CROP) r.sp = *static_cast<const pr_scene_proj_crop *>(scene);
    else { r.sp.view = *static_cast<const pr_scene_proj *>(scene); r.sp.tl_x = r.sp.tl_y = 0; }
    const int rc = refine_submit_async(sl, tris_dev, n_tris, P, W, H, proj, K, scene_kind, scene, crit, roi, results_host, results_dev);
    if (rc != PR_OK) {                                            // part of the batch may already be queued: do not leave it running
        slot_drain(sl);                                           // behind the caller's back (its buffers may go away next)
        sl.pending = false;
        return rc;
    }
    sl.pending = true;
    return PR_OK;
}

int refine_submit_async(Slot &sl, const pr_triangle *tris_dev, size_t n_tris, uint32_t P, uint32_t W, uint32_t H,
                        const pr_mat4 *proj, const float K[9], int scene_kind, const void *scene, pr_criteria crit, pr_roi roi,
                        pr_result *results_host, pr_result *results_dev)
{
    const size_t img = (size_t)W * H;
    PR_TRY(slot_streams(sl));
    SceneSel sc;
    std::memset(static_cast<void *>(&sc), 0, sizeof sc);
    // the packed copy of the scene does not depend on the render: when it has to be (re)built that happens on the first side
    // stream (idle until the loop forks), the loop waits for it; batches without pose groups build it in line
    const uint32_t groups_hint = std::max(1u, std::min({ pose_groups_for(scene_kind), 4u, std::min<uint32_t>(P, (uint32_t)std::max(32, opt.sub_batch)) / 32u }));
    for (uint32_t k = 1; k < groups_hint; ++k) PR_TRY(ensure_stream(sl.side[k - 1], &sl.join[k - 1]));
    hipStream_t scene_stream = groups_hint > 1 ? sl.side[0] : sl.stream;
    const Camera cam{ W, H, K[0], K[4], K[2], K[5] };            // kd-tree scenes: the pixel grid of the scene points under this camera
    // nothing of this batch may run beside the loop of a TIMED batch on the other slot -- not even the small checks on the scene stream
    if (scene_stream != sl.stream)
        for (Slot &o : g->slots) if (&o != &sl && o.pending && !o.delivered && o.timed && o.progress_valid) HIP_TRY(hipStreamWaitEvent(scene_stream, o.progress, 0));
    PR_TRY(make_scene(scene_kind, scene, /*want_packed=*/true, sc, &sl.packed, scene_stream, scene_kind == PR_SCENE_NN ? &cam : nullptr, /*verify_now=*/false));

    PR_TRY(ensure_model_box(tris_dev, n_tris));                 // once per triangle buffer ...
    const size_t res_off = ((size_t)P * 4 + 63) & ~(size_t)63;
    sl.flag_off = (res_off + sizeof(pr_result) * P + 63) & ~(size_t)63;
    PR_TRY(sl.h_out.ensure(sl.flag_off + 64));
    void *h_in_dev = nullptr, *h_out_dev = nullptr;              // the pinned staging buffers as the device sees them
    HIP_TRY(hipHostGetDevicePointer(&h_in_dev, sl.h_in.p, 0));
    HIP_TRY(hipHostGetDevicePointer(&h_out_dev, sl.h_out.p, 0));
    // ... and checked against the buffer's present content by every batch (refine_wait acts on the flag)
    PR_TRY(sl.aabb_keys.ensure(6 * sizeof(uint32_t)));
    *reinterpret_cast<volatile uint32_t *>(sl.h_out.as<unsigned char>() + sl.flag_off) = 0u;
    HIP_TRY(prk::launch_model_aabb(tris_dev, (uint32_t)n_tris, sl.aabb_keys.as<uint32_t>(), nullptr, g->aabb_host,
                                   reinterpret_cast<uint32_t *>(static_cast<unsigned char *>(h_out_dev) + sl.flag_off), scene_stream));
    if (opt.scene_cache) {
        // ... and so are the scene caches: a sampled fingerprint of the caller's arrays against the one taken when the cache was built
        // (after the box check on the same stream: that one writes the flag either way, this one only ever raises it)
        uint32_t *flag = reinterpret_cast<uint32_t *>(static_cast<unsigned char *>(h_out_dev) + sl.flag_off);
        if (scene_kind == PR_SCENE_NN) {
            const pr_scene_nn *sn = static_cast<const pr_scene_nn *>(scene);
            HIP_TRY(prk::launch_scene_fingerprint(sn->pcd, (size_t)sn->n_points * sizeof(pr_vec3), sn->nodes, (size_t)sn->n_nodes * sizeof(pr_kdnode), sn->normal,
                                                  (size_t)sn->n_points * sizeof(pr_vec3), g->nndepth.as<uint32_t>() + 12, flag, true, scene_stream));
        } else if (sl.packed.valid) {
            const pr_scene_proj *sp = static_cast<const pr_scene_proj *>(scene);
            const size_t n = (size_t)sp->width * sp->height;
            const size_t tables = ((sp->width + sp->height) * sizeof(float) + 15) & ~(size_t)15;
            uint32_t *exact_dev = reinterpret_cast<uint32_t *>(reinterpret_cast<unsigned char *>(sl.packed.rec.as<float4>() + n) + tables);
            HIP_TRY(prk::launch_scene_fingerprint(sp->pcd, n * sizeof(pr_vec3), sp->normal, n * sizeof(pr_vec3), nullptr, 0, exact_dev + 1, flag, true, scene_stream));
        }
    }
    HIP_TRY(hipEventRecord(sl.scene_ready, scene_stream));

    // staging: [poses][boxes]; the cloud stride and the grid come from the largest box
    const size_t in_bytes = (sizeof(pr_mat4) + sizeof(int4)) * (size_t)P;
    PR_TRY(sl.poses_bbox.ensure(in_bytes + 16));
    pr_mat4 *h_poses = sl.h_in.as<pr_mat4>();
    int32_t *h_box = reinterpret_cast<int32_t *>(h_poses + P);
    size_t max_area = 1;
    for (uint32_t i = 0; i < P; ++i) {
        pose_bbox_host(g->aabb_host, h_poses[i], *proj, W, H, roi, h_box + 4 * (size_t)i);
        const int32_t *b = h_box + 4 * (size_t)i;
        max_area = std::max(max_area, (size_t)std::max(0, b[2] - b[0] + 1) * (size_t)std::max(0, b[3] - b[1] + 1));   // an off-screen pose has an empty box
    }
    const size_t cstride = (max_area + 3) & ~(size_t)3;
    const uint32_t steps = (uint32_t)std::max(1, opt.steps);
    const uint32_t ppb = steps * prk::kPointsPerStep;
    const uint32_t nblk = (uint32_t)((max_area + ppb - 1) / ppb);       // bound from the pixel boxes: capacity of the partial sums
    // grid: one workgroup per block of the largest cloud of the previous batch (workgroups loop if this batch's clouds are
    // larger, surplus workgroups exit at once); the box bound itself would launch ~60 % empty workgroups (-2.5 % poses/s)
    const uint32_t grid_x = g->cloud_hint ? std::min(nblk, (g->cloud_hint + ppb - 1) / ppb) : nblk;

    // Large batches run as consecutive sub-batches that reuse the same depth / cloud / partial-sum memory: the clouds of
    // <= 512 hypotheses (~140 MB touched) stay in the 256 MiB Infinity Cache over their 21 passes (1024 poses as one batch:
    // 199 k poses/s, as 2 x 512: see DESIGN.md)
    // ... and, for large frames, small enough that the depth workspace of a sub-batch stays within ~4 GiB (as the synchronous path
    // bounds its chunks) and the clouds within 2^30 points (offsets are 32-bit): 64 hypotheses per sub-batch at 4096 x 4096, all 512 up to 2 M pixels
    uint32_t sub_cap = (uint32_t)std::max(32, opt.sub_batch);
    sub_cap = (uint32_t)std::max<size_t>(1, std::min<size_t>({ (size_t)sub_cap, ((size_t)4 << 30) / (img * sizeof(int32_t)), ((size_t)1 << 30) / cstride }));     // (2^30 cloud points per sub-batch: 12 GiB of clouds, 24 GiB of kd-tree search state at most)
    const uint32_t n_sub = (P + sub_cap - 1) / sub_cap;
    const uint32_t sub = (P + n_sub - 1) / n_sub;
    PR_TRY(sl.depth.ensure(sizeof(int32_t) * img * sub));
    PR_TRY(sl.row_count.ensure(sizeof(uint32_t) * (size_t)H * sub));
    PR_TRY(sl.row_off.ensure(sizeof(uint32_t) * (size_t)H * sub));
    PR_TRY(sl.counts.ensure(sizeof(uint32_t) * P));
    PR_TRY(sl.cloud.ensure(sizeof(pr_vec3) * cstride * sub));
    PR_TRY(sl.meta.ensure(sizeof(prk::PoseMeta) * P));
    PR_TRY(sl.partial.ensure(sizeof(float) * prk::kAccStride * (size_t)nblk * sub));
    PR_TRY(sl.dstate.ensure(sizeof(prk::DevIcpState) * P));
    PR_TRY(sl.arrive.ensure(sizeof(uint32_t) * P));
    pr_result *dres = results_dev;
    if (!dres) { PR_TRY(sl.dresults.ensure(sizeof(pr_result) * P)); dres = sl.dresults.as<pr_result>(); }
    // kd-tree scene: winners | slack | queue | two queue counters per hypothesis, indexed like the clouds of one sub-batch (icp_drive
    // has the same layout); the search kernel's grid comes from the box bound, surplus workgroups exit at once
    uint32_t *nn_prev = nullptr;
    const size_t nn_span = cstride * (size_t)sub;
    if (sc.kind == PR_SCENE_NN) {
        sc.nn_split = (sc.nn.rec32 && opt.nn_split) ? 1u : 0u;
        sc.nn_max_points = (uint32_t)std::min<size_t>(max_area, 0xffffffffu);
        if (sc.nn.rec32 && (opt.nn_seed || sc.nn_split)) {
            PR_TRY(sl.nn_prev.ensure(sizeof(uint32_t) * (nn_span * prk::kNNWordsPerPoint + prk::kQCountStride * (size_t)sub) + 64));
            nn_prev = sl.nn_prev.as<uint32_t>();
        }
    }

    hipStream_t st = sl.stream;
    // A TIMED batch (profile 3) stays asynchronous: its render runs under the other slot's loop like any other, but its own loop waits
    // until the other slot's batch is complete, runs as ONE pose group, lets the other slot's next render start only when it has
    // finished, and carries HIP events around every launch -- the timed launches have the chip to themselves as in the synchronous timed
    // path (profile 1), without that path's host round trips and its idle render.
    const bool timed = (opt.profile == 3);
    sl.timed = timed; sl.t_used = 0; sl.t_spans.clear();
    // A batch submitted while the other slot is idle starts a pipeline.  Option start_overlap says when the NEXT batch's render may start
    // in that case: -1 (default) = by the rule below, like any other batch; a pass number releases it earlier (see the option's note).
    bool pipeline_start = true;
    for (Slot &o : g->slots) if (&o != &sl && o.pending && !o.delivered) pipeline_start = false;
    // a failed event creation or record drops the TIMING of this batch (never the batch): t_fail, checked when everything is enqueued
    bool t_fail = false;
    auto t_event = [&]() -> size_t {
        if (sl.t_used == sl.t_events.size()) {
            hipEvent_t e = nullptr;
            if (hipEventCreate(&e) != hipSuccess || !e) { (void)hipGetLastError(); t_fail = true; return kNoEvent; }
            sl.t_events.push_back(e);
        }
        return sl.t_used++;
    };
    auto t_record = [&](size_t e) { if (e == kNoEvent) { t_fail = true; return; } if (hipEventRecord(sl.t_events[e], st) != hipSuccess) { (void)hipGetLastError(); t_fail = true; } };
    auto t_begin = [&]() -> size_t { const size_t e = t_event(); t_record(e); return e; };
    auto t_end = [&](size_t e0, int kind, uint32_t q0, uint32_t nq, bool edge) { const size_t e1 = t_event(); t_record(e1); if (!t_fail) sl.t_spans.push_back({ e0, e1, kind, q0, nq, edge, false, { 0, 0, 0 } }); };
    pr_mat4 *d_poses = sl.poses_bbox.as<pr_mat4>();
    int4 *d_box = reinterpret_cast<int4 *>(d_poses + P);
    // Both phases are bound by the same units, so a batch that renders while the other slot is in the middle of its ICP loop
    // slows that loop by more than it gains; its render is therefore held back until the other slot has issued pass
    // `overlap_pass` of its (last sub-batch's) loop -- late enough to disturb little, early enough that the GPU never idles.
    for (Slot &o : g->slots)
        if (&o != &sl && o.pending && !o.delivered && o.progress_valid) HIP_TRY(hipStreamWaitEvent(st, o.progress, 0));
    sl.progress_valid = false;
    HIP_TRY(prk::launch_stage_words(h_in_dev, d_poses, in_bytes, st));
    const bool fused = opt.fused_solve != 0;
    const pr_roi none{ 0, 0, 0, 0 };                              // the ROI is already part of the host-computed boxes
    for (uint32_t q0 = 0; q0 < P; q0 += sub) {
        const uint32_t nq = std::min(sub, P - q0);
        prk::PoseMeta *meta = sl.meta.as<prk::PoseMeta>() + q0;
        prk::DevIcpState *dstate = sl.dstate.as<prk::DevIcpState>() + q0;
        uint32_t *arrive = sl.arrive.as<uint32_t>() + q0;
        size_t te = timed ? t_begin() : 0;
        HIP_TRY(prk::launch_render_boxes(tris_dev, (uint32_t)n_tris, d_poses + q0, nq, nullptr, d_box + q0, sl.depth.as<int32_t>(),
                                         sl.row_count.as<uint32_t>(), sl.row_off.as<uint32_t>(), sl.counts.as<uint32_t>() + q0, W, H, *proj, none, st,
                                         /*compute_boxes=*/false, meta, dstate, arrive, (uint32_t)cstride));
        if (timed) { t_end(te, kSpanRender, q0, nq, false); te = t_begin(); }
        HIP_TRY(prk::launch_emit_box(sl.depth.as<int32_t>(), nq, W, H, d_box + q0, K[0], K[4], K[2], K[5], sl.row_count.as<uint32_t>(),
                                     sl.row_off.as<uint32_t>(), sl.cloud.as<pr_vec3>(), cstride, st));
        if (timed) t_end(te, kSpanCloud, q0, nq, false);
        if (q0 == 0) HIP_TRY(hipStreamWaitEvent(st, sl.scene_ready, 0));
        if (timed && q0 == 0)                                     // the timed loop starts when the other slot's batch is complete
            for (Slot &o : g->slots) if (&o != &sl && o.pending && !o.delivered && o.done) HIP_TRY(hipStreamWaitEvent(st, o.done, 0));

        // the iteration loop: (max_iteration+1) x [pass (+ fused finalize/solve)], pose groups on the slot's side streams
        const uint32_t n_groups = timed ? 1u : std::max(1u, std::min({ pose_groups_for(scene_kind), 4u, nq / 32u }));
        auto group_begin = [&](uint32_t grp) { return (uint32_t)(((uint64_t)nq * grp) / n_groups); };
        if (nn_prev) HIP_TRY(hipMemsetAsync(nn_prev + 6 * nn_span, 0, sizeof(uint32_t) * prk::kQCountStride * nq, st));
        if (n_groups > 1) {
            for (uint32_t k = 1; k < n_groups; ++k) PR_TRY(ensure_stream(sl.side[k - 1], &sl.join[k - 1]));
            HIP_TRY(hipEventRecord(sl.fork, st));
            for (uint32_t k = 1; k < n_groups; ++k) HIP_TRY(hipStreamWaitEvent(sl.side[k - 1], sl.fork, 0));
        }
        // When may the OTHER slot start rendering?  Measured optimum (21 passes of obj_06, overlap_pass swept per batch size):
        // pass 4 or earlier at 128 hypotheses, 9-10 at 256, 14-16 at 384, 16 at 512 -- i.e. when about 2800 hypothesis-passes of
        // this loop are left (never fewer than 5 passes): that much loop work is what a render hides behind without stretching the
        // passes it runs beside.  The render's own weight scales that figure: triangles per hypothesis against cloud points per
        // hypothesis (31 468 and 27 400 there; cloud_hint = the largest cloud of the previous batch).
        uint32_t auto_overlap = 0;
        {
            const double weight = ((double)std::max<size_t>(n_tris, 1) / 31468.0) * (27400.0 / (double)std::max(g->cloud_hint, 1000u));
            const double left = std::max(5.0, 2800.0 * weight / (double)std::max(nq, 1u));   // passes of this sub-batch's loop still to run
            const double passes = (double)crit.max_iteration + 1.0;
            auto_overlap = left >= passes ? 0u : (uint32_t)(passes - left + 0.5);
            // kd-tree scenes: the loop is seven times a render and its first passes are the heavy ones -- the other slot's render (and with it
            // that slot's own first passes) should start at once: 37.3 / 37.5 k against 36.0 k poses/s for the rule above (pass 1 / 2 / 3 / 6: 37.2 /
            // 37.2 / 37.1 / 37.0 k; 10: 35.7 k; 16: 34.0 k)
            if (scene_kind == PR_SCENE_NN) auto_overlap = 0u;
        }
        prk::IcpBatch b{};
        b.cloud = sl.cloud.as<pr_vec3>(); b.nblk = nblk; b.grid_x = grid_x; b.steps = steps;
        if (nn_prev) {
            b.nn_prev = nn_prev; b.nn_slack = reinterpret_cast<float *>(nn_prev + nn_span);
            b.nn_queue = reinterpret_cast<uint2 *>(nn_prev + 2 * nn_span); b.nn_queue2 = reinterpret_cast<uint2 *>(nn_prev + 4 * nn_span); b.nn_qcount = nn_prev + 6 * nn_span;
        }
        for (uint32_t it = 0; it <= (uint32_t)crit.max_iteration; ++it) {
            for (uint32_t grp = 0; grp < n_groups; ++grp) {
                const uint32_t p0 = group_begin(grp), np = group_begin(grp + 1) - p0;
                hipStream_t gs = grp ? sl.side[grp - 1] : st;
                prk::IcpBatch bb = b;
                bb.meta = meta + p0; bb.partial = sl.partial.as<float>() + (size_t)p0 * nblk * prk::kAccStride;
                if (bb.nn_qcount) bb.nn_qcount += prk::kQCountStride * (size_t)p0;
                bb.iter = it;
                if (fused) { bb.fused = 1; bb.crit = crit; bb.st = dstate + p0; bb.arrive = arrive + p0; }
                bb.score_only = (it == (uint32_t)crit.max_iteration) ? 1u : 0u;
                if (timed) {                                         // (one group: gs == st)
                    const size_t e0 = t_begin();
                    // a kd-tree pass is four kernels: three more events between them give each kernel's own time (pr_profile_nn)
                    const bool marks = sc.kind == PR_SCENE_NN && sc.nn_split && bb.nn_prev && np <= 32768u;
                    size_t mi[3] = { 0, 0, 0 }; hipEvent_t me[3] = { nullptr, nullptr, nullptr };
                    bool marks_ok = marks;
                    if (marks) for (int k = 0; k < 3; ++k) { mi[k] = t_event(); if (mi[k] == kNoEvent) marks_ok = false; else me[k] = sl.t_events[mi[k]]; }
                    HIP_TRY(launch_pass(bb, sc, np, gs, marks_ok ? me : nullptr));
                    t_end(e0, kSpanIcp, q0 + p0, np, it == 0 || it == (uint32_t)crit.max_iteration);
                    if (marks_ok && !t_fail) { Slot::TSpan &ts = sl.t_spans.back(); ts.marks = true; for (int k = 0; k < 3; ++k) ts.m[k] = mi[k]; }
                } else HIP_TRY(launch_pass(bb, sc, np, gs));
                if (!fused) HIP_TRY(prk::launch_icp_finalize_solve(bb.partial, meta + p0, nblk, steps, dstate + p0, crit, it, np, gs));
            }
            if (q0 + sub >= P && it == (timed ? (uint32_t)crit.max_iteration : std::min<uint32_t>((uint32_t)crit.max_iteration, (pipeline_start && opt.start_overlap >= 0) ? (uint32_t)opt.start_overlap : (opt.overlap_pass >= 0 ? (uint32_t)opt.overlap_pass : auto_overlap)))) {
                HIP_TRY(hipEventRecord(sl.progress, st));
                sl.progress_valid = true;
            }
        }
        for (uint32_t k = 1; k < n_groups; ++k) { HIP_TRY(hipEventRecord(sl.join[k - 1], sl.side[k - 1])); HIP_TRY(hipStreamWaitEvent(st, sl.join[k - 1], 0)); }
    }
    HIP_TRY(prk::launch_pack_export(sl.dstate.as<prk::DevIcpState>(), dres, sl.counts.as<uint32_t>(), static_cast<uint32_t *>(h_out_dev),
                                    results_host ? reinterpret_cast<pr_result *>(static_cast<unsigned char *>(h_out_dev) + res_off) : nullptr, P, st));
    HIP_TRY(hipEventRecord(sl.done, st));
    if (t_fail) { sl.timed = false; sl.t_spans.clear(); sl.t_used = 0; g->stat_timing_dropped++; }   // the batch runs; its timing is dropped, and counted
    return PR_OK;
}

// KDTree_cpu::build_tree on the device: level loop driven from the host (one 16-byte read-back per level)
int kd_build_dev(pr_vec3 *pcd, pr_vec3 *nrm, uint32_t n, int max_leaf, pr_kdnode *nodes, size_t cap, uint32_t *n_nodes)
{
    if (n == 0 || cap == 0) { set_error("kd-tree build: no points"); return PR_ERR_INVALID; }
    const uint32_t cap32 = (uint32_t)std::min<size_t>(cap, 0x7fffffff);
    PR_TRY(g->kd_idx.ensure(sizeof(int) * n));
    PR_TRY(g->kd_scratch.ensure(sizeof(int) * n));
    PR_TRY(g->kd_child.ensure(sizeof(int) * cap32));
    PR_TRY(g->kd_ctrl.ensure(sizeof(uint32_t) * 4));
    PR_TRY(g->kd_tmp.ensure(sizeof(pr_vec3) * 2 * (size_t)n));
    HIP_TRY(prk::launch_kd_init(nodes, cap32, g->kd_idx.as<int>(), n, g->kd_ctrl.as<uint32_t>(), g->stream));
    uint32_t ctrl[4] = { 0, 1, 1, 1 };
    for (int level = 0; level < 4096; ++level) {
        HIP_TRY(prk::launch_kd_level(nodes, g->kd_ctrl.as<uint32_t>(), max_leaf, g->kd_child.as<int>(), cap32, 0, pcd, g->kd_idx.as<int>(),
                                     g->kd_scratch.as<int>(), /*plan_only=*/true, g->stream));
        HIP_TRY(hipMemcpyAsync(ctrl, g->kd_ctrl.p, sizeof ctrl, hipMemcpyDeviceToHost, g->stream));
        HIP_TRY(hipStreamSynchronize(g->stream));
        if (ctrl[3] > cap32) { set_error("kd-tree build: node capacity %u too small", cap32); return PR_ERR_NOMEM; }
        if (ctrl[3] == ctrl[2]) break;                             // no node of this level split: done (pcd_scene.cpp:166-168)
        HIP_TRY(prk::launch_kd_level(nodes, g->kd_ctrl.as<uint32_t>(), max_leaf, g->kd_child.as<int>(), cap32, ctrl[1] - ctrl[0], pcd,
                                     g->kd_idx.as<int>(), g->kd_scratch.as<int>(), /*plan_only=*/false, g->stream));
    }
    pr_vec3 *tp = g->kd_tmp.as<pr_vec3>(), *tn = tp + n;
    HIP_TRY(prk::launch_kd_permute(pcd, nrm, g->kd_idx.as<int>(), n, tp, tn, g->stream));
    HIP_TRY(hipMemcpyAsync(pcd, tp, sizeof(pr_vec3) * n, hipMemcpyDeviceToDevice, g->stream));
    HIP_TRY(hipMemcpyAsync(nrm, tn, sizeof(pr_vec3) * n, hipMemcpyDeviceToDevice, g->stream));
    HIP_TRY(hipStreamSynchronize(g->stream));
    if (n_nodes) *n_nodes = ctrl[2];
    return PR_OK;
}

template <typename T>
int scene_nn_prepare_dev_t(const T *depth, const float K[9], uint32_t W, uint32_t H, int max_leaf, pr_vec3 *pcd, pr_vec3 *nrm,
                           pr_kdnode *nodes, size_t cap, uint32_t *n_points, uint32_t *n_nodes)
{
    const size_t px = (size_t)W * H;
    PR_TRY(g->nn_full.ensure(sizeof(pr_vec3) * 2 * px));
    PR_TRY(g->row_count.ensure(sizeof(uint32_t) * H));
    PR_TRY(g->row_off.ensure(sizeof(uint32_t) * H));
    PR_TRY(g->counts.ensure(sizeof(uint32_t)));
    pr_vec3 *full_pcd = g->nn_full.as<pr_vec3>(), *full_nrm = full_pcd + px;
    // normals of every pixel (get_normal sees the uint16 image), then the valid pixels in row-major order
    HIP_TRY(prk::launch_scene_proj_prepare<T>(depth, W, H, K[0], K[4], K[2], K[5], full_pcd, full_nrm, g->stream));
    HIP_TRY(prk::launch_nn_gather<T>(depth, W, H, K[0], K[4], K[2], K[5], full_nrm, g->row_count.as<uint32_t>(), g->row_off.as<uint32_t>(),
                                     g->counts.as<uint32_t>(), nullptr, nullptr, false, g->stream));
    uint32_t n = 0;
    HIP_TRY(hipMemcpyAsync(&n, g->counts.p, sizeof n, hipMemcpyDeviceToHost, g->stream));
    HIP_TRY(hipStreamSynchronize(g->stream));
    if (n_points) *n_points = n;
    if (n == 0) { if (n_nodes) *n_nodes = 0; return PR_OK; }
    HIP_TRY(prk::launch_nn_gather<T>(depth, W, H, K[0], K[4], K[2], K[5], full_nrm, g->row_count.as<uint32_t>(), g->row_off.as<uint32_t>(),
                                     g->counts.as<uint32_t>(), pcd, nrm, true, g->stream));
    return kd_build_dev(pcd, nrm, n, max_leaf, nodes, cap, n_nodes);
}

// ---- RCCL, opened on first use ----------------------------------------------------------------------------
// The gather of the solved transforms is the job's only collective.  librccl is half a gigabyte of code objects
// that a single-GPU host never needs, and a Python host usually has one loaded already (PyTorch's): dlopen by soname picks
// that one up, otherwise the ROCm copy is loaded.
struct Rccl {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
Rccl g_rccl;
int rccl_load()
{
    std::lock_guard<std::mutex> lk(g_reg_mu);
    if (g_rccl.lib) return PR_OK;
    void *h = nullptr;
    for (const char *name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" }) { h = dlopen(name, RTLD_NOW | RTLD_LOCAL); if (h) break; }
    if (!h) { set_error("cannot open librccl (%s)", dlerror()); return PR_ERR_COMM; }
    Rccl r; r.lib = h;
#define PR_SYM(field, name) r.field = reinterpret_cast<decltype(r.field)>(dlsym(h, name)); if (!r.field) { set_error("librccl lacks %s", name); dlclose(h); return PR_ERR_COMM; }
    PR_SYM(GetUniqueId, "ncclGetUniqueId") PR_SYM(CommInitRank, "ncclCommInitRank") PR_SYM(CommInitAll, "ncclCommInitAll")
    PR_SYM(CommDestroy, "ncclCommDestroy") PR_SYM(GroupStart, "ncclGroupStart") PR_SYM(GroupEnd, "ncclGroupEnd")
    PR_SYM(Send, "ncclSend") PR_SYM(Recv, "ncclRecv") PR_SYM(GetErrorString, "ncclGetErrorString")
#undef PR_SYM
    g_rccl = r;
    return PR_OK;
}
#define NCCL_TRY(expr)                                                                          \
    do {                                                                                        \
        ncclResult_t r_ = (expr);                                                               \
        if (r_ != ncclSuccess) { set_error("%s failed: %s", #expr, g_rccl.GetErrorString(r_)); return PR_ERR_COMM; } \
    } while (0)

void comm_teardown(Ctx *c)
{
    if (c->comm && g_rccl.CommDestroy) { (void)hipSetDevice(c->device); if (c->stream) (void)hipStreamSynchronize(c->stream); (void)g_rccl.CommDestroy(c->comm); }
    c->comm = nullptr; c->comm_rank = 0; c->comm_world = 1;
}

void ctx_teardown(Ctx *c)        // c->mu held (or c unreachable); the calling thread is bound to c
{
    if (!c->ready) return;
    (void)hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    for (Slot &sl : c->slots) slot_release(sl);
    for (DevBuf *b : { &c->aabb, &c->aabb_keys, &c->bbox, &c->poses, &c->depth, &c->row_count, &c->row_off, &c->counts, &c->cloud, &c->meta, &c->partial,
                       &c->sums, &c->packed.rec, &c->topo, &c->bmin, &c->bmax, &c->pts, &c->nnrec, &c->nnrec32, &c->nndesc, &c->nnwide, &c->nnwq, &c->nn_prev, &c->nndepth, &c->dstate, &c->dresults, &c->vbdesc, &c->flowsync, &c->arrive, &c->conv16, &c->conv8, &c->kd_idx, &c->kd_scratch, &c->kd_child, &c->kd_ctrl, &c->kd_tmp, &c->nn_full, &c->gather_tmp, &c->nn_cells, &c->nn_grid, &c->nn_counters }) b->release();
    for (PinBuf *b : { &c->h_sums, &c->h_meta, &c->h_counts, &c->h_results, &c->h_dstate, &c->h_flow }) b->release();
    c->packed = PackedCache(); c->nn_cache.valid = false;
    for (auto &gr : c->graphs) destroy_graph(gr);
    c->graphs.clear();
    for (hipEvent_t e : c->ev_pool) hipEventDestroy(e);
    for (auto &e : c->gather_ev) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
    c->gather_ev.clear(); c->gather_ev_used = 0;
    c->ev_pool.clear(); c->ev_used = 0; c->spans.clear();
    hipStreamDestroy(c->stream);
    if (c->ev_fork) hipEventDestroy(c->ev_fork);
    c->ev_fork = nullptr;
    for (int i = 0; i < 3; ++i) {
        if (c->side[i]) hipStreamDestroy(c->side[i]);
        if (c->ev_join[i]) hipEventDestroy(c->ev_join[i]);
        c->side[i] = nullptr; c->ev_join[i] = nullptr;
    }
    c->stream = nullptr; c->ready = false; c->mesh_key = nullptr; c->aabb_host_valid = false; c->cloud_hint = 0;
}

PrivateCtx::~PrivateCtx()               // a thread that ends with a private context releases it
{
    if (!c) return;
    private_unregister(c);
    { std::lock_guard<std::mutex> lk(c->mu); comm_teardown(c); ctx_teardown(c); }
    delete c; c = nullptr;
}

}  // namespace

// =================================================================================================
extern "C" {

const char *pr_last_error(void) { return prh::g_err.c_str(); }
const char *pr_version(void) { return "pose_refine_amd 0.1 (gfx950)"; }
int pr_abi_version(void) { return PR_ABI_VERSION; }
static_assert(sizeof(pr_scene_nn) == 72, "pr_scene_nn layout (PR_ABI_VERSION)");

int pr_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static void hw_queues_hint()
{
    // The two asynchronous slots and their pose groups need four streams that really run side by side; the runtime maps all
    // streams of a process onto GPU_MAX_HW_QUEUES hardware queues (default 4, shared with every other stream of the process),
    // and streams that share a queue serialise.  This only takes effect if the HIP runtime has not been initialised yet --
    // hosts that initialise it earlier (PyTorch) set the variable themselves, as bench.py does.
    // 16: the slots' four streams, the third pose group of kd-tree batches and the streams of private contexts (host threads issuing batches
    // of their own) all get a queue to themselves -- with 8, two such threads beside the slots ran at 166-198 k poses/s instead of 236-240 k.
    setenv("GPU_MAX_HW_QUEUES", "16", /*overwrite=*/0);
}

int pr_init(int device)
{
    hw_queues_hint();
    if (device < 0) { set_error("pr_init: device must be >= 0"); return PR_ERR_INVALID; }
    if (tl_private.c && tl_private.c->device != device) { set_error("pr_init: this thread owns a private context on device %d (pr_thread_context(0) first)", tl_private.c->device); return PR_ERR_INVALID; }
    if (!tl_private.c) PR_TRY(bind_shared(device));
    std::lock_guard<std::mutex> lk(g->mu);
    return require_ctx();
}
int pr_set_device(int device) { return pr_init(device); }

int pr_shutdown(void)
{
    if (!g) return PR_OK;
    std::lock_guard<std::mutex> lk(g->mu);
    comm_teardown(g);
    ctx_teardown(g);
    return PR_OK;
}

// A private context for the calling thread on its current device: own stream, workspaces, slots and caches, so that host threads
// that each drive their own hypotheses (the reference's usage: icp.cu:170,178 cudaStreamPerThread, README.md:15) do not
// serialise on the device's shared context.  pr_thread_context(0) (or thread exit) releases it and re-binds to the shared one.
int pr_thread_context(int enable)
{
    PR_TRY(bind_default());
    if (enable) {
        if (tl_private.c) return PR_OK;
        Ctx *c = new Ctx();
        c->device = g->device; c->is_private = true;
        private_register(c);
        tl_private.c = c; g = c;
        std::lock_guard<std::mutex> lk(g->mu);
        return require_ctx();
    }
    if (!tl_private.c) return PR_OK;
    const int dev = tl_private.c->device;
    private_unregister(tl_private.c);
    { std::lock_guard<std::mutex> lk(tl_private.c->mu); comm_teardown(tl_private.c); ctx_teardown(tl_private.c); }
    delete tl_private.c; tl_private.c = nullptr; g = nullptr;
    return bind_shared(dev);
}

int pr_sync(void)
{
    PR_ENTER();
    HIP_TRY(hipStreamSynchronize(g->stream));
    return PR_OK;
}

int pr_malloc(void **dev_ptr, size_t bytes)
{
    PR_ENTER();
    if (!dev_ptr) { set_error("pr_malloc: null out pointer"); return PR_ERR_INVALID; }
    HIP_TRY(hipMalloc(dev_ptr, bytes ? bytes : 1));
    return PR_OK;
}
int pr_invalidate(const void *dev_ptr, size_t bytes)
{
    if (!dev_ptr) return PR_OK;
    g_writes.note(dev_ptr, bytes);
    return PR_OK;
}
int pr_free(void *dev_ptr)
{
    if (!dev_ptr) return PR_OK;
    Ctx *self = nullptr;
    {
        PR_ENTER();
        self = g;
        // nothing this context still has in flight may outlive the buffer: the library stream and every slot with an unfinished batch
        HIP_TRY(hipStreamSynchronize(g->stream));
        for (Slot &sl : g->slots) if (sl.pending && !sl.delivered) slot_drain(sl);
        g_writes.note(dev_ptr, 0);                                     // the address may come back with other content
        if (dev_ptr == g->mesh_key) { g->mesh_key = nullptr; g->aabb_host_valid = false; }
    }
    // ... nor anything another context of this device has in flight (a private-context thread, or another thread on the shared context,
    // may have a batch running on the buffer): every other context is locked -- i.e. between two of its calls, never inside a stream
    // capture -- and its streams are waited for, one context at a time.  (hipDeviceSynchronize would do it in one call, but it breaks
    // a graph capture another thread has open.)
    {
        std::vector<Ctx *> others;
        { std::lock_guard<std::mutex> lk(g_reg_mu); for (Ctx *c : g_shared) if (c && c != self && c->device == self->device) others.push_back(c); }
        std::lock_guard<std::mutex> plk(g_private_mu);
        for (Ctx *c : g_private) if (c != self && c->device == self->device) others.push_back(c);
        for (Ctx *c : others) {
            std::lock_guard<std::mutex> lk(c->mu);
            if (!c->ready) continue;
            if (c->stream) (void)hipStreamSynchronize(c->stream);
            for (hipStream_t sd : c->side) if (sd) (void)hipStreamSynchronize(sd);
            for (Slot &sl : c->slots) slot_drain(sl);
        }
    }
    HIP_TRY(hipFree(dev_ptr));
    return PR_OK;
}
static int copy_sync(void *dst, const void *src, size_t bytes, hipMemcpyKind kind)
{
    PR_ENTER();
    if (bytes == 0) return PR_OK;
    if (kind != hipMemcpyDeviceToHost) {
        g_writes.note(dst, bytes);
        if (g->mesh_key && reinterpret_cast<uintptr_t>(dst) < reinterpret_cast<uintptr_t>(g->mesh_key) + g->mesh_n * sizeof(pr_triangle) &&
            reinterpret_cast<uintptr_t>(g->mesh_key) < reinterpret_cast<uintptr_t>(dst) + bytes) { g->mesh_key = nullptr; g->aabb_host_valid = false; }
    }
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, kind, g->stream));
    HIP_TRY(hipStreamSynchronize(g->stream));
    return PR_OK;
}
int pr_memcpy_h2d(void *d, const void *s, size_t n) { return copy_sync(d, s, n, hipMemcpyHostToDevice); }
int pr_memcpy_d2h(void *d, const void *s, size_t n) { return copy_sync(d, s, n, hipMemcpyDeviceToHost); }
int pr_memcpy_d2d(void *d, const void *s, size_t n) { return copy_sync(d, s, n, hipMemcpyDeviceToDevice); }

int pr_fill_i32(int32_t *dev_dst, size_t count, int32_t value)
{
    PR_ENTER();
    if (!dev_dst && count) { set_error("pr_fill_i32: null destination"); return PR_ERR_INVALID; }
    g_writes.note(dev_dst, count * sizeof(int32_t));
    HIP_TRY(prk::launch_fill_i32(dev_dst, count, value, g->stream));
    HIP_TRY(hipStreamSynchronize(g->stream));
    return PR_OK;
}

int pr_render(const pr_triangle *tris_dev, size_t n_tris, const pr_mat4 *poses_host, size_t n_poses, size_t width, size_t height,
              const pr_mat4 *proj, pr_roi roi, int32_t *depth_dev_out)
{
    PR_ENTER();
    if (depth_dev_out) g_writes.note(depth_dev_out, sizeof(int32_t) * n_poses * ((roi.width > 0 && roi.height > 0) ? (size_t)roi.width * roi.height : width * height));
    PR_TRY(render_impl(tris_dev, n_tris, poses_host, n_poses, width, height, proj, roi, depth_dev_out, true));
    HIP_TRY(hipStreamSynchronize(g->stream));            // renderer.cu:295 cudaDeviceSynchronize
    drain_spans();
    return PR_OK;
}

int pr_render_to_host(const pr_triangle *tris_dev, size_t n_tris, const pr_mat4 *poses_host, size_t n_poses, size_t width, size_t height,
                      const pr_mat4 *proj, pr_roi roi, int32_t *depth_host_out)
{
    PR_ENTER();
    size_t rw = width, rh = height;
    if (roi.width > 0 && roi.height > 0) { rw = (size_t)roi.width; rh = (size_t)roi.height; }
    PR_TRY(g->depth.ensure(sizeof(int32_t) * std::max<size_t>(1, n_poses * rw * rh)));
    PR_TRY(render_impl(tris_dev, n_tris, poses_host, n_poses, width, height, proj, roi, g->depth.as<int32_t>(), true));
    HIP_TRY(hipMemcpyAsync(depth_host_out, g->depth.p, sizeof(int32_t) * n_poses * rw * rh, hipMemcpyDeviceToHost, g->stream));
    HIP_TRY(hipStreamSynchronize(g->stream));
    drain_spans();
    return PR_OK;
}

int pr_depth2cloud_i32(const int32_t *depth_dev, uint32_t width, uint32_t height, const float K[9], uint32_t stride, uint32_t tl_x,
                       uint32_t tl_y, pr_vec3 **cloud_dev_out, uint32_t *n_points)
{
    PR_ENTER();
    return depth2cloud_impl<int32_t>(depth_dev, width, height, K, stride, tl_x, tl_y, cloud_dev_out, n_points);
}
int pr_depth2cloud_u16(const uint16_t *depth_dev, uint32_t width, uint32_t height, const float K[9], uint32_t stride, uint32_t tl_x,
                       uint32_t tl_y, pr_vec3 **cloud_dev_out, uint32_t *n_points)
{
    PR_ENTER();
    return depth2cloud_impl<uint16_t>(depth_dev, width, height, K, stride, tl_x, tl_y, cloud_dev_out, n_points);
}

int pr_scene_proj_prepare_dev(const void *depth_dev, int depth_is_i32, const float K[9], size_t width, size_t height,
                              pr_vec3 *pcd_dev_out, pr_vec3 *normal_dev_out)
{
    PR_ENTER();
    if (!depth_dev || !K || !pcd_dev_out || !normal_dev_out || width == 0 || height == 0) { set_error("pr_scene_proj_prepare_dev: bad arguments"); return PR_ERR_INVALID; }
    g_writes.note(pcd_dev_out, width * height * sizeof(pr_vec3)); g_writes.note(normal_dev_out, width * height * sizeof(pr_vec3));
    if (depth_is_i32) HIP_TRY(prk::launch_scene_proj_prepare<int32_t>(static_cast<const int32_t *>(depth_dev), (uint32_t)width, (uint32_t)height, K[0], K[4], K[2], K[5], pcd_dev_out, normal_dev_out, g->stream));
    else HIP_TRY(prk::launch_scene_proj_prepare<uint16_t>(static_cast<const uint16_t *>(depth_dev), (uint32_t)width, (uint32_t)height, K[0], K[4], K[2], K[5], pcd_dev_out, normal_dev_out, g->stream));
    HIP_TRY(hipStreamSynchronize(g->stream));
    return PR_OK;
}

int pr_kdtree_build_dev(pr_vec3 *pcd_dev, pr_vec3 *normal_dev, size_t n_points, int max_leaf, pr_kdnode *nodes_dev_out, size_t cap_nodes, uint32_t *n_nodes)
{
    PR_ENTER();
    if (!pcd_dev || !normal_dev || !nodes_dev_out) { set_error("pr_kdtree_build_dev: bad arguments"); return PR_ERR_INVALID; }
    g_writes.note(pcd_dev, n_points * sizeof(pr_vec3)); g_writes.note(normal_dev, n_points * sizeof(pr_vec3)); g_writes.note(nodes_dev_out, cap_nodes * sizeof(pr_kdnode));
    return kd_build_dev(pcd_dev, normal_dev, (uint32_t)n_points, max_leaf, nodes_dev_out, cap_nodes, n_nodes);
}

int pr_scene_nn_prepare_dev(const void *depth_dev, int depth_is_i32, const float K[9], int width, int height, int max_leaf,
                            pr_vec3 *pcd_dev_out, pr_vec3 *normal_dev_out, pr_kdnode *nodes_dev_out, size_t cap_nodes,
                            uint32_t *n_points, uint32_t *n_nodes)
{
    PR_ENTER();
    if (!depth_dev || !K || !pcd_dev_out || !normal_dev_out || !nodes_dev_out || width <= 0 || height <= 0) { set_error("pr_scene_nn_prepare_dev: bad arguments"); return PR_ERR_INVALID; }
    g_writes.note(pcd_dev_out, (size_t)width * height * sizeof(pr_vec3)); g_writes.note(normal_dev_out, (size_t)width * height * sizeof(pr_vec3)); g_writes.note(nodes_dev_out, cap_nodes * sizeof(pr_kdnode));
    if (depth_is_i32) return scene_nn_prepare_dev_t<int32_t>(static_cast<const int32_t *>(depth_dev), K, (uint32_t)width, (uint32_t)height, max_leaf,
                                                             pcd_dev_out, normal_dev_out, nodes_dev_out, cap_nodes, n_points, n_nodes);
    return scene_nn_prepare_dev_t<uint16_t>(static_cast<const uint16_t *>(depth_dev), K, (uint32_t)width, (uint32_t)height, max_leaf,
                                            pcd_dev_out, normal_dev_out, nodes_dev_out, cap_nodes, n_points, n_nodes);
}

int pr_raw2depth_mask(const int32_t *raw_dev, size_t count, uint16_t *depth_host_out, uint8_t *mask_host_out)
{
    PR_ENTER();
    if (count == 0) return PR_OK;                                  // (an empty render stack has no array)
    if (!raw_dev || (!depth_host_out && !mask_host_out)) { set_error("pr_raw2depth_mask: bad arguments"); return PR_ERR_INVALID; }
    if (depth_host_out) PR_TRY(g->conv16.ensure(count * sizeof(uint16_t) + 16));
    if (mask_host_out) PR_TRY(g->conv8.ensure(count + 16));
    HIP_TRY(prk::launch_raw2depth_mask(raw_dev, count, depth_host_out ? g->conv16.as<uint16_t>() : nullptr, mask_host_out ? g->conv8.as<uint8_t>() : nullptr, g->stream));
    // ONE copy per output for the whole stack (the reference issues one thrust::copy per pose, renderer.cu:370-373)
    if (depth_host_out) HIP_TRY(hipMemcpyAsync(depth_host_out, g->conv16.p, count * sizeof(uint16_t), hipMemcpyDeviceToHost, g->stream));
    if (mask_host_out) HIP_TRY(hipMemcpyAsync(mask_host_out, g->conv8.p, count, hipMemcpyDeviceToHost, g->stream));
    HIP_TRY(hipStreamSynchronize(g->stream));
    return PR_OK;
}

int pr_icp_batch(pr_vec3 *clouds_dev, const uint32_t *offsets_host, uint32_t n_clouds, int scene_kind, const void *scene,
                 pr_criteria crit, pr_result *results_host)
{
    PR_ENTER();
    if (!offsets_host || (n_clouds && !results_host)) { set_error("pr_icp_batch: bad arguments"); return PR_ERR_INVALID; }
    if (n_clouds == 0) return PR_OK;
    if (!clouds_dev) {                                            // clouds without a single point come without an array (an empty device_vector):
        if (offsets_host[n_clouds] != offsets_host[0]) { set_error("pr_icp_batch: null cloud array"); return PR_ERR_INVALID; }
        for (uint32_t i = 0; i < n_clouds; ++i) {                 // count == 0 -> identity, fitness 0, rmse 0 (icp.cu:183), like any empty cloud
            identity16(results_host[i].T); results_host[i].fitness = 0.0f; results_host[i].inlier_rmse = 0.0f;
        }
        return PR_OK;
    }
    SceneSel sc;
    std::memset(static_cast<void *>(&sc), 0, sizeof sc);
    PR_TRY(make_scene(scene_kind, scene, /*want_packed=*/false, sc));
    std::vector<uint32_t> start(n_clouds), count(n_clouds);
    for (uint32_t i = 0; i < n_clouds; ++i) {
        if (offsets_host[i + 1] < offsets_host[i]) { set_error("pr_icp_batch: offsets must be non-decreasing"); return PR_ERR_INVALID; }
        start[i] = offsets_host[i]; count[i] = offsets_host[i + 1] - offsets_host[i];
    }
    // the hypothesis index is the y dimension of every launch of the loop (65 535 at most): longer lists run in pieces, one after the other
    // (the clouds are independent of each other, so the pieces are)
#ifndef PR_MAX_CLOUDS_PER_RUN
#define PR_MAX_CLOUDS_PER_RUN 32768                               // (a larger value in an experiment build exercises the launchers' own split)
#endif
    constexpr uint32_t kMaxCloudsPerRun = PR_MAX_CLOUDS_PER_RUN;
    for (uint32_t c0 = 0; c0 < n_clouds; c0 += kMaxCloudsPerRun) {
        const uint32_t nc = std::min(kMaxCloudsPerRun, n_clouds - c0);
        PR_TRY(icp_drive(clouds_dev, start.data() + c0, count.data() + c0, nc, sc, crit, results_host + c0, nullptr));
    }
    return PR_OK;
}

int pr_icp_proj(pr_vec3 *cloud_dev, uint32_t n_points, const pr_scene_proj *scene, pr_criteria crit, pr_result *result_out)
{
    const uint32_t off[2] = { 0, n_points };
    return pr_icp_batch(cloud_dev, off, 1, PR_SCENE_PROJ, scene, crit, result_out);
}
int pr_icp_nn(pr_vec3 *cloud_dev, uint32_t n_points, const pr_scene_nn *scene, pr_criteria crit, pr_result *result_out)
{
    const uint32_t off[2] = { 0, n_points };
    return pr_icp_batch(cloud_dev, off, 1, PR_SCENE_NN, scene, crit, result_out);
}

// Audit entry: what ONE correspondence pass contributes, point by point (29 floats each: the 21 upper-triangle products of J J^T row by
// row, the 6 products J r, r^2 and the inlier flag -- thrust__pcd2Ab, icp.h:128-209), with the pending update applied to the cloud first
// like the fused pass does.  A caller that adds the rows in point order reproduces the reference's single-thread sums (icp.cpp:139-148).
int pr_debug_contrib29(pr_vec3 *cloud_dev, uint32_t n_points, int scene_kind, const void *scene, const float *update16, int want_packed, float *contrib_host)
{
    PR_ENTER();
    if (n_points == 0) return PR_OK;
    if (!cloud_dev || !contrib_host) { set_error("pr_debug_contrib29: null buffer"); return PR_ERR_INVALID; }
    SceneSel sc;
    std::memset(static_cast<void *>(&sc), 0, sizeof sc);
    PR_TRY(make_scene(scene_kind, scene, want_packed != 0, sc));
    DevBuf out;
    PR_TRY(out.ensure(sizeof(float) * 29 * (size_t)n_points));
    hipError_t e;
    if (sc.kind == PR_SCENE_NN) e = prk::launch_contrib29_nn(cloud_dev, n_points, update16, sc.nn, out.as<float>(), g->stream);
    else if (sc.packed) e = prk::launch_contrib29_proj_packed(cloud_dev, n_points, update16, sc.pk, out.as<float>(), g->stream);
    else e = prk::launch_contrib29_proj_aos(cloud_dev, n_points, update16, sc.aos, out.as<float>(), g->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(contrib_host, out.p, sizeof(float) * 29 * (size_t)n_points, hipMemcpyDeviceToHost, g->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(g->stream);
    g_writes.note(cloud_dev, sizeof(pr_vec3) * (size_t)n_points);
    out.release();
    if (e != hipSuccess) { (void)hipGetLastError(); set_error("pr_debug_contrib29: %s", hipGetErrorString(e)); return PR_ERR_HIP; }
    return PR_OK;
}

int pr_refine_batch_roi(const pr_triangle *tris_dev, size_t n_tris, const pr_mat4 *poses_host, uint32_t n_poses, uint32_t width, uint32_t height,
                        const pr_mat4 *proj, const float K[9], int scene_kind, const void *scene, pr_criteria crit, pr_roi roi,
                        pr_result *results_host, uint32_t *cloud_sizes_host)
{
    PR_ENTER();
    if (!results_host) { set_error("pr_refine_batch: results_host is null"); return PR_ERR_INVALID; }
    if (n_poses == 0) return PR_OK;
    const int slot = free_slot();
    if (slot < 0) { set_error("pr_refine_batch: both asynchronous slots hold unfinished batches (pr_refine_wait one of them first)"); return PR_ERR_INVALID; }
    PR_TRY(refine_submit(slot, tris_dev, n_tris, poses_host, n_poses, width, height, proj, K, scene_kind, scene, crit, roi, results_host, nullptr, cloud_sizes_host));
    return refine_wait(slot);
}
int pr_refine_batch(const pr_triangle *tris_dev, size_t n_tris, const pr_mat4 *poses_host, uint32_t n_poses, uint32_t width, uint32_t height,
                    const pr_mat4 *proj, const float K[9], int scene_kind, const void *scene, pr_criteria crit,
                    pr_result *results_host, uint32_t *cloud_sizes_host)
{
    return pr_refine_batch_roi(tris_dev, n_tris, poses_host, n_poses, width, height, proj, K, scene_kind, scene, crit, pr_roi{ 0, 0, 0, 0 }, results_host, cloud_sizes_host);
}
int pr_refine_batch_dev(const pr_triangle *tris_dev, size_t n_tris, const pr_mat4 *poses_host, uint32_t n_poses, uint32_t width, uint32_t height,
                        const pr_mat4 *proj, const float K[9], int scene_kind, const void *scene, pr_criteria crit,
                        pr_result *results_dev, uint32_t *cloud_sizes_host)
{
    PR_ENTER();
    if (!results_dev) { set_error("pr_refine_batch_dev: results_dev is null"); return PR_ERR_INVALID; }
    if (n_poses == 0) return PR_OK;
    const int slot = free_slot();
    if (slot < 0) { set_error("pr_refine_batch_dev: both asynchronous slots hold unfinished batches (pr_refine_wait one of them first)"); return PR_ERR_INVALID; }
    PR_TRY(refine_submit(slot, tris_dev, n_tris, poses_host, n_poses, width, height, proj, K, scene_kind, scene, crit, pr_roi{ 0, 0, 0, 0 }, nullptr, results_dev, cloud_sizes_host));
    return refine_wait(slot);
}
int pr_refine_submit_roi(int slot, const pr_triangle *tris_dev, size_t n_tris, const pr_mat4 *poses_host, uint32_t n_poses, uint32_t width, uint32_t height,
                         const pr_mat4 *proj, const float K[9], int scene_kind, const void *scene, pr_criteria crit, pr_roi roi,
                         pr_result *results_host, pr_result *results_dev, uint32_t *cloud_sizes_host)
{
    PR_ENTER();
    return refine_submit(slot, tris_dev, n_tris, poses_host, n_poses, width, height, proj, K, scene_kind, scene, crit, roi, results_host, results_dev, cloud_sizes_host);
}
int pr_refine_submit(int slot, const pr_triangle *tris_dev, size_t n_tris, const pr_mat4 *poses_host, uint32_t n_poses, uint32_t width, uint32_t height,
                     const pr_mat4 *proj, const float K[9], int scene_kind, const void *scene, pr_criteria crit,
                     pr_result *results_host, pr_result *results_dev, uint32_t *cloud_sizes_host)
{
    return pr_refine_submit_roi(slot, tris_dev, n_tris, poses_host, n_poses, width, height, proj, K, scene_kind, scene, crit, pr_roi{ 0, 0, 0, 0 }, results_host, results_dev, cloud_sizes_host);
}
int pr_refine_wait(int slot)
{
    PR_ENTER();
    return refine_wait(slot);
}

// Scene_projective restricted to a window of the frame: copies the window's rows of the full-frame arrays
int pr_scene_proj_crop_dev(const pr_vec3 *pcd_full_dev, const pr_vec3 *normal_full_dev, size_t width, size_t height, pr_roi window,
                           pr_vec3 *pcd_out_dev, pr_vec3 *normal_out_dev)
{
    PR_ENTER();
    if (!pcd_full_dev || !normal_full_dev || !pcd_out_dev || !normal_out_dev || window.width <= 0 || window.height <= 0 || window.x < 0 || window.y < 0 ||
        (size_t)window.x + (size_t)window.width > width || (size_t)window.y + (size_t)window.height > height) { set_error("pr_scene_proj_crop_dev: bad arguments"); return PR_ERR_INVALID; }
    const size_t cw = (size_t)window.width, ch = (size_t)window.height;
    g_writes.note(pcd_out_dev, cw * ch * sizeof(pr_vec3)); g_writes.note(normal_out_dev, cw * ch * sizeof(pr_vec3));
    const size_t off = (size_t)window.y * width + (size_t)window.x;
    HIP_TRY(hipMemcpy2DAsync(pcd_out_dev, cw * sizeof(pr_vec3), pcd_full_dev + off, width * sizeof(pr_vec3), cw * sizeof(pr_vec3), ch, hipMemcpyDeviceToDevice, g->stream));
    HIP_TRY(hipMemcpy2DAsync(normal_out_dev, cw * sizeof(pr_vec3), normal_full_dev + off, width * sizeof(pr_vec3), cw * sizeof(pr_vec3), ch, hipMemcpyDeviceToDevice, g->stream));
    HIP_TRY(hipStreamSynchronize(g->stream));
    return PR_OK;
}

// ---- the job's one collective -----------------------------------------------------------------------------------------
int pr_comm_id(unsigned char id_out[PR_COMM_ID_BYTES])
{
    static_assert(PR_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    if (!id_out) { set_error("pr_comm_id: null argument"); return PR_ERR_INVALID; }
    PR_TRY(rccl_load());
    ncclUniqueId id;
    NCCL_TRY(g_rccl.GetUniqueId(&id));
    std::memcpy(id_out, id.internal, PR_COMM_ID_BYTES);
    return PR_OK;
}
int pr_comm_init_rank(const unsigned char id_in[PR_COMM_ID_BYTES], int rank, int world)
{
    if (!id_in || world < 1 || rank < 0 || rank >= world) { set_error("pr_comm_init_rank: bad arguments"); return PR_ERR_INVALID; }
    PR_TRY(rccl_load());
    PR_ENTER();
    comm_teardown(g);
    ncclUniqueId id;
    std::memcpy(id.internal, id_in, PR_COMM_ID_BYTES);
    NCCL_TRY(g_rccl.CommInitRank(&g->comm, world, id, rank));
    g->comm_rank = rank; g->comm_world = world;
    return PR_OK;
}
int pr_comm_init_all(int n_devices)
{
    if (n_devices < 1) { set_error("pr_comm_init_all: n_devices must be >= 1"); return PR_ERR_INVALID; }
    PR_TRY(rccl_load());
    Ctx *mine = g;
    std::vector<Ctx *> cs((size_t)n_devices);
    std::vector<int> devs((size_t)n_devices);
    for (int d = 0; d < n_devices; ++d) {                          // the shared context of every device, initialised
        PR_TRY(bind_shared(d));
        std::lock_guard<std::mutex> lk(g->mu);
        PR_TRY(require_ctx());
        comm_teardown(g);
        cs[(size_t)d] = g; devs[(size_t)d] = d;
    }
    std::vector<ncclComm_t> comms((size_t)n_devices, nullptr);
    NCCL_TRY(g_rccl.CommInitAll(comms.data(), n_devices, devs.data()));
    for (int d = 0; d < n_devices; ++d) { cs[(size_t)d]->comm = comms[(size_t)d]; cs[(size_t)d]->comm_rank = d; cs[(size_t)d]->comm_world = n_devices; }
    g = mine;                                                       // the calling thread keeps the context it had (or none)
    if (g) (void)hipSetDevice(g->device);
    return PR_OK;
}
int pr_comm_destroy(void)
{
    if (!g) return PR_OK;
    std::lock_guard<std::mutex> lk(g->mu);
    comm_teardown(g);
    return PR_OK;
}
int pr_comm_rank(int *rank, int *world)
{
    PR_TRY(bind_default());
    if (rank) *rank = g->comm_rank;
    if (world) *world = g->comm_world;
    return PR_OK;
}
// Gather of the sharded results to `root`, in global hypothesis order: rank r contributes the pr_pr_shard_range(n_total, r, world)
// block.  Grouped ncclSend / ncclRecv of exactly the bytes each rank owns (72 B per hypothesis; 36.9 KB per rank at 4096
// hypotheses on 8 GPUs -- one hop over xGMI, latency-bound).  Enqueued on the context's stream; pr_sync / pr_memcpy_d2h order after it.
int pr_gather_results(const pr_result *send_dev, uint32_t n_local, uint32_t n_total, int root, pr_result *recv_dev)
{
    PR_ENTER();
    const int world = g->comm_world, rank = g->comm_rank;
    if (root < 0 || root >= world) { set_error("pr_gather_results: root %d outside 0..%d", root, world - 1); return PR_ERR_INVALID; }
    uint32_t first = 0, count = 0;
    pr_shard_range(n_total, (uint32_t)rank, (uint32_t)world, &first, &count);
    if (count != n_local) { set_error("pr_gather_results: rank %d holds %u results, its shard of %u over %d ranks has %u", rank, n_local, n_total, world, count); return PR_ERR_INVALID; }
    if ((n_local && !send_dev) || (rank == root && n_total && !recv_dev)) { set_error("pr_gather_results: null buffer"); return PR_ERR_INVALID; }
    if (!g->comm) {                                               // no communicator: a single-rank job gathers by copying
        if (n_local && recv_dev != send_dev) HIP_TRY(hipMemcpyAsync(recv_dev, send_dev, sizeof(pr_result) * n_local, hipMemcpyDeviceToDevice, g->stream));
        if (n_local) g_writes.note(recv_dev, sizeof(pr_result) * n_local);
        return PR_OK;
    }
    std::pair<hipEvent_t, hipEvent_t> *ev = nullptr;                // option "profile": the exchange's own time on the stream it runs on
    if (opt.profile != 0) {
        if (g->gather_ev_used == g->gather_ev.size()) { std::pair<hipEvent_t, hipEvent_t> e; HIP_TRY(hipEventCreate(&e.first)); HIP_TRY(hipEventCreate(&e.second)); g->gather_ev.push_back(e); }
        ev = &g->gather_ev[g->gather_ev_used++];
        HIP_TRY(hipEventRecord(ev->first, g->stream));
    }
    NCCL_TRY(g_rccl.GroupStart());
    if (n_local) NCCL_TRY(g_rccl.Send(send_dev, (size_t)n_local * sizeof(pr_result), ncclChar, root, g->comm, g->stream));
    if (rank == root) {
        for (int r = 0; r < world; ++r) {
            uint32_t f = 0, c = 0;
            pr_shard_range(n_total, (uint32_t)r, (uint32_t)world, &f, &c);
            if (c) NCCL_TRY(g_rccl.Recv(recv_dev + f, (size_t)c * sizeof(pr_result), ncclChar, r, g->comm, g->stream));
        }
        g_writes.note(recv_dev, sizeof(pr_result) * n_total);
    }
    NCCL_TRY(g_rccl.GroupEnd());
    if (ev) HIP_TRY(hipEventRecord(ev->second, g->stream));
    return PR_OK;
}
int pr_gather_profile(double *gather_ms, uint64_t *gathers)
{
    PR_ENTER();
    HIP_TRY(hipStreamSynchronize(g->stream));
    for (size_t i = 0; i < g->gather_ev_used; ++i) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, g->gather_ev[i].first, g->gather_ev[i].second) == hipSuccess) { g->gather_ms += ms; g->gather_n++; }
    }
    g->gather_ev_used = 0;
    if (gather_ms) *gather_ms = g->gather_ms;
    if (gathers) *gathers = g->gather_n;
    return PR_OK;
}

int pr_set_option(const char *name, int value)
{
    std::lock_guard<std::mutex> lk(g_reg_mu);
    if (!name) { set_error("pr_set_option: null name"); return PR_ERR_INVALID; }
    const std::string n(name);
    if (n == "solve") { if (value != PR_SOLVE_HOST && value != PR_SOLVE_DEVICE) { set_error("solve must be 0 or 1"); return PR_ERR_INVALID; } opt.solve_mode = value; }
    else if (n == "host_worker") opt.host_worker = value ? 1 : 0;
    else if (n == "start_overlap") opt.start_overlap = value;
    else if (n == "points_per_block") { if (value < 1024 || value % 1024) { set_error("points_per_block must be a multiple of 1024"); return PR_ERR_INVALID; } opt.steps = value / 1024; }
    else if (n == "profile") { if (value < 0 || value > 3) { set_error("profile must be 0, 1 (every launch, synchronous calls), 2 (every launch of one call in sample_period) or 3 (every launch, asynchronous batches stay asynchronous)"); return PR_ERR_INVALID; } opt.profile = value; }
    else if (n == "sample_period") opt.sample_period = std::max(1, value);
    else if (n == "scene_cache") opt.scene_cache = value ? 1 : 0;
    else if (n == "nn_lds_nodes") opt.nn_lds_nodes = std::max(0, value);
    else if (n == "nn_lds_records") opt.nn_lds_records = std::max(0, value);
    else if (n == "nn_compact") opt.nn_compact = value ? 1 : 0;
    else if (n == "nn_wide") opt.nn_wide = value ? 1 : 0;
    else if (n == "blocking_wait") opt.blocking_wait = value ? 1 : 0;
    else if (n == "nn_seed") opt.nn_seed = value ? 1 : 0;
    else if (n == "nn_stack") opt.nn_stack = value ? 1 : 0;
    else if (n == "nn_split") opt.nn_split = value ? 1 : 0;
    else if (n == "nn_run") opt.nn_run = std::min(256, std::max(1, value));
    else if (n == "nn_grid") opt.nn_grid = value ? 1 : 0;
    else if (n == "nn_count") opt.nn_count = value ? 1 : 0;
    else if (n == "graph") opt.use_graph = value ? 1 : 0;
    else if (n == "icp_flow") opt.icp_flow = value ? 1 : 0;
    else if (n == "fused_solve") opt.fused_solve = value ? 1 : 0;
    else if (n == "sub_batch") opt.sub_batch = std::min(32768, std::max(32, value));    // (the hypothesis index is the y dimension of the launches)
    else if (n == "overlap_pass") opt.overlap_pass = std::max(-1, value);
    else if (n == "pose_groups") opt.pose_groups = std::min(4, std::max(0, value));
    else if (n == "eager_streams") opt.eager_streams = value ? 1 : 0;
    else if (n == "raster_mode") { if (value < 0 || value > 1) { set_error("raster_mode must be 0 or 1"); return PR_ERR_INVALID; } opt.raster_mode = value; }
    else { set_error("unknown option %s", name); return PR_ERR_INVALID; }
    return PR_OK;
}
int pr_get_option(const char *name, int *value)
{
    std::lock_guard<std::mutex> lk(g_reg_mu);
    if (!name || !value) { set_error("pr_get_option: null argument"); return PR_ERR_INVALID; }
    const std::string n(name);
    if (n == "solve") *value = opt.solve_mode;
    else if (n == "host_worker") *value = opt.host_worker;
    else if (n == "start_overlap") *value = opt.start_overlap;
    else if (n == "points_per_block") *value = opt.steps * 1024;
    else if (n == "profile") *value = opt.profile;
    else if (n == "sample_period") *value = opt.sample_period;
    else if (n == "scene_cache") *value = opt.scene_cache;
    else if (n == "nn_lds_nodes") *value = opt.nn_lds_nodes;
    else if (n == "nn_lds_records") *value = opt.nn_lds_records;
    else if (n == "nn_compact") *value = opt.nn_compact;
    else if (n == "nn_wide") *value = opt.nn_wide;
    else if (n == "blocking_wait") *value = opt.blocking_wait;
    else if (n == "nn_seed") *value = opt.nn_seed;
    else if (n == "nn_stack") *value = opt.nn_stack;
    else if (n == "nn_split") *value = opt.nn_split;
    else if (n == "nn_run") *value = opt.nn_run;
    else if (n == "nn_grid") *value = opt.nn_grid;
    else if (n == "nn_count") *value = opt.nn_count;
    else if (n == "raster_mode") *value = opt.raster_mode;
    else if (n == "eager_streams") *value = opt.eager_streams;
    else if (n == "graph") *value = opt.use_graph;
    else if (n == "icp_flow") *value = opt.icp_flow;
    else if (n == "fused_solve") *value = opt.fused_solve;
    else if (n == "sub_batch") *value = opt.sub_batch;
    else if (n == "overlap_pass") *value = opt.overlap_pass;
    else if (n == "pose_groups") *value = opt.pose_groups;
    else { set_error("unknown option %s", name); return PR_ERR_INVALID; }
    return PR_OK;
}

// work counters of the kd-tree search kernel (option "nn_count"): out[pass * 8 + k], k = queries, settled by the pixel window,
// handed to the tree, pyramid descents, tree nodes visited, leaves scanned, leaf points tested, window cells read; reading resets them
int pr_nn_counters(uint64_t *out, uint32_t passes)
{
    PR_ENTER();
    if (!out || passes == 0 || passes > kCounterPasses) { set_error("pr_nn_counters: bad arguments"); return PR_ERR_INVALID; }
    std::memset(out, 0, sizeof(uint64_t) * 8 * passes);
    if (!g->nn_counters.p) return PR_OK;
    HIP_TRY(hipStreamSynchronize(g->stream));
    HIP_TRY(hipMemcpy(out, g->nn_counters.p, sizeof(uint64_t) * 8 * passes, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemset(g->nn_counters.p, 0, sizeof(unsigned long long) * 8 * kCounterPasses));
    return PR_OK;
}

int pr_profile_reset(void)
{
    PR_TRY(bind_default());
    std::lock_guard<std::mutex> lk(g->mu);
    g->icp_ms = g->render_ms = g->cloud_ms = 0; g->icp_launches = g->icp_points = g->icp_bytes = 0; g->sample_clock = 0;
    g->gather_ms = 0; g->gather_n = 0; g->gather_ev_used = 0;
    g->icp_launch_us.clear();
    for (double &v : g->nn_part_ms) v = 0;
    g->nn_part_n = 0;
    return PR_OK;
}
int pr_stats(uint64_t *batches_repeated, uint64_t *timings_dropped)
{
    PR_TRY(bind_default());
    std::lock_guard<std::mutex> lk(g->mu);
    if (batches_repeated) *batches_repeated = g->stat_repeated;
    if (timings_dropped) *timings_dropped = g->stat_timing_dropped;
    return PR_OK;
}
int pr_profile_launches(float *launch_us, uint32_t capacity, uint32_t *n)
{
    PR_TRY(bind_default());
    std::lock_guard<std::mutex> lk(g->mu);
    const uint32_t have = (uint32_t)g->icp_launch_us.size();
    if (n) *n = have;
    if (launch_us) std::memcpy(launch_us, g->icp_launch_us.data(), sizeof(float) * std::min(have, capacity));
    return PR_OK;
}
int pr_profile_nn(double part_ms[4], uint64_t *passes)
{
    PR_TRY(bind_default());
    std::lock_guard<std::mutex> lk(g->mu);
    if (part_ms) for (int k = 0; k < 4; ++k) part_ms[k] = g->nn_part_ms[k];
    if (passes) *passes = g->nn_part_n;
    return PR_OK;
}
int pr_profile_read(double *kernel_ms, uint64_t *launches, uint64_t *points, uint64_t *algorithmic_bytes, double *render_ms, double *cloud_ms)
{
    PR_TRY(bind_default());
    std::lock_guard<std::mutex> lk(g->mu);
    if (kernel_ms) *kernel_ms = g->icp_ms;
    if (launches) *launches = g->icp_launches;
    if (points) *points = g->icp_points;
    if (algorithmic_bytes) *algorithmic_bytes = g->icp_bytes;
    if (render_ms) *render_ms = g->render_ms;
    if (cloud_ms) *cloud_ms = g->cloud_ms;
    return PR_OK;
}

}  // extern "C"
