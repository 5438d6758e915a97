// pr_context.cpp -- contexts and their registry, memory entry points, options, profiling read-outs (include/pose_refine.h)
#include "pr_runtime.h"

namespace prh {
thread_local std::string g_err;
void set_error(const char *fmt, ...)
{
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    g_err = buf;
}
}  // namespace prh

namespace prr {

static std::mutex &g_reg_mu_trace() { static std::mutex m; return m; }

// ---- PR_TRACE: a ring of (thread, tag, nanoseconds), written to the named file at exit ---------------------------------------------
namespace {
struct TraceRec { uint64_t ns; const char *tag; uint32_t tid; };
constexpr size_t kTraceCap = 1u << 18;
TraceRec *g_trace = nullptr;
std::atomic<uint64_t> g_trace_n{ 0 };
std::atomic<int> g_trace_state{ 0 };                              // 0 unknown, 1 off, 2 on
std::string g_trace_path;
void trace_dump()
{
    FILE *f = std::fopen(g_trace_path.c_str(), "w");
    if (!f) return;
    const uint64_t n = std::min<uint64_t>(g_trace_n.load(), kTraceCap);
    for (uint64_t i = 0; i < n; ++i) std::fprintf(f, "%llu %u %s\n", (unsigned long long)g_trace[i].ns, g_trace[i].tid, g_trace[i].tag);
    std::fclose(f);
}
}  // namespace
void trace_mark(const char *tag)
{
    int st = g_trace_state.load(std::memory_order_acquire);
    if (st == 1) return;
    if (st == 0) {
        std::lock_guard<std::mutex> lk(g_reg_mu_trace());
        st = g_trace_state.load();
        if (st == 0) {
            const char *p = std::getenv("PR_TRACE");
            if (p && *p) { g_trace_path = p; g_trace = new TraceRec[kTraceCap]; std::atexit(trace_dump); st = 2; } else st = 1;
            g_trace_state.store(st, std::memory_order_release);
        }
        if (st == 1) return;
    }
    timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    static thread_local uint32_t tid = (uint32_t)std::hash<std::thread::id>()(std::this_thread::get_id());
    const uint64_t i = g_trace_n.fetch_add(1);
    if (i < kTraceCap) g_trace[i] = TraceRec{ (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec, tag, tid };
}

WriteLog g_writes;
std::atomic<uint64_t> g_flag_overtook{ 0 };
Options opt;
std::mutex g_reg_mu;
std::vector<Ctx *> g_shared;         // index = device ordinal
int g_default_device = -1;
thread_local Ctx *g = nullptr;
thread_local PrivateCtx tl_private;
std::mutex g_private_mu;
std::vector<Ctx *> g_private;

void ctx_teardown(Ctx *c)        // c->mu held (or c unreachable); the calling thread is bound to c
{
    if (!c->ready) return;
    (void)hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    for (Slot &sl : c->slots) slot_release(sl);
    for (DevBuf *b : { &c->aabb, &c->aabb_keys, &c->bbox, &c->poses, &c->depth, &c->row_count, &c->row_off, &c->counts, &c->cloud, &c->meta, &c->partial,
                       &c->sums, &c->packed.rec, &c->nn_prev, &c->dstate, &c->dresults, &c->arrive, &c->conv16, &c->conv8, &c->kd_scratch, &c->kd_tmp, &c->nn_full, &c->gather_tmp, &c->nn_counters }) b->release();
    for (NNDerived &d : c->nn_sets) d.release();
    for (PinBuf *b : { &c->h_sums, &c->h_meta, &c->h_counts, &c->h_results, &c->h_dstate, &c->h_poses, &c->h_flags }) b->release();
    c->packed = PackedCache();
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
    private_wait_unpinned(c);
    delete c; c = nullptr;
}

}  // namespace prr

using namespace prr;

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
    private_wait_unpinned(tl_private.c);
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
        auto drain_locked = [](Ctx *c) {
            std::lock_guard<std::mutex> lk(c->mu);
            if (!c->ready) return;
            if (c->stream) (void)hipStreamSynchronize(c->stream);
            for (hipStream_t sd : c->side) if (sd) (void)hipStreamSynchronize(sd);
            for (Slot &sl : c->slots) slot_drain(sl);
        };
        std::vector<Ctx *> shared;
        { std::lock_guard<std::mutex> lk(g_reg_mu); for (Ctx *c : g_shared) if (c && c != self && c->device == self->device) shared.push_back(c); }
        for (Ctx *c : shared) drain_locked(c);                       // (shared contexts are never deleted)
        // private contexts: pinned ONE AT A TIME, just before its mutex is taken, and unpinned before the next is looked at -- this thread never
        // holds a pin on one context while it blocks on another's mutex (with all of them pinned up front, a thread tearing its
        // context down joins a slot's helper thread, whose exit spins until ITS context is unpinned, while this thread waits for the first one's
        // mutex: a three-way wait).  A context is unregistered before it is torn down, so one that is found here is alive until the pin goes.
        std::vector<Ctx *> seen;
        for (;;) {
            Ctx *c = nullptr;
            {
                std::lock_guard<std::mutex> plk(g_private_mu);
                for (Ctx *p : g_private)
                    if (p != self && p->device == self->device && std::find(seen.begin(), seen.end(), p) == seen.end()) { c = p; c->pins.fetch_add(1, std::memory_order_acq_rel); break; }
            }
            if (!c) break;
            seen.push_back(c);
            drain_locked(c);
            c->pins.fetch_sub(1, std::memory_order_acq_rel);
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
        note_write(dst, bytes);
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
    note_write(dev_dst, count * sizeof(int32_t));
    HIP_TRY(prk::launch_fill_i32(dev_dst, count, value, g->stream));
    HIP_TRY(hipStreamSynchronize(g->stream));
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
    else if (n == "host_poll") opt.host_poll = value ? 1 : 0;
    else if (n == "graph") opt.use_graph = value ? 1 : 0;
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
    else if (n == "host_poll") *value = opt.host_poll;
    else if (n == "stat_flag_overtook") *value = (int)std::min<uint64_t>(g_flag_overtook.load(), 0x7fffffffull);   // read-only
    else if (n == "raster_mode") *value = opt.raster_mode;
    else if (n == "eager_streams") *value = opt.eager_streams;
    else if (n == "graph") *value = opt.use_graph;
    else if (n == "fused_solve") *value = opt.fused_solve;
    else if (n == "sub_batch") *value = opt.sub_batch;
    else if (n == "overlap_pass") *value = opt.overlap_pass;
    else if (n == "pose_groups") *value = opt.pose_groups;
    else { set_error("unknown option %s", name); return PR_ERR_INVALID; }
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
