// pr_comm.cpp -- the job's one collective: RCCL opened on first use, communicators per context, grouped send / receive of the result records
#include "pr_runtime.h"

namespace prr {

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
    // PR_RCCL_LIBRARY=<path>: bind the nine entry points from that library instead.  Test use: tests/rccl_loopback (ranks of ONE process matched
    // through a table, bytes moved by hipMemcpyAsync) lets the N > 1 branch of pr_gather_results run on a box with one GPU, where RCCL itself
    // refuses two ranks on a device.  A path that cannot be opened is an error, never a silent fall-back to the real library.
    const char *forced = std::getenv("PR_RCCL_LIBRARY");
    if (forced && *forced) {
        h = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
        if (!h) { set_error("PR_RCCL_LIBRARY=%s cannot be opened (%s)", forced, dlerror()); return PR_ERR_COMM; }
    }
    else for (const char *name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" }) { h = dlopen(name, RTLD_NOW | RTLD_LOCAL); if (h) break; }
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

}  // namespace prr

using namespace prr;

extern "C" {

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
        if (n_local) note_write(recv_dev, sizeof(pr_result) * n_local);
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
        note_write(recv_dev, sizeof(pr_result) * n_total);
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

}  // extern "C"
