// loopback_rccl.cpp -- TEST INFRASTRUCTURE, not part of the product: a stand-in for librccl that matches ncclSend / ncclRecv between the
// ranks of ONE process through a shared table and moves the bytes with hipMemcpyAsync.  It exists so that the N > 1 branch of
// pr_gather_results (csrc/pr_comm.cpp: grouped send / receive with per-rank offsets and counts) can execute on a box with a single GPU --
// real RCCL refuses two ranks on one device -- before the first run on an 8-GPU node.  Selected with PR_RCCL_LIBRARY=<this .so>
// (read by rccl_load()); never loaded otherwise.  Only the nine entry points the library binds are here.
//
//   hipcc -shared -fPIC -O2 tests/rccl_loopback/loopback_rccl.cpp -o tests/rccl_loopback/librccl_loopback.so
//
// Semantics kept from NCCL: ncclCommInitRank blocks until `world` ranks have joined the id; operations between ncclGroupStart and
// ncclGroupEnd are issued together at ncclGroupEnd; a send from rank s to rank d pairs with the d-side receive from s in posting order;
// a transfer is ordered behind everything enqueued before it on BOTH streams, and the sender's stream continues only when the bytes have
// left its buffer.  UNLIKE RCCL, ncclGroupEnd BLOCKS ON THE HOST until the peers of its operations have issued theirs (the copy needs both addresses):
// a rank must not wait for another rank's context (pr_free does) between its own entry into a gather and the root's.  A size mismatch between a send and its receive is an error (ncclInvalidArgument), as a truncated receive would be.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <condition_variable>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

namespace {

struct Post { const void *src; size_t bytes; hipEvent_t ready; hipEvent_t taken; bool consumed = false, failed = false; int src_device; };
struct Group {
    std::mutex mu;
    std::condition_variable cv;
    int world = 0, joined = 0, left = 0;
    std::map<std::pair<int, int>, std::deque<std::shared_ptr<Post>>> box;      // (from, to) -> sends posted, not yet received
};
struct Comm { std::shared_ptr<Group> grp; int rank = 0, world = 1, device = 0; };
struct Op { bool send; const void *sbuf; void *rbuf; size_t bytes; int peer; Comm *comm; hipStream_t stream; };

std::mutex g_mu;
std::map<unsigned long long, std::shared_ptr<Group>> g_groups;
unsigned long long g_next_id = 1;
thread_local int tl_depth = 0;
thread_local std::vector<Op> tl_ops;

size_t type_bytes(ncclDataType_t t)
{
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 1;
    }
}

ncclResult_t flush()
{
    std::vector<Op> ops;
    ops.swap(tl_ops);
    std::vector<std::shared_ptr<Post>> mine;
    // 1. post every send (never blocks)
    for (const Op &o : ops) {
        if (!o.send) continue;
        auto p = std::make_shared<Post>();
        p->src = o.sbuf; p->bytes = o.bytes; p->src_device = o.comm->device;
        if (hipSetDevice(o.comm->device) != hipSuccess) return ncclUnhandledCudaError;
        if (hipEventCreateWithFlags(&p->ready, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&p->taken, hipEventDisableTiming) != hipSuccess) return ncclUnhandledCudaError;
        if (hipEventRecord(p->ready, o.stream) != hipSuccess) return ncclUnhandledCudaError;
        { std::lock_guard<std::mutex> lk(o.comm->grp->mu); o.comm->grp->box[{ o.comm->rank, o.peer }].push_back(p); }
        o.comm->grp->cv.notify_all();
        mine.push_back(p);
    }
    // 2. every receive: wait for its send to be posted, order the copy behind the sender's stream, enqueue it on the receiver's
    ncclResult_t rc = ncclSuccess;
    for (const Op &o : ops) {
        if (o.send) continue;
        std::shared_ptr<Post> p;
        {
            std::unique_lock<std::mutex> lk(o.comm->grp->mu);
            auto &q = o.comm->grp->box[{ o.peer, o.comm->rank }];
            o.comm->grp->cv.wait(lk, [&] { return !q.empty(); });
            p = q.front(); q.pop_front();
        }
        bool ok = p->bytes == o.bytes;
        if (ok) ok = hipSetDevice(o.comm->device) == hipSuccess && hipStreamWaitEvent(o.stream, p->ready, 0) == hipSuccess &&
                     hipMemcpyAsync(o.rbuf, p->src, o.bytes, hipMemcpyDefault, o.stream) == hipSuccess && hipEventRecord(p->taken, o.stream) == hipSuccess;
        { std::lock_guard<std::mutex> lk(o.comm->grp->mu); p->consumed = true; p->failed = !ok; }
        o.comm->grp->cv.notify_all();
        if (!ok) rc = (p->bytes == o.bytes) ? ncclUnhandledCudaError : ncclInvalidArgument;
    }
    // 3. every send of this call: its stream goes on once the receiver's copy has run
    for (size_t i = 0, k = 0; i < ops.size(); ++i) {
        const Op &o = ops[i];
        if (!o.send) continue;
        std::shared_ptr<Post> p = mine[k++];
        { std::unique_lock<std::mutex> lk(o.comm->grp->mu); o.comm->grp->cv.wait(lk, [&] { return p->consumed; }); }
        if (p->failed) rc = (rc == ncclSuccess) ? ncclInvalidArgument : rc;
        else if (hipSetDevice(o.comm->device) != hipSuccess || hipStreamWaitEvent(o.stream, p->taken, 0) != hipSuccess) rc = ncclUnhandledCudaError;
        // (the events are left to the process: they may still be referenced by queued waits)
    }
    return rc;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    if (!id) return ncclInvalidArgument;
    std::lock_guard<std::mutex> lk(g_mu);
    std::memset(id->internal, 0, sizeof id->internal);
    const unsigned long long v = g_next_id++;
    std::memcpy(id->internal, "PRLOOPBK", 8);
    std::memcpy(id->internal + 8, &v, sizeof v);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks || std::memcmp(id.internal, "PRLOOPBK", 8) != 0) return ncclInvalidArgument;
    unsigned long long v = 0;
    std::memcpy(&v, id.internal + 8, sizeof v);
    std::shared_ptr<Group> grp;
    { std::lock_guard<std::mutex> lk(g_mu); auto &slot = g_groups[v]; if (!slot) { slot = std::make_shared<Group>(); slot->world = nranks; } grp = slot; }
    if (grp->world != nranks) return ncclInvalidArgument;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return ncclUnhandledCudaError;
    { std::unique_lock<std::mutex> lk(grp->mu); grp->joined++; grp->cv.notify_all(); grp->cv.wait(lk, [&] { return grp->joined >= grp->world; }); }
    Comm *c = new Comm{ grp, rank, nranks, dev };
    *comm = reinterpret_cast<ncclComm_t>(c);
    return ncclSuccess;
}

ncclResult_t ncclCommInitAll(ncclComm_t *comms, int ndev, const int *devlist)
{
    if (!comms || ndev < 1) return ncclInvalidArgument;
    auto grp = std::make_shared<Group>();
    grp->world = ndev; grp->joined = ndev;
    for (int i = 0; i < ndev; ++i) comms[i] = reinterpret_cast<ncclComm_t>(new Comm{ grp, i, ndev, devlist ? devlist[i] : i });
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    delete reinterpret_cast<Comm *>(comm);
    return ncclSuccess;
}

ncclResult_t ncclGroupStart(void) { ++tl_depth; return ncclSuccess; }

ncclResult_t ncclGroupEnd(void)
{
    if (tl_depth <= 0) return ncclInvalidUsage;
    if (--tl_depth > 0) return ncclSuccess;
    return flush();
}

ncclResult_t ncclSend(const void *sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream)
{
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (!c || peer < 0 || peer >= c->world || (!sendbuff && count)) return ncclInvalidArgument;
    tl_ops.push_back(Op{ true, sendbuff, nullptr, count * type_bytes(datatype), peer, c, stream });
    return tl_depth > 0 ? ncclSuccess : flush();
}

ncclResult_t ncclRecv(void *recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream)
{
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (!c || peer < 0 || peer >= c->world || (!recvbuff && count)) return ncclInvalidArgument;
    tl_ops.push_back(Op{ false, nullptr, recvbuff, count * type_bytes(datatype), peer, c, stream });
    return tl_depth > 0 ? ncclSuccess : flush();
}

const char *ncclGetErrorString(ncclResult_t r)
{
    switch (r) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "loopback stand-in: a HIP call failed";
    case ncclInvalidArgument: return "loopback stand-in: invalid argument (or a send / receive size mismatch)";
    case ncclInvalidUsage: return "loopback stand-in: invalid usage";
    default: return "loopback stand-in: error";
    }
}

}  // extern "C"
