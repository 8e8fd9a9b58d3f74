// loopback_rccl.cpp -- TEST DOUBLE for the nine RCCL entry points libarctic_hip.so resolves (csrc/renderer.cpp: struct Rccl), so that
// the R > 1 branch of the exchange (arctic_comm_init's layout all-gather, arctic_gather_frame's grouped send / recv, the sharded shadow
// map's in-place all-gather) can be EXECUTED on one GPU: RCCL itself refuses two ranks on one device, and the pool hands out one GPU.
// Ranks = threads of one process, each driving its own handle; a "communicator" is a rendezvous in host memory, a transfer a
// hipMemcpyAsync on the receiver's stream, ordered against the sender's stream by events -- the same stream semantics NCCL gives
// (a collective is enqueued on every rank's stream and runs when all of them have reached it; buffers may be reused by whatever the
// stream runs next).  Loaded only through ARCTIC_RCCL_LIB (honoured only when set); never part of the product.
//   g++ -std=c++17 -O1 -shared -fPIC -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tests/cpp/loopback_rccl.cpp -o tests/cpp/libloopback_rccl.so -L/opt/rocm/lib -lamdhip64
#include <hip/hip_runtime.h>
#include <condition_variable>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace {

constexpr int ID_BYTES = 128;
struct Entry { int peer; char *ptr; size_t bytes; bool send; };   // peer -1: every rank (all-gather)
struct Group {
    int world = 0;
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0;
    long generation = 0;
    struct Post { std::vector<Entry> entries; hipEvent_t ready = nullptr, done = nullptr; };
    std::vector<Post> post;
    void barrier() {
        std::unique_lock<std::mutex> lock(m);
        const long g = generation;
        if (++arrived == world) { arrived = 0; ++generation; cv.notify_all(); }
        else cv.wait(lock, [&] { return generation != g; });
    }
};
struct Comm { std::shared_ptr<Group> g; int rank = 0; };

std::mutex g_registry_mutex;
std::map<std::string, std::shared_ptr<Group>> g_registry;
unsigned long long g_next_id = 1;

size_t type_bytes(int t) { static const size_t s[] = {1, 1, 4, 4, 8, 8, 2, 4, 8, 2}; return t >= 0 && t < 10 ? s[t] : 0; }

// one exchange step of rank `c`: publish what I send and what I want, meet, copy what I want on MY stream behind the senders'
// streams, meet, let my stream wait for whoever read from me
int exchange(Comm *c, std::vector<Entry> entries, hipStream_t stream) {
    Group &g = *c->g;
    Group::Post &me = g.post[(size_t)c->rank];
    if (hipEventRecord(me.ready, stream) != hipSuccess) return 1;
    me.entries = std::move(entries);
    g.barrier();
    std::vector<size_t> cursor((size_t)g.world, 0);   // per sender: the next of its entries addressed to me
    for (const Entry &want : me.entries) {
        if (want.send) continue;
        Group::Post &from = g.post[(size_t)want.peer];
        const Entry *match = nullptr;
        for (size_t &i = cursor[(size_t)want.peer]; i < from.entries.size();) {
            const Entry &e = from.entries[i++];
            if (e.send && (e.peer == c->rank || e.peer == -1)) { match = &e; break; }
        }
        if (!match || match->bytes != want.bytes) return 2;   // an unmatched or mis-sized transfer: what would hang or corrupt on real RCCL
        if (want.peer != c->rank && hipStreamWaitEvent(stream, from.ready, 0) != hipSuccess) return 1;
        if (match->ptr != want.ptr && hipMemcpyAsync(want.ptr, match->ptr, want.bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess) return 1;
        if (match->peer == -1) cursor[(size_t)want.peer] = 0;   // an all-gather entry serves every receiver
    }
    if (hipEventRecord(me.done, stream) != hipSuccess) return 1;
    g.barrier();
    for (int p = 0; p < g.world; ++p)
        if (p != c->rank && hipStreamWaitEvent(stream, g.post[(size_t)p].done, 0) != hipSuccess) return 1;
    g.barrier();   // nobody publishes the next step's entries while a peer still reads this one's
    return 0;
}

thread_local int t_group_depth = 0;
thread_local std::vector<Entry> t_group_entries;
thread_local Comm *t_group_comm = nullptr;
thread_local hipStream_t t_group_stream = nullptr;

}  // namespace

extern "C" {

struct ncclUniqueId { char internal[ID_BYTES]; };

int ncclGetUniqueId(ncclUniqueId *id) {
    if (!id) return 4;
    std::lock_guard<std::mutex> lock(g_registry_mutex);
    std::memset(id->internal, 0, ID_BYTES);
    std::snprintf(id->internal, ID_BYTES, "arctic-loopback-%llu", g_next_id++);
    return 0;
}

int ncclCommInitRank(void **comm, int world, ncclUniqueId id, int rank) {
    if (!comm || world < 1 || rank < 0 || rank >= world) return 4;
    std::shared_ptr<Group> g;
    {
        std::lock_guard<std::mutex> lock(g_registry_mutex);
        std::shared_ptr<Group> &slot = g_registry[std::string(id.internal, ID_BYTES)];
        if (!slot) { slot = std::make_shared<Group>(); slot->world = world; slot->post.resize((size_t)world); }
        if (slot->world != world) return 4;
        g = slot;
    }
    Group::Post &me = g->post[(size_t)rank];
    if (hipEventCreateWithFlags(&me.ready, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&me.done, hipEventDisableTiming) != hipSuccess) return 1;
    Comm *c = new Comm;
    c->g = g; c->rank = rank;
    *comm = c;
    g->barrier();   // like the real call: returns when every rank has joined
    return 0;
}

int ncclCommDestroy(void *comm) {
    Comm *c = static_cast<Comm *>(comm);
    if (!c) return 4;
    Group::Post &me = c->g->post[(size_t)c->rank];
    if (me.ready) (void)hipEventDestroy(me.ready);
    if (me.done) (void)hipEventDestroy(me.done);
    me.ready = me.done = nullptr;
    delete c;
    return 0;
}

int ncclAllGather(const void *sendbuf, void *recvbuf, size_t count, int type, void *comm, hipStream_t stream) {
    Comm *c = static_cast<Comm *>(comm);
    const size_t bytes = count * type_bytes(type);
    if (!c || !bytes) return 4;
    std::vector<Entry> e;
    e.push_back({-1, const_cast<char *>(static_cast<const char *>(sendbuf)), bytes, true});
    for (int p = 0; p < c->g->world; ++p) e.push_back({p, static_cast<char *>(recvbuf) + (size_t)p * bytes, bytes, false});
    return exchange(c, std::move(e), stream);
}

static int post(Comm *c, Entry e, hipStream_t stream) {
    if (!c || t_group_depth == 0) return 4;   // the library only posts inside a group
    if (t_group_comm && (t_group_comm != c || t_group_stream != stream)) return 4;
    t_group_comm = c; t_group_stream = stream;
    t_group_entries.push_back(e);
    return 0;
}
int ncclSend(const void *buf, size_t count, int type, int peer, void *comm, hipStream_t stream) {
    return post(static_cast<Comm *>(comm), {peer, const_cast<char *>(static_cast<const char *>(buf)), count * type_bytes(type), true}, stream);
}
int ncclRecv(void *buf, size_t count, int type, int peer, void *comm, hipStream_t stream) {
    return post(static_cast<Comm *>(comm), {peer, static_cast<char *>(buf), count * type_bytes(type), false}, stream);
}
int ncclGroupStart() { ++t_group_depth; return 0; }
int ncclGroupEnd() {
    if (t_group_depth == 0) return 4;
    if (--t_group_depth) return 0;
    int rc = 0;
    if (t_group_comm) rc = exchange(t_group_comm, std::move(t_group_entries), t_group_stream);
    t_group_entries.clear(); t_group_comm = nullptr; t_group_stream = nullptr;
    return rc;
}
const char *ncclGetErrorString(int rc) {
    switch (rc) {
    case 0: return "loopback: success";
    case 1: return "loopback: a HIP call failed";
    case 2: return "loopback: a receive without a matching send of the same size";
    default: return "loopback: invalid argument";
    }
}

}  // extern "C"
