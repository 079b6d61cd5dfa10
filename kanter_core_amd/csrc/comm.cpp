// The exchange step of a multi-GPU evaluation, behind the C ABI: slots that cross a rank boundary of a partition plan
// (partition.cpp) and the finished row bands of a band plan move between the processes of ONE node.  One process per GPU;
// every process holds the same graph and works through the same list of transfers in the same order.
//
// The reference has no distributed layer; what it has is the readiness rule that makes one possible -- a node needs nothing
// but its parents' slot data (src/engine.rs:213-275) -- and kc_live_graph_import_slot_data is the receiving end of that rule.
//
// Three parts:
//
//   * the MAILBOX: a POSIX shared-memory segment (named inside the communicator id) with one single-producer / single-consumer
//     ring of messages per ordered pair of ranks.  A message is what a receiver has to know before it can take a slot: size,
//     which planes are constants (Mix's alpha = 1 travels as a scalar, not as 64 MiB of ones) or aliases of each other, and
//     -- for the IPC wire -- where the planes are.  Host to host, no GPU involved: a receiver waiting for a message waits for the
//     producer's HOST to have enqueued its kernels, never for plane data.
//
//   * the WIRE, one of two (named in the communicator id, so every rank uses the same):
//       "ipc"  (default)  the receiver PULLS: it maps the sender's planes (hipIpcGetMemHandle / hipIpcOpenMemHandle, mappings
//              cached), makes one stream per peer wait -- on the device, hipStreamWaitValue64 -- for a counter in the shared
//              segment that the sender's stream writes behind the kernels that produce the planes (hipStreamWriteValue64), copies
//              (hipMemcpyAsync: the DMA engines over xGMI on a multi-GPU node) and writes an acknowledgement counter the same
//              way, which is what lets the sender drop its reference.  One stream per peer: the inbound transfers of a fan-in
//              run at the same time, each over its own link.  Works between processes that share one GPU as well, which is how
//              the N > 1 path is tested on a one-GPU box (tests/test_gpu_comm_ranks.py).
//       "rccl" ncclSend / ncclRecv (librccl bound with dlopen at first use), all transfers of one level in ONE group on one
//              communication stream behind an event of the compute stream.  RCCL refuses two ranks on one device; a world of
//              one rank can send to itself.
//     No host thread waits for plane data with either.
//
//   * the EXCHANGE: the transfer list is worked through level by level (kc_transfer::level: transfers of one level do not
//     depend on each other).  Per level a rank (1) evaluates what it sends (kernels enqueued, nobody waits), describes it and
//     posts the messages, (2) takes the messages of what it receives and allocates planes, (3) runs the wire, (4) hands the
//     received slots over as kc_live_graph_import_slot_data does, the compute stream waiting for the transfers on the device.
//
// Failure: any error on any rank sets the segment's abort flag; every host-side wait on every rank then fails instead of
// hanging (host-side waits also time out, KC_COMM_TIMEOUT_S, default 120 s), an open RCCL group is always closed, and nothing a
// failed call created stays referenced.  The communicator is unusable afterwards (kc_comm_destroy, kc_comm_init again).
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cstring>
#include <random>
#include <thread>

#include "kc_runtime.hpp"

namespace kc {
namespace {

// ------------------------------------------------------------------------------------------ RCCL, bound at first use
struct NcclId {
    char internal[128];
};
typedef void *NcclComm;
enum { NCCL_UINT8 = 1, NCCL_FLOAT = 7 };

struct Rccl {
    void *lib = nullptr;
    int (*get_unique_id)(NcclId *) = nullptr;
    int (*comm_init_rank)(NcclComm *, int, NcclId, int) = nullptr;
    int (*comm_destroy)(NcclComm) = nullptr;
    int (*send)(const void *, size_t, int, int, NcclComm, hipStream_t) = nullptr;
    int (*recv)(void *, size_t, int, int, NcclComm, hipStream_t) = nullptr;
    int (*group_start)() = nullptr;
    int (*group_end)() = nullptr;
    const char *(*error_string)(int) = nullptr;
    bool ok = false;
};

Rccl &rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : { "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so" }) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.lib) break;
        }
        if (!r.lib) return;
#define KC_NCCL_SYM(field, sym) *(void **)(&r.field) = dlsym(r.lib, sym)
        KC_NCCL_SYM(get_unique_id, "ncclGetUniqueId");
        KC_NCCL_SYM(comm_init_rank, "ncclCommInitRank");
        KC_NCCL_SYM(comm_destroy, "ncclCommDestroy");
        KC_NCCL_SYM(send, "ncclSend");
        KC_NCCL_SYM(recv, "ncclRecv");
        KC_NCCL_SYM(group_start, "ncclGroupStart");
        KC_NCCL_SYM(group_end, "ncclGroupEnd");
        KC_NCCL_SYM(error_string, "ncclGetErrorString");
#undef KC_NCCL_SYM
        r.ok = r.get_unique_id && r.comm_init_rank && r.comm_destroy && r.send && r.recv && r.group_start && r.group_end && r.error_string;
    });
    return r;
}

int nccl_fail(int rc, const char *what)
{
    set_error(std::string("RCCL: ") + what + ": " + (rccl().error_string ? rccl().error_string(rc) : "error"));
    return KC_ERR_GENERIC;
}

// An RCCL group that is closed on every way out of the scope that opened it.
struct NcclGroup {
    bool open = false;
    int start()
    {
        const int rc = rccl().group_start();
        if (rc != 0) return nccl_fail(rc, "ncclGroupStart");
        open = true;
        return KC_OK;
    }
    int end()
    {
        if (!open) return KC_OK;
        open = false;
        const int rc = rccl().group_end();
        return rc != 0 ? nccl_fail(rc, "ncclGroupEnd") : KC_OK;
    }
    ~NcclGroup()
    {
        if (open) (void)rccl().group_end();
    }
};

// ------------------------------------------------------------------------------------------ the shared segment
enum { WIRE_IPC = 0, WIRE_RCCL = 1 };
constexpr int kMaxWorld = 16;
constexpr int kRing = 64;  // messages in flight per ordered pair of ranks (a level with more slots between two ranks than this would stall)
constexpr uint32_t kShmMagic = 0x4b434d42u;  // "KCMB"
constexpr uint32_t kMsgSlot = 0x4b43534cu;   // "KCSL"
constexpr uint32_t kMsgBand = 0x4b434244u;   // "KCBD"

struct WireMem {
    hipIpcMemHandle_t handle;  // of the allocation the plane lives in
    uint64_t offset, bytes;    // the plane's whole pitched buffer inside it
    uint64_t local;            // the sender's own address (a rank sending to itself reads it directly)
    uint64_t pad;
};
static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");

struct Msg {
    uint32_t magic, node, slot;
    uint32_t w, h, n_planes;  // n_planes: 1 = Gray, 4 = Rgba; h: the rows that travel
    uint32_t kind[4];         // 0 = the idx-th plane on the wire, 1 = broadcast constant cval
    uint32_t idx[4];
    float cval[4];
    uint32_t n_mem;
    int32_t y0;               // kMsgBand: first row of the band in the full image
    uint32_t full_h, order;   // order: position in the transfer list (set by the receiver too): sequences the rccl group
    uint64_t produced;        // ipc: the sender's `produced` counter has this value once the planes are complete
    WireMem mem[4];
};

struct alignas(64) Ring {  // [src][dst]: written by src, read by dst
    std::atomic<uint64_t> head;  // messages posted
    char pad0[56];
    std::atomic<uint64_t> tail;  // messages taken
    char pad1[56];
    Msg msg[kRing];
};

// Written by STREAMS (hipStreamWriteValue64), waited for by streams of other processes and read by hosts.
struct alignas(64) RankCell {
    uint64_t produced;           // this rank's sends whose planes are complete
    char pad[56];
    uint64_t copied[kMaxWorld];  // copied[s]: messages from rank s whose planes this rank has finished copying
    std::atomic<uint64_t> epoch; // bumped when this rank gives blocks back to the driver (kc_pool_trim): peers drop their mappings
    char pad2[56];
};

struct Shm {
    std::atomic<uint32_t> magic;
    uint32_t world, wire;
    std::atomic<uint32_t> attached, abort;
    alignas(4096) RankCell cell[kMaxWorld];
    alignas(4096) Ring ring[kMaxWorld][kMaxWorld];
};

// The communicator id (KC_COMM_ID_BYTES = 256): [0, 128) ncclUniqueId (rccl wire), [128, 192) segment name, [192] wire.
constexpr size_t kIdNameAt = 128, kIdWireAt = 192;

struct PendingSend {
    hipEvent_t done = nullptr;  // rccl: fires when the send has left
    std::vector<std::pair<int, uint64_t>> acks;  // ipc: (dst, copied[me] on dst must reach this)
    kc_image *img = nullptr;    // retained until then
    std::vector<kc_plane *> extra;  // dense copies made for the wire
};

struct Opened {
    void *ptr = nullptr;
    int src = 0;
    uint64_t last_use = 0;
};

struct Comm {
    bool active = false;
    int rank = 0, world = 1, wire = WIRE_IPC;
    Shm *shm = nullptr;
    std::string shm_name;
    bool registered = false;
    NcclComm nccl = nullptr;
    hipStream_t flag_stream = nullptr, data_stream = nullptr;
    hipStream_t in_stream[kMaxWorld] = {};
    hipEvent_t in_ev[kMaxWorld] = {};
    hipEvent_t compute_ev = nullptr, recv_ev = nullptr;
    uint64_t produced_seq = 0;
    uint64_t sent_to[kMaxWorld] = {}, recv_from[kMaxWorld] = {};
    uint64_t peer_epoch[kMaxWorld] = {};
    std::deque<PendingSend> pending;
    std::unordered_map<void *, hipIpcMemHandle_t> exported;
    std::map<std::array<char, 64>, Opened> opened;
    uint64_t open_clock = 0;
    double timeout_s = 120.0;
    uint64_t planes_sent = 0, planes_received = 0, bytes_sent = 0;
};

Comm &comm()
{
    static Comm c;
    return c;
}

size_t natural_pitch(uint32_t w) { return ((size_t)w * 4 + 255) / 256 * 256; }  // what plane_new_mem gives a plane of this width

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// Host-side wait: polls `ready` until it holds, a peer has failed, or the time is up.
template <class F> int wait_for(Comm &cm, const char *what, F ready)
{
    const double t0 = now_s();
    for (unsigned spins = 0;; ++spins) {
        if (ready()) return KC_OK;
        if (cm.shm && cm.shm->abort.load(std::memory_order_acquire)) {
            set_error(std::string("exchange: another rank reported a failure while this one was waiting for ") + what);
            return KC_ERR_GENERIC;
        }
        if (spins > 2000) {
            if (now_s() - t0 > cm.timeout_s) {
                set_error(std::string("exchange: timed out waiting for ") + what);
                if (cm.shm) cm.shm->abort.store(1, std::memory_order_release);
                return KC_ERR_GENERIC;
            }
            std::this_thread::sleep_for(std::chrono::microseconds(spins > 20000 ? 200 : 20));
        } else {
            std::this_thread::yield();
        }
    }
}

void *dev_addr(const void *host_field)
{
    void *d = nullptr;
    if (hipHostGetDevicePointer(&d, const_cast<void *>(host_field), 0) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return d;
}

uint64_t host_read(const uint64_t *p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }

// Drops the references of sends that have arrived (all of them when `wait`).
int reap(Comm &cm, bool wait)
{
    int status = KC_OK;
    while (!cm.pending.empty()) {
        PendingSend &p = cm.pending.front();
        bool done = true;
        if (p.done) {
            if (wait) (void)hipEventSynchronize(p.done);
            else if (hipEventQuery(p.done) != hipSuccess) {
                (void)hipGetLastError();
                done = false;
            }
        }
        for (auto &a : p.acks) {
            const uint64_t *cell = &cm.shm->cell[a.first].copied[cm.rank];
            if (host_read(cell) >= a.second) continue;
            if (!wait) {
                done = false;
                break;
            }
            const int s = wait_for(cm, "a peer to finish copying a sent slot", [&] { return host_read(cell) >= a.second; });
            if (s != KC_OK) status = s;  // the peer is gone: let go of the planes all the same
        }
        if (!done) break;
        if (p.done) (void)hipEventDestroy(p.done);
        image_release(p.img);
        for (auto *q : p.extra) plane_release(q);
        cm.pending.pop_front();
    }
    return status;
}

void close_mappings_of(Comm &cm, int src)
{
    bool synced = false;
    for (auto it = cm.opened.begin(); it != cm.opened.end();) {
        if (src >= 0 && it->second.src != src) {
            ++it;
            continue;
        }
        if (!synced && cm.in_stream[it->second.src]) (void)hipStreamSynchronize(cm.in_stream[it->second.src]);
        synced = src >= 0;
        (void)hipIpcCloseMemHandle(it->second.ptr);
        it = cm.opened.erase(it);
    }
}

// The sender's plane as this process sees it.
int map_peer(Comm &cm, int src, const WireMem &m, const char **out)
{
    if (src == cm.rank) {
        *out = (const char *)(uintptr_t)m.local;
        return KC_OK;
    }
    const uint64_t ep = cm.shm->cell[src].epoch.load(std::memory_order_acquire);
    if (ep != cm.peer_epoch[src]) {  // the peer has freed blocks: mappings of them would keep the memory alive
        close_mappings_of(cm, src);
        cm.peer_epoch[src] = ep;
    }
    std::array<char, 64> key;
    std::memcpy(key.data(), &m.handle, 64);
    auto it = cm.opened.find(key);
    if (it == cm.opened.end()) {
        if (cm.opened.size() >= 1024) {  // the least recently used mapping goes
            auto lru = cm.opened.begin();
            for (auto j = cm.opened.begin(); j != cm.opened.end(); ++j)
                if (j->second.last_use < lru->second.last_use) lru = j;
            if (cm.in_stream[lru->second.src]) (void)hipStreamSynchronize(cm.in_stream[lru->second.src]);
            (void)hipIpcCloseMemHandle(lru->second.ptr);
            cm.opened.erase(lru);
        }
        void *p = nullptr;
        KC_HIP(hipIpcOpenMemHandle(&p, m.handle, hipIpcMemLazyEnablePeerAccess));
        it = cm.opened.emplace(key, Opened{ p, src, 0 }).first;
        ctx().counters["comm_ipc_mappings_opened"]++;
    }
    it->second.last_use = ++cm.open_clock;
    *out = (const char *)it->second.ptr + m.offset;
    return KC_OK;
}

int post(Comm &cm, int dst, const Msg &m)
{
    Ring &r = cm.shm->ring[cm.rank][dst];
    const uint64_t head = r.head.load(std::memory_order_relaxed);
    KC_TRY(wait_for(cm, "room in a peer's mailbox", [&] { return head - r.tail.load(std::memory_order_acquire) < (uint64_t)kRing; }));
    r.msg[head % kRing] = m;
    r.head.store(head + 1, std::memory_order_release);
    cm.sent_to[dst] = head + 1;
    return KC_OK;
}

int take(Comm &cm, int src, Msg *m)
{
    Ring &r = cm.shm->ring[src][cm.rank];
    const uint64_t tail = r.tail.load(std::memory_order_relaxed);
    KC_TRY(wait_for(cm, "a peer's description of a slot", [&] { return r.head.load(std::memory_order_acquire) > tail; }));
    *m = r.msg[tail % kRing];
    r.tail.store(tail + 1, std::memory_order_release);
    cm.recv_from[src] = tail + 1;
    return KC_OK;
}

hipStream_t in_stream(Comm &cm, int src)
{
    if (!cm.in_stream[src]) {
        if (hipStreamCreateWithFlags(&cm.in_stream[src], hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&cm.in_ev[src], hipEventDisableTiming) != hipSuccess)
            return nullptr;
    }
    return cm.in_stream[src];
}

// ------------------------------------------------------------------------------------------ one outgoing / incoming image
struct Outgoing {
    kc_image *img = nullptr;          // retained
    std::vector<kc_plane *> planes;   // what travels (borrowed from img, or from `extra`)
    std::vector<kc_plane *> extra;    // dense copies, owned
    std::vector<int> dsts;
    Msg msg{};
    bool handed_over = false;  // references moved into a PendingSend
    ~Outgoing()
    {
        if (handed_over) return;
        image_release(img);
        for (auto *q : extra) plane_release(q);
    }
};

struct Incoming {
    int src = 0;
    uint32_t node = 0, slot = 0;
    uint64_t seq = 0;  // the message's number among those `src` has sent to this rank: what the acknowledgement says
    Msg msg{};
    std::vector<kc_plane *> planes;  // owned until imported
    std::vector<char *> dst;         // where each plane's bytes go (band gather: a row offset inside a full plane)
    ~Incoming()
    {
        for (auto *q : planes) plane_release(q);
    }
};

// Describes `img` (w x rows) for the wire: constants as scalars, aliased planes once, everything else resident with the pool's
// pitch in a block of the pool (caller-owned memory and odd pitches through a dense copy on the compute stream).
int describe(Comm &cm, kc_image *img, Outgoing &o)
{
    Context &c = ctx();
    o.img = img;
    image_retain(img);
    Msg &m = o.msg;
    std::memset(&m, 0, sizeof m);
    m.w = img->w();
    m.h = img->h();
    m.n_planes = (uint32_t)img->n;
    for (int p = 0; p < img->n; ++p) {
        kc_plane *pl = img->planes[p];
        if (pl->kind == kc_plane::CONST) {
            m.kind[p] = 1;
            m.cval[p] = pl->cval;
            continue;
        }
        int alias = -1;
        for (int q = 0; q < p; ++q)
            if (img->planes[q] == pl) alias = q;
        if (alias >= 0) {  // Gray -> Rgba is [p, p, p, ones]: travels once
            m.idx[p] = m.idx[alias];
            continue;
        }
        KC_TRY(plane_materialize(pl));
        const kc_plane *owner = pl;
        while (owner->view_of) owner = owner->view_of;
        kc_plane *snd = pl;
        if (pl->pitch != natural_pitch(pl->w) || (cm.wire == WIRE_IPC && !owner->owned)) {
            kc_plane *dense = nullptr;
            KC_TRY(plane_new_mem(pl->w, pl->h, &dense));
            o.extra.push_back(dense);
            KC_HIP(hipMemcpy2DAsync(dense->dptr, dense->pitch, pl->dptr, pl->pitch, (size_t)pl->w * 4, pl->h, hipMemcpyDeviceToDevice, c.stream));
            snd = dense;
            owner = dense;
        }
        const uint32_t k = (uint32_t)o.planes.size();
        m.idx[p] = k;
        o.planes.push_back(snd);
        WireMem &wm = m.mem[k];
        wm.bytes = (uint64_t)snd->h * snd->pitch;
        wm.local = (uint64_t)(uintptr_t)snd->dptr;
        if (cm.wire == WIRE_IPC) {
            void *base = owner->dptr;
            wm.offset = (uint64_t)((char *)snd->dptr - (char *)base);
            auto it = cm.exported.find(base);
            if (it == cm.exported.end()) {
                hipIpcMemHandle_t h;
                KC_HIP(hipIpcGetMemHandle(&h, base));
                it = cm.exported.emplace(base, h).first;
            }
            wm.handle = it->second;
        }
    }
    m.n_mem = (uint32_t)o.planes.size();
    return KC_OK;
}

// ipc: the `produced` counter reaches a new value behind everything the compute stream holds now.  All writes of the counter
// come from ONE stream, so it never goes backwards whatever stream the caller computes on.
int mark_produced(Comm &cm, Msg &m)
{
    Context &c = ctx();
    KC_HIP(hipEventRecord(cm.compute_ev, c.stream));
    KC_HIP(hipStreamWaitEvent(cm.flag_stream, cm.compute_ev, 0));
    void *d = dev_addr(&cm.shm->cell[cm.rank].produced);
    if (!d) return hip_fail(hipErrorInvalidValue, "device address of the produced counter");
    KC_HIP(hipStreamWriteValue64(cm.flag_stream, d, ++cm.produced_seq, 0));
    m.produced = cm.produced_seq;
    return KC_OK;
}

// Posts `o` to its destinations and keeps it referenced until they have it (ipc); the rccl sends follow in the wire phase.
int post_outgoing(Comm &cm, Outgoing &o)
{
    if (cm.wire == WIRE_IPC && !o.planes.empty()) KC_TRY(mark_produced(cm, o.msg));
    PendingSend ps;
    for (int d : o.dsts) {
        KC_TRY(post(cm, d, o.msg));
        if (cm.wire == WIRE_IPC && !o.planes.empty()) ps.acks.push_back({ d, cm.sent_to[d] });
        cm.planes_sent += o.planes.size();
        for (auto *p : o.planes) cm.bytes_sent += (uint64_t)p->h * p->pitch;
    }
    if (cm.wire == WIRE_IPC) {
        if (!ps.acks.empty()) {
            ps.img = o.img;
            ps.extra = o.extra;
            o.handed_over = true;
            cm.pending.push_back(std::move(ps));
        }
    }
    return KC_OK;
}

bool msg_is_sane(const Msg &m, uint32_t magic)
{
    if (m.magic != magic || (m.n_planes != 1 && m.n_planes != 4) || m.w == 0 || m.h == 0 || m.n_mem > 4) return false;
    for (uint32_t p = 0; p < m.n_planes; ++p)
        if (m.kind[p] > 1 || (m.kind[p] == 0 && m.idx[p] >= m.n_mem)) return false;
    for (uint32_t k = 0; k < m.n_mem; ++k)
        if (m.mem[k].bytes != (uint64_t)m.h * natural_pitch(m.w)) return false;
    return true;
}

// The wire phase of one batch: the planes of everything in `outs` leave, the planes of `ins[i]` arrive at `ins[i]->dst`.
// Both lists are in the order of the transfer list, which is the same on every rank: per pair of ranks the sends and the
// receives therefore match one to one.
int run_wire(Comm &cm, std::vector<std::unique_ptr<Outgoing>> &outs, std::vector<std::unique_ptr<Incoming>> &ins)
{
    Context &c = ctx();
    bool any_in = false, any_out = false;
    for (auto &i : ins) any_in |= !i->dst.empty();
    for (auto &o : outs) any_out |= !o->planes.empty();
    if (cm.wire == WIRE_IPC) {
        if (!any_in) return KC_OK;
        // behind the compute stream: the last readers of the recycled blocks we receive into
        KC_HIP(hipEventRecord(cm.compute_ev, c.stream));
        for (auto &i : ins) {
            if (i->dst.empty()) continue;
            hipStream_t s = in_stream(cm, i->src);
            if (!s) return hip_fail(hipErrorOutOfMemory, "stream for a peer");
            KC_HIP(hipStreamWaitEvent(s, cm.compute_ev, 0));
            void *flag = dev_addr(&cm.shm->cell[i->src].produced);
            void *ack = dev_addr(&cm.shm->cell[cm.rank].copied[i->src]);
            if (!flag || !ack) return hip_fail(hipErrorInvalidValue, "device address of a counter in the shared segment");
            // the sender's host posted the message after it had enqueued the write of this value: the wait cannot outlive it
            KC_HIP(hipStreamWaitValue64(s, flag, i->msg.produced, hipStreamWaitValueGte, ~0ull));
            for (size_t k = 0; k < i->dst.size(); ++k) {
                const char *src_ptr = nullptr;
                KC_TRY(map_peer(cm, i->src, i->msg.mem[k], &src_ptr));
                KC_HIP(hipMemcpyAsync(i->dst[k], src_ptr, i->msg.mem[k].bytes, hipMemcpyDeviceToDevice, s));
                cm.planes_received++;
            }
            KC_HIP(hipStreamWriteValue64(s, ack, i->seq, 0));  // the sender may let go of the planes
            KC_HIP(hipEventRecord(cm.in_ev[i->src], s));
            KC_HIP(hipStreamWaitEvent(c.stream, cm.in_ev[i->src], 0));  // whatever consumes the slot is enqueued after this
        }
        return KC_OK;
    }
    // ---- rccl: one group
    if (!any_in && !any_out) return KC_OK;
    Rccl &r = rccl();
    KC_HIP(hipEventRecord(cm.compute_ev, c.stream));
    KC_HIP(hipStreamWaitEvent(cm.data_stream, cm.compute_ev, 0));
    NcclGroup g;
    KC_TRY(g.start());
    // per pair of ranks, sends and receives in list order: walk both lists by their position in the transfer list
    size_t oi = 0, ii = 0;
    while (oi < outs.size() || ii < ins.size()) {
        const bool take_out = oi < outs.size() && (ii >= ins.size() || outs[oi]->msg.order <= ins[ii]->msg.order);
        if (take_out) {
            Outgoing &o = *outs[oi++];
            for (int d : o.dsts)
                for (auto *p : o.planes) {
                    const int rc = r.send(p->dptr, (size_t)p->h * (p->pitch / 4), NCCL_FLOAT, d, cm.nccl, cm.data_stream);
                    if (rc != 0) return nccl_fail(rc, "ncclSend (plane)");
                }
        } else {
            Incoming &i = *ins[ii++];
            for (size_t k = 0; k < i.dst.size(); ++k) {
                const int rc = r.recv(i.dst[k], i.msg.mem[k].bytes / 4, NCCL_FLOAT, i.src, cm.nccl, cm.data_stream);
                if (rc != 0) return nccl_fail(rc, "ncclRecv (plane)");
                cm.planes_received++;
            }
        }
    }
    KC_TRY(g.end());
    for (auto &o : outs) {
        if (o->planes.empty()) continue;
        PendingSend ps;
        KC_HIP(hipEventCreateWithFlags(&ps.done, hipEventDisableTiming));
        hipError_t e = hipEventRecord(ps.done, cm.data_stream);
        if (e != hipSuccess) {
            (void)hipEventDestroy(ps.done);
            return hip_fail(e, "hipEventRecord (send)");
        }
        ps.img = o->img;
        ps.extra = o->extra;
        o->handed_over = true;
        cm.pending.push_back(std::move(ps));
    }
    if (any_in) {
        KC_HIP(hipEventRecord(cm.recv_ev, cm.data_stream));
        KC_HIP(hipStreamWaitEvent(c.stream, cm.recv_ev, 0));
    }
    return KC_OK;
}

// The received image: fresh planes for what travelled, constants rebuilt, aliases restored.
kc_image *image_from_msg(const Msg &m, const std::vector<kc_plane *> &mem, uint32_t rows)
{
    kc_plane *planes[4] = { nullptr, nullptr, nullptr, nullptr };
    std::vector<kc_plane *> consts;
    for (uint32_t p = 0; p < m.n_planes; ++p) {
        if (m.kind[p] == 1) {
            planes[p] = plane_new_const(m.w, rows, m.cval[p]);
            consts.push_back(planes[p]);
        } else {
            planes[p] = mem[m.idx[p]];
        }
    }
    kc_image *img = image_new((int)m.n_planes, planes);  // retains the planes
    for (auto *q : consts) plane_release(q);
    return img;
}

int exchange_level(Comm &cm, kc_live_graph &lg, const kc_transfer *t, uint32_t n, uint32_t list_pos0)
{
    std::vector<std::unique_ptr<Outgoing>> outs;
    std::vector<std::unique_ptr<Incoming>> ins;
    struct Recv {
        uint32_t node, slot, pos;
        int src;
    };
    std::vector<Recv> recvs;
    // ---- (1) what this rank sends: evaluate (kernels enqueued, nobody waits), describe, post
    for (uint32_t i = 0; i < n;) {
        uint32_t j = i + 1;
        while (j < n && t[j].node_id == t[i].node_id && t[j].slot_id == t[i].slot_id && t[j].src_rank == t[i].src_rank) ++j;
        const int src = t[i].src_rank;
        std::vector<int> dsts;
        bool receiver = false;
        for (uint32_t k = i; k < j; ++k) {
            if (t[k].dst_rank < 0 || t[k].dst_rank >= cm.world || src < 0 || src >= cm.world) {
                set_error("transfer names a rank outside the communicator");
                return KC_ERR_INVALID_ARG;
            }
            dsts.push_back(t[k].dst_rank);
            receiver |= t[k].dst_rank == cm.rank;
        }
        if (receiver) recvs.push_back(Recv{ t[i].node_id, t[i].slot_id, list_pos0 + i, src });
        if (src == cm.rank) {
            KC_TRY(lg.await_clean(t[i].node_id));
            const SlotData *sd = lg.find_slot(t[i].node_id, t[i].slot_id);
            if (!sd) {
                set_error("transfer of a slot the producer does not have");
                return KC_ERR_NO_SLOT_DATA;
            }
            outs.emplace_back(new Outgoing());
            Outgoing &o = *outs.back();
            KC_TRY(describe(cm, sd->image, o));
            o.msg.magic = kMsgSlot;
            o.msg.node = t[i].node_id;
            o.msg.slot = t[i].slot_id;
            o.msg.order = list_pos0 + i;
            o.dsts = dsts;
            KC_TRY(post_outgoing(cm, o));
        }
        i = j;
    }
    // ---- (2) what this rank receives: the descriptions (waits for the producers' hosts only), then planes from the pool
    for (auto &rv : recvs) {
        ins.emplace_back(new Incoming());
        Incoming &in = *ins.back();
        in.src = rv.src;
        in.node = rv.node;
        in.slot = rv.slot;
        KC_TRY(take(cm, rv.src, &in.msg));
        in.seq = cm.recv_from[rv.src];
        if (!msg_is_sane(in.msg, kMsgSlot) || in.msg.node != rv.node || in.msg.slot != rv.slot) {
            set_error("exchange: a peer described another slot than the one this rank expects (do all ranks pass the same transfer list?)");
            return KC_ERR_GENERIC;
        }
        in.msg.order = rv.pos;
        for (uint32_t k = 0; k < in.msg.n_mem; ++k) {
            kc_plane *p = nullptr;
            KC_TRY(plane_new_mem(in.msg.w, in.msg.h, &p));
            in.planes.push_back(p);
            in.dst.push_back((char *)p->dptr);
        }
    }
    // ---- (3) the planes
    KC_TRY(run_wire(cm, outs, ins));
    // ---- (4) hand the slots over
    for (auto &in : ins) {
        kc_image *img = image_from_msg(in->msg, in->planes, in->msg.h);
        const int s = lg.import_slot_data(in->node, in->slot, img);
        image_release(img);
        KC_TRY(s);
    }
    return KC_OK;
}

int exchange_impl(Comm &cm, kc_live_graph &lg, const kc_transfer *t, uint32_t n)
{
    (void)reap(cm, false);
    for (uint32_t i = 0; i < n;) {
        uint32_t j = i + 1;
        while (j < n && t[j].level == t[i].level) ++j;
        KC_TRY(exchange_level(cm, lg, t + i, j - i, i));
        i = j;
    }
    return KC_OK;
}

int gather_impl(Comm &cm, kc_image *band, int32_t y0, uint32_t full_h, int home, kc_image **out)
{
    Context &c = ctx();
    (void)reap(cm, false);
    const uint32_t rows = band->h();
    if (y0 < 0 || (uint64_t)y0 + rows > full_h) {
        set_error("gather: the band does not lie inside the image");
        return KC_ERR_INVALID_ARG;
    }
    std::vector<std::unique_ptr<Outgoing>> outs;
    std::vector<std::unique_ptr<Incoming>> ins;
    outs.emplace_back(new Outgoing());
    Outgoing &mine = *outs.back();
    KC_TRY(describe(cm, band, mine));
    mine.msg.magic = kMsgBand;
    mine.msg.y0 = y0;
    mine.msg.full_h = full_h;
    mine.msg.order = (uint32_t)cm.rank;
    if (cm.rank != home) {
        mine.dsts = { home };
        KC_TRY(post_outgoing(cm, mine));
        KC_TRY(run_wire(cm, outs, ins));
        return KC_OK;
    }
    // ---- the home rank: the full image, its own rows copied in place, everybody else's pulled / received into theirs
    const Msg m0 = mine.msg;
    std::vector<kc_plane *> full;
    struct Drop {
        std::vector<kc_plane *> &v;
        ~Drop()
        {
            for (auto *q : v) plane_release(q);
        }
    } drop{ full };
    for (uint32_t k = 0; k < m0.n_mem; ++k) {
        kc_plane *p = nullptr;
        KC_TRY(plane_new_mem(m0.w, full_h, &p));
        full.push_back(p);
        const kc_plane *src = mine.planes[k];
        KC_HIP(hipMemcpyAsync((char *)p->dptr + (size_t)y0 * p->pitch, src->dptr, (size_t)rows * src->pitch, hipMemcpyDeviceToDevice, c.stream));
    }
    std::vector<std::pair<int64_t, int64_t>> covered{ { y0, (int64_t)y0 + rows } };
    for (int r = 0; r < cm.world; ++r) {
        if (r == home) continue;
        ins.emplace_back(new Incoming());
        Incoming &in = *ins.back();
        in.src = r;
        KC_TRY(take(cm, r, &in.msg));
        in.seq = cm.recv_from[r];
        const Msg &m = in.msg;
        bool same = msg_is_sane(m, kMsgBand) && m.w == m0.w && m.n_planes == m0.n_planes && m.n_mem == m0.n_mem && m.full_h == full_h &&
                    m.y0 >= 0 && (uint64_t)m.y0 + m.h <= full_h;
        for (uint32_t p = 0; same && p < m0.n_planes; ++p)
            same = m.kind[p] == m0.kind[p] && (m.kind[p] == 1 ? std::memcmp(&m.cval[p], &m0.cval[p], 4) == 0 : m.idx[p] == m0.idx[p]);
        if (!same) {
            set_error("gather: rank " + std::to_string(r) + " holds a band of another shape than the home rank's");
            return KC_ERR_GENERIC;
        }
        in.msg.order = (uint32_t)r;
        for (uint32_t k = 0; k < m.n_mem; ++k) in.dst.push_back((char *)full[k]->dptr + (size_t)m.y0 * full[k]->pitch);
        covered.push_back({ m.y0, (int64_t)m.y0 + m.h });
    }
    std::sort(covered.begin(), covered.end());
    int64_t at = 0;
    for (auto &iv : covered) {
        if (iv.first != at) break;
        at = iv.second;
    }
    if (at != (int64_t)full_h) {
        set_error("gather: the bands of the ranks do not tile the image");
        return KC_ERR_GENERIC;
    }
    std::vector<std::unique_ptr<Outgoing>> none;  // nothing of the home rank's goes over the wire
    KC_TRY(run_wire(cm, none, ins));
    *out = image_from_msg(m0, full, full_h);
    return KC_OK;
}

void abort_peers(Comm &cm)
{
    if (cm.shm) cm.shm->abort.store(1, std::memory_order_release);
}

}  // namespace

// ------------------------------------------------------------------------------------------ the interface (kc_runtime.hpp)
int comm_unique_id(void *id, size_t bytes)
{
    if (!id || bytes < 256) {
        set_error("kc_comm_unique_id: the buffer must hold KC_COMM_ID_BYTES bytes");
        return KC_ERR_INVALID_ARG;
    }
    std::memset(id, 0, bytes);
    const char *env = std::getenv("KC_COMM_TRANSPORT");
    const int wire = env && std::strcmp(env, "rccl") == 0 ? WIRE_RCCL : WIRE_IPC;
    if (env && std::strcmp(env, "rccl") != 0 && std::strcmp(env, "ipc") != 0) {
        set_error("KC_COMM_TRANSPORT must be ipc or rccl");
        return KC_ERR_INVALID_ARG;
    }
    if (wire == WIRE_RCCL) {
        Rccl &r = rccl();
        if (!r.ok) {
            set_error("librccl not available");
            return KC_ERR_UNSUPPORTED;
        }
        NcclId a;
        const int rc = r.get_unique_id(&a);
        if (rc != 0) return nccl_fail(rc, "ncclGetUniqueId");
        std::memcpy(id, &a, sizeof a);
    }
    std::random_device rd;
    const uint64_t salt = ((uint64_t)rd() << 32) ^ rd() ^ (uint64_t)now_s();
    std::snprintf((char *)id + kIdNameAt, 60, "/kc_comm_%d_%016llx", (int)getpid(), (unsigned long long)salt);
    ((char *)id)[kIdWireAt] = (char)wire;
    return KC_OK;
}

void comm_sync();

int comm_destroy()
{
    Comm &cm = comm();
    if (!cm.active) return KC_OK;
    comm_sync();
    close_mappings_of(cm, -1);
    if (cm.nccl) (void)rccl().comm_destroy(cm.nccl);
    if (cm.registered) (void)hipHostUnregister(cm.shm->cell);
    if (cm.shm) (void)munmap(cm.shm, sizeof(Shm));
    for (int r = 0; r < kMaxWorld; ++r) {
        if (cm.in_stream[r]) (void)hipStreamDestroy(cm.in_stream[r]);
        if (cm.in_ev[r]) (void)hipEventDestroy(cm.in_ev[r]);
    }
    if (cm.compute_ev) (void)hipEventDestroy(cm.compute_ev);
    if (cm.recv_ev) (void)hipEventDestroy(cm.recv_ev);
    if (cm.flag_stream) (void)hipStreamDestroy(cm.flag_stream);
    if (cm.data_stream) (void)hipStreamDestroy(cm.data_stream);
    cm = Comm{};
    return KC_OK;
}

int comm_init(int rank, int world, const void *id, size_t bytes)
{
    KC_TRY(need_init());
    Comm &cm = comm();
    if (cm.active) {
        set_error("kc_comm_init: a communicator exists already (kc_comm_destroy first)");
        return KC_ERR_INVALID_ARG;
    }
    if (!id || bytes < 256 || world < 1 || world > kMaxWorld || rank < 0 || rank >= world) {
        set_error("kc_comm_init: bad rank / world size (1.." + std::to_string(kMaxWorld) + ") / id");
        return KC_ERR_INVALID_ARG;
    }
    const char *idc = (const char *)id;
    char name[64] = {};
    std::memcpy(name, idc + kIdNameAt, 60);
    const int wire = idc[kIdWireAt];
    if (name[0] != '/' || (wire != WIRE_IPC && wire != WIRE_RCCL)) {
        set_error("kc_comm_init: not an id made by kc_comm_unique_id");
        return KC_ERR_INVALID_ARG;
    }
    if (const char *t = std::getenv("KC_COMM_TIMEOUT_S"))
        if (std::atof(t) > 0) cm.timeout_s = std::atof(t);
    cm.rank = rank;
    cm.world = world;
    cm.wire = wire;
    cm.shm_name = name;
    // a failure returns a status: no retry, nothing re-executed; what has been created is torn down again
    auto fail = [&](int st) {
        if (cm.shm) cm.shm->abort.store(1, std::memory_order_release);
        if (rank == 0) (void)shm_unlink(name);
        cm.active = true;  // so that comm_destroy() walks the members
        const std::string msg = last_error();
        (void)comm_destroy();
        set_error(msg);
        return st;
    };
    // ---- the shared segment: rank 0 creates it, the others wait for it
    int fd = -1;
    if (rank == 0) {
        fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)sizeof(Shm)) != 0) {
            if (fd >= 0) close(fd);
            set_error(std::string("kc_comm_init: cannot create the shared segment ") + name + ": " + std::strerror(errno));
            return fail(KC_ERR_GENERIC);
        }
    } else {
        const int s = wait_for(cm, "rank 0 to create the shared segment", [&] {
            fd = shm_open(name, O_RDWR, 0600);
            if (fd < 0) return false;
            struct stat st;
            if (fstat(fd, &st) == 0 && (size_t)st.st_size >= sizeof(Shm)) return true;
            close(fd);
            fd = -1;
            return false;
        });
        if (s != KC_OK) return fail(s);
    }
    void *mem = mmap(nullptr, sizeof(Shm), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (mem == MAP_FAILED) {
        set_error(std::string("kc_comm_init: mmap of the shared segment: ") + std::strerror(errno));
        return fail(KC_ERR_GENERIC);
    }
    cm.shm = (Shm *)mem;
    if (rank == 0) {  // a fresh segment is all zeros: counters and rings start empty
        cm.shm->world = (uint32_t)world;
        cm.shm->wire = (uint32_t)wire;
        cm.shm->magic.store(kShmMagic, std::memory_order_release);
    } else {
        const int s = wait_for(cm, "rank 0 to initialise the shared segment", [&] { return cm.shm->magic.load(std::memory_order_acquire) == kShmMagic; });
        if (s != KC_OK) return fail(s);
        if (cm.shm->world != (uint32_t)world || cm.shm->wire != (uint32_t)wire) {
            set_error("kc_comm_init: the ranks disagree about the world size");
            return fail(KC_ERR_INVALID_ARG);
        }
    }
    // ---- device side
    hipError_t e = hipHostRegister(cm.shm->cell, sizeof cm.shm->cell, hipHostRegisterMapped);
    cm.registered = e == hipSuccess;
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&cm.flag_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&cm.data_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&cm.compute_ev, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&cm.recv_ev, hipEventDisableTiming);
    if (e != hipSuccess) return fail(hip_fail(e, "kc_comm_init"));
    if (wire == WIRE_IPC) {
        int can = 0;
        (void)hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, ctx().device);
        if (!can) {
            set_error("kc_comm_init: this device cannot wait for a value in a stream (hipDeviceAttributeCanUseStreamWaitValue); use KC_COMM_TRANSPORT=rccl");
            return fail(KC_ERR_UNSUPPORTED);
        }
    } else {
        Rccl &r = rccl();
        if (!r.ok) {
            set_error("librccl not available");
            return fail(KC_ERR_UNSUPPORTED);
        }
        NcclId a;
        std::memcpy(&a, id, sizeof a);
        const int rc = r.comm_init_rank(&cm.nccl, world, a, rank);
        if (rc != 0) return fail(nccl_fail(rc, "ncclCommInitRank"));
    }
    // ---- everybody is here: the name can go (the segment lives as long as somebody maps it)
    cm.shm->attached.fetch_add(1, std::memory_order_acq_rel);
    const int s = wait_for(cm, "all ranks to attach", [&] { return cm.shm->attached.load(std::memory_order_acquire) >= (uint32_t)world; });
    if (rank == 0) (void)shm_unlink(name);
    if (s != KC_OK) return fail(s);
    cm.active = true;
    return KC_OK;
}

void comm_info(int *rank, int *world)
{
    Comm &cm = comm();
    if (rank) *rank = cm.active ? cm.rank : 0;
    if (world) *world = cm.active ? cm.world : 0;
}

const char *comm_wire_name()
{
    Comm &cm = comm();
    return !cm.active ? "" : cm.wire == WIRE_IPC ? "ipc" : "rccl";
}

void comm_stats(uint64_t *planes_sent, uint64_t *planes_received, uint64_t *bytes_sent)
{
    Comm &cm = comm();
    if (planes_sent) *planes_sent = cm.planes_sent;
    if (planes_received) *planes_received = cm.planes_received;
    if (bytes_sent) *bytes_sent = cm.bytes_sent;
}

// kc_pool_trim has given blocks back to the driver: their handles are void, and peers must not keep them mapped
void comm_blocks_freed()
{
    Comm &cm = comm();
    if (!cm.active) return;
    cm.exported.clear();
    cm.shm->cell[cm.rank].epoch.fetch_add(1, std::memory_order_acq_rel);
}

// Works through `t[0 .. n)` level by level (the caller holds the context lock).  Consecutive entries of one slot from one
// rank are one multi-destination send.  An entry whose source and destination are both this rank is legal: the slot is
// replaced by the copy that came back.
int comm_exchange(kc_live_graph &lg, const kc_transfer *t, uint32_t n)
{
    Comm &cm = comm();
    if (!cm.active) {
        set_error("no communicator: kc_comm_init first");
        return KC_ERR_INVALID_ARG;
    }
    const int s = exchange_impl(cm, lg, t, n);
    if (s != KC_OK) abort_peers(cm);
    return s;
}

// Every rank passes its band (rows [y0, y0 + band rows) of an image `full_h` rows high); on `home`, *out (+1 ref) is the image.
int comm_gather_bands(kc_image *band, int32_t y0, uint32_t full_h, int home, kc_image **out)
{
    Comm &cm = comm();
    if (out) *out = nullptr;
    if (!cm.active) {
        set_error("no communicator: kc_comm_init first");
        return KC_ERR_INVALID_ARG;
    }
    if (!band || !out || home < 0 || home >= cm.world) {
        set_error("gather: bad arguments");
        return KC_ERR_INVALID_ARG;
    }
    const int s = gather_impl(cm, band, y0, full_h, home, out);
    if (s != KC_OK) abort_peers(cm);
    return s;
}

int comm_evaluate_partitioned(kc_live_graph &lg, const kc_partition &plan, uint32_t root, kc_image **out)
{
    Comm &cm = comm();
    if (out) *out = nullptr;
    const int world = cm.active ? cm.world : 1, rank = cm.active ? cm.rank : 0;
    if (plan.world != world) {
        set_error("the plan was made for another world size than the communicator's");
        return KC_ERR_INVALID_ARG;
    }
    if (plan.kind == KC_PLAN_BANDS) {
        const Node *rn = lg.g.find(root);
        if (!rn) return KC_ERR_INVALID_NODE_ID;
        uint32_t slot = 0;
        if (rn->type == KC_NODE_GRAPH) {
            const std::vector<uint32_t> outs = rn->graph ? rn->graph->output_ids() : std::vector<uint32_t>{};
            if (outs.empty()) return KC_ERR_NO_SLOT_DATA;
            slot = outs[0];
        } else {
            const std::vector<Slot> slots = node_output_slots(*rn);
            if (!slots.empty()) slot = slots[0].slot_id;
        }
        const kc_band_range b = plan.bands[(size_t)rank];
        kc_image *band = nullptr;
        int s = band_evaluate(lg, root, slot, b.y0, b.y1, &band);
        if (s != KC_OK) {
            if (cm.active) abort_peers(cm);
            return s;
        }
        if (!plan.gather || world == 1) {
            if (out) *out = band;
            else image_release(band);
            return KC_OK;
        }
        kc_image *full = nullptr;
        s = comm_gather_bands(band, b.y0, plan.full_h, plan.home, &full);
        image_release(band);
        if (s != KC_OK) return s;
        if (out) *out = full;
        else if (full) image_release(full);
        return KC_OK;
    }
    if (!plan.xfers.empty()) {
        if (!cm.active) {
            set_error("no communicator: kc_comm_init first");
            return KC_ERR_INVALID_ARG;
        }
        // What this rank can compute before anything arrives, and something that does depend on arrivals will read: enqueue
        // it now, so that it runs while the transfers are in flight (the home rank's own branch of a fan-in).
        std::unordered_map<uint32_t, const kc_placement *> place;
        for (auto &pl : plan.nodes) place[pl.node_id] = &pl;
        std::unordered_map<uint32_t, bool> remote;  // depends on data of another rank
        for (auto &pl : plan.nodes) {               // topological order
            bool rem = pl.rank >= 0 && pl.rank != rank;
            for (uint32_t p : lg.g.get_parents(pl.node_id)) {
                auto it = remote.find(p);
                if (it != remote.end()) rem |= it->second;
            }
            remote[pl.node_id] = rem;
        }
        for (auto &pl : plan.nodes) {
            if (pl.kind != KC_KIND_COMPUTE || pl.rank != rank || remote[pl.node_id]) continue;
            bool feeds_join = false;
            for (uint32_t ch : lg.g.get_children(pl.node_id)) {
                auto it = place.find(ch);
                feeds_join |= it != place.end() && it->second->rank == rank && remote[ch];
            }
            if (feeds_join) {
                const int s = lg.await_clean(pl.node_id);
                if (s != KC_OK) {
                    abort_peers(cm);
                    return s;
                }
            }
        }
        KC_TRY(comm_exchange(lg, plan.xfers.data(), (uint32_t)plan.xfers.size()));
    }
    if (rank != plan.home) return KC_OK;
    const int s = lg.await_clean(root);
    if (s != KC_OK) {
        if (cm.active) abort_peers(cm);
        return s;
    }
    if (out) {
        // the root's first output slot
        const SlotData *sd = nullptr;
        for (uint32_t k = 0; k < 4 && !sd; ++k) sd = lg.find_slot(root, k);
        if (!sd) return KC_ERR_NO_SLOT_DATA;
        image_retain(sd->image);
        *out = sd->image;
    }
    return KC_OK;
}

// kc_sync / kc_shutdown: transfers in flight hold references -- let them go
void comm_sync()
{
    Comm &cm = comm();
    if (!cm.active) return;
    if (cm.flag_stream) (void)hipStreamSynchronize(cm.flag_stream);
    if (cm.data_stream) (void)hipStreamSynchronize(cm.data_stream);
    for (int r = 0; r < kMaxWorld; ++r)
        if (cm.in_stream[r]) (void)hipStreamSynchronize(cm.in_stream[r]);
    (void)reap(cm, true);
}

}  // namespace kc
