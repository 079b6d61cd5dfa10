// The exchange step of a multi-GPU evaluation, behind the C ABI: slots that cross a rank boundary of a partition plan
// (partition.cpp) move between processes with RCCL send / recv over xGMI.  One process per GPU; every process holds the same
// graph and works through the same list of transfers in the same order.
//
// The reference has no distributed layer; what it has is the readiness rule that makes one possible -- a node needs nothing
// but its parents' slot data (src/engine.rs:213-275) -- and kc_live_graph_import_slot_data is the receiving end of that rule.
//
// librccl is bound at first use (dlopen), like hiprtc in specialize.cpp: a process that never calls kc_comm_* needs no RCCL.
// Two communicators over the same ranks: a slot's DESCRIPTION (size, which planes are constants or aliases: 64 bytes) travels
// on one with its own stream, its planes on the other, so that a receiver blocking on the next description (it needs the size
// before it can allocate and post the receive) never waits for plane data still in flight -- on the home rank of a fan-in the
// inbound transfers of all branches then run at the same time, each over its own xGMI link.
// Streams: planes are sent behind an event of the compute stream (the kernels that produce them have only been enqueued);
// received planes come from the stream-ordered pool, so the receive waits for an event of the compute stream too (the block may
// still be read by queued kernels) and the compute stream waits for the receive before anything consumes the slot.  A sent
// image stays referenced until the event behind its send has fired.  No host thread waits for plane data.
#include <dlfcn.h>

#include <cstring>

#include "kc_runtime.hpp"

namespace kc {
namespace {

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
#define KC_NCCL(call, what)                      \
    do {                                         \
        const int rc_ = (call);                  \
        if (rc_ != 0) return nccl_fail(rc_, what); \
    } while (0)

// What a receiver has to know about a slot before it can post the receives of its planes.
struct SlotHeader {
    uint32_t magic, w, h, n_planes;  // n_planes: 1 = Gray, 4 = Rgba
    uint32_t kind[4];                // 0 = the idx-th plane sent, 1 = broadcast constant cval
    uint32_t idx[4];
    float cval[4];
};
static_assert(sizeof(SlotHeader) == 64, "one 64-byte message");
constexpr uint32_t kHeaderMagic = 0x4b43534cu;  // "KCSL"
constexpr int kHeaderRing = 64;

struct PendingSend {
    hipEvent_t done;
    kc_image *img;  // retained until `done` has fired
    std::vector<kc_plane *> extra;  // dense copies made for the send
};

struct Comm {
    bool active = false;
    int rank = 0, world = 1;
    NcclComm hdr = nullptr, data = nullptr;
    hipStream_t hdr_stream = nullptr, data_stream = nullptr;
    hipEvent_t compute_ev = nullptr, recv_ev = nullptr;
    SlotHeader *dev_ring = nullptr;   // kHeaderRing outgoing + kHeaderRing incoming descriptions in HBM
    SlotHeader *host_ring = nullptr;  // pinned mirror
    hipEvent_t ring_ev[kHeaderRing] = {};
    uint32_t ring_next = 0;
    std::deque<PendingSend> pending;
    uint64_t planes_sent = 0, planes_received = 0, bytes_sent = 0;
};

Comm &comm()
{
    static Comm c;
    return c;
}

// Drops the references of sends whose event has fired (all of them when `wait`).
void reap(Comm &cm, bool wait)
{
    while (!cm.pending.empty()) {
        PendingSend &p = cm.pending.front();
        if (wait) (void)hipEventSynchronize(p.done);
        else if (hipEventQuery(p.done) != hipSuccess) {
            (void)hipGetLastError();
            break;
        }
        (void)hipEventDestroy(p.done);
        image_release(p.img);
        for (auto *q : p.extra) plane_release(q);
        cm.pending.pop_front();
    }
}

size_t natural_pitch(uint32_t w) { return ((size_t)w * 4 + 255) / 256 * 256; }  // what plane_new_mem gives a plane of this width

}  // namespace

int comm_unique_id(void *id, size_t bytes)
{
    Rccl &r = rccl();
    if (!r.ok) {
        set_error("librccl not available");
        return KC_ERR_UNSUPPORTED;
    }
    if (!id || bytes < 2 * sizeof(NcclId)) {
        set_error("kc_comm_unique_id: the buffer must hold KC_COMM_ID_BYTES bytes");
        return KC_ERR_INVALID_ARG;
    }
    NcclId a, b;
    KC_NCCL(r.get_unique_id(&a), "ncclGetUniqueId");
    KC_NCCL(r.get_unique_id(&b), "ncclGetUniqueId");
    std::memcpy(id, &a, sizeof a);
    std::memcpy((char *)id + sizeof a, &b, sizeof b);
    return KC_OK;
}

int comm_destroy()
{
    Comm &cm = comm();
    if (!cm.active) return KC_OK;
    reap(cm, true);
    if (cm.hdr_stream) (void)hipStreamSynchronize(cm.hdr_stream);
    if (cm.data_stream) (void)hipStreamSynchronize(cm.data_stream);
    Rccl &r = rccl();
    if (cm.hdr) (void)r.comm_destroy(cm.hdr);
    if (cm.data) (void)r.comm_destroy(cm.data);
    for (auto &e : cm.ring_ev)
        if (e) (void)hipEventDestroy(e);
    if (cm.compute_ev) (void)hipEventDestroy(cm.compute_ev);
    if (cm.recv_ev) (void)hipEventDestroy(cm.recv_ev);
    if (cm.dev_ring) (void)hipFree(cm.dev_ring);
    if (cm.host_ring) (void)hipHostFree(cm.host_ring);
    if (cm.hdr_stream) (void)hipStreamDestroy(cm.hdr_stream);
    if (cm.data_stream) (void)hipStreamDestroy(cm.data_stream);
    cm = Comm{};
    return KC_OK;
}

int comm_init(int rank, int world, const void *id, size_t bytes)
{
    KC_TRY(need_init());
    Rccl &r = rccl();
    if (!r.ok) {
        set_error("librccl not available");
        return KC_ERR_UNSUPPORTED;
    }
    Comm &cm = comm();
    if (cm.active) {
        set_error("kc_comm_init: a communicator exists already (kc_comm_destroy first)");
        return KC_ERR_INVALID_ARG;
    }
    if (!id || bytes < 2 * sizeof(NcclId) || world < 1 || rank < 0 || rank >= world) {
        set_error("kc_comm_init: bad rank / world size / id");
        return KC_ERR_INVALID_ARG;
    }
    NcclId a, b;
    std::memcpy(&a, id, sizeof a);
    std::memcpy(&b, (const char *)id + sizeof a, sizeof b);
    cm.rank = rank;
    cm.world = world;
    // a failure returns a status: no retry, nothing re-executed; what has been created is torn down again
    int s = KC_OK;
    auto fail = [&](int st) {
        cm.active = true;  // so that comm_destroy() walks the members
        (void)comm_destroy();
        return st;
    };
    int rc = r.comm_init_rank(&cm.hdr, world, a, rank);
    if (rc != 0) return fail(nccl_fail(rc, "ncclCommInitRank (descriptions)"));
    rc = r.comm_init_rank(&cm.data, world, b, rank);
    if (rc != 0) return fail(nccl_fail(rc, "ncclCommInitRank (planes)"));
    hipError_t e = hipStreamCreateWithFlags(&cm.hdr_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&cm.data_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&cm.compute_ev, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&cm.recv_ev, hipEventDisableTiming);
    if (e == hipSuccess) e = hipMalloc((void **)&cm.dev_ring, 2 * kHeaderRing * sizeof(SlotHeader));
    if (e == hipSuccess) e = hipHostMalloc((void **)&cm.host_ring, 2 * kHeaderRing * sizeof(SlotHeader), hipHostMallocDefault);
    for (int i = 0; i < kHeaderRing && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&cm.ring_ev[i], hipEventDisableTiming);
    if (e != hipSuccess) {
        s = hip_fail(e, "kc_comm_init");
        return fail(s);
    }
    cm.active = true;
    return KC_OK;
}

void comm_info(int *rank, int *world)
{
    Comm &cm = comm();
    if (rank) *rank = cm.active ? cm.rank : 0;
    if (world) *world = cm.active ? cm.world : 0;
}

void comm_stats(uint64_t *planes_sent, uint64_t *planes_received, uint64_t *bytes_sent)
{
    Comm &cm = comm();
    if (planes_sent) *planes_sent = cm.planes_sent;
    if (planes_received) *planes_received = cm.planes_received;
    if (bytes_sent) *bytes_sent = cm.bytes_sent;
}

// Works through `t[0 .. n)` in order (the caller holds the context lock).  Consecutive entries of one slot from one rank
// are one multi-destination send.  An entry whose source and destination are both this rank sends to itself (RCCL allows
// that inside a group): the slot is replaced by the copy that came back.
int comm_exchange(kc_live_graph &lg, const kc_transfer *t, uint32_t n)
{
    Comm &cm = comm();
    if (!cm.active) {
        set_error("no communicator: kc_comm_init first");
        return KC_ERR_INVALID_ARG;
    }
    Rccl &r = rccl();
    Context &c = ctx();
    reap(cm, false);
    for (uint32_t i = 0; i < n;) {
        uint32_t j = i + 1;
        while (j < n && t[j].node_id == t[i].node_id && t[j].slot_id == t[i].slot_id && t[j].src_rank == t[i].src_rank) ++j;
        const uint32_t node = t[i].node_id, slot = t[i].slot_id;
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
        i = j;
        const bool sender = src == cm.rank;
        if (!sender && !receiver) continue;

        // ---- the sender's side of the description ----
        const uint32_t ring = cm.ring_next++ % kHeaderRing;
        KC_HIP(hipEventSynchronize(cm.ring_ev[ring]));  // the slot's previous use (fires at once until the ring has wrapped)
        SlotHeader *h_out = &cm.host_ring[ring], *h_in = &cm.host_ring[kHeaderRing + ring];
        SlotHeader *d_out = &cm.dev_ring[ring], *d_in = &cm.dev_ring[kHeaderRing + ring];
        kc_image *img = nullptr;
        std::vector<kc_plane *> send_planes, extra;
        if (sender) {
            KC_TRY(lg.await_clean(node));  // enqueues the branch's kernels; nobody waits for them
            const SlotData *sd = lg.find_slot(node, slot);
            if (!sd) {
                set_error("transfer of a slot the producer does not have");
                return KC_ERR_NO_SLOT_DATA;
            }
            img = sd->image;
            image_retain(img);
            std::memset(h_out, 0, sizeof *h_out);
            h_out->magic = kHeaderMagic;
            h_out->w = img->w();
            h_out->h = img->h();
            h_out->n_planes = (uint32_t)img->n;
            int s = KC_OK;
            for (int p = 0; p < img->n && s == KC_OK; ++p) {
                kc_plane *pl = img->planes[p];
                if (pl->kind == kc_plane::CONST) {  // Mix's alpha = 1, a broadcast Value: a scalar, not 64 MiB of ones
                    h_out->kind[p] = 1;
                    h_out->cval[p] = pl->cval;
                    continue;
                }
                int alias = -1;
                for (int q = 0; q < p; ++q)
                    if (img->planes[q] == pl) alias = q;
                if (alias >= 0) {  // aliased planes (Gray -> Rgba is [p, p, p, ones]) travel once
                    h_out->idx[p] = h_out->idx[alias];
                    continue;
                }
                s = plane_materialize(pl);
                if (s != KC_OK) break;
                kc_plane *snd = pl;
                if (pl->pitch != natural_pitch(pl->w)) {  // caller-owned memory / a row view: the receiver's plane has the pool's pitch
                    kc_plane *dense = nullptr;
                    s = plane_new_mem(pl->w, pl->h, &dense);
                    if (s != KC_OK) break;
                    hipError_t e = hipMemcpy2DAsync(dense->dptr, dense->pitch, pl->dptr, pl->pitch, (size_t)pl->w * 4, pl->h,
                                                    hipMemcpyDeviceToDevice, c.stream);
                    if (e != hipSuccess) {
                        plane_release(dense);
                        s = hip_fail(e, "dense copy for a transfer");
                        break;
                    }
                    extra.push_back(dense);
                    snd = dense;
                }
                h_out->idx[p] = (uint32_t)send_planes.size();
                send_planes.push_back(snd);
            }
            if (s != KC_OK) {
                image_release(img);
                for (auto *q : extra) plane_release(q);
                return s;
            }
            KC_HIP(hipMemcpyAsync(d_out, h_out, sizeof *h_out, hipMemcpyHostToDevice, cm.hdr_stream));
        }
        KC_NCCL(r.group_start(), "ncclGroupStart");
        if (sender)
            for (int d : dsts) KC_NCCL(r.send(d_out, sizeof(SlotHeader), NCCL_UINT8, d, cm.hdr, cm.hdr_stream), "ncclSend (description)");
        if (receiver) KC_NCCL(r.recv(d_in, sizeof(SlotHeader), NCCL_UINT8, src, cm.hdr, cm.hdr_stream), "ncclRecv (description)");
        KC_NCCL(r.group_end(), "ncclGroupEnd");
        SlotHeader hdr{};
        std::vector<kc_plane *> recv_planes;
        if (receiver) {
            KC_HIP(hipMemcpyAsync(h_in, d_in, sizeof *h_in, hipMemcpyDeviceToHost, cm.hdr_stream));
            KC_HIP(hipEventRecord(cm.ring_ev[ring], cm.hdr_stream));
            KC_HIP(hipStreamSynchronize(cm.hdr_stream));  // 64 bytes on their own stream: waits for the producer's host, not for plane data
            hdr = *h_in;
            uint32_t n_mem = 0;
            bool good = hdr.magic == kHeaderMagic && (hdr.n_planes == 1 || hdr.n_planes == 4) && hdr.w > 0 && hdr.h > 0;
            for (uint32_t p = 0; good && p < hdr.n_planes; ++p) {
                if (hdr.kind[p] > 1 || hdr.idx[p] >= 4) good = false;
                else if (hdr.kind[p] == 0) n_mem = std::max(n_mem, hdr.idx[p] + 1);
            }
            if (!good) {
                if (img) image_release(img);
                for (auto *q : extra) plane_release(q);
                set_error("malformed slot description received");
                return KC_ERR_GENERIC;
            }
            for (uint32_t k = 0; k < n_mem; ++k) {
                kc_plane *p = nullptr;
                int s = plane_new_mem(hdr.w, hdr.h, &p);
                if (s != KC_OK) {
                    for (auto *q : recv_planes) plane_release(q);
                    if (img) image_release(img);
                    for (auto *q : extra) plane_release(q);
                    return s;
                }
                recv_planes.push_back(p);
            }
        } else {
            KC_HIP(hipEventRecord(cm.ring_ev[ring], cm.hdr_stream));
        }

        // ---- the planes ----
        // behind the compute stream: the producing kernels (sender) / the last readers of the recycled blocks (receiver)
        KC_HIP(hipEventRecord(cm.compute_ev, c.stream));
        KC_HIP(hipStreamWaitEvent(cm.data_stream, cm.compute_ev, 0));
        if (!send_planes.empty() || !recv_planes.empty()) {
            KC_NCCL(r.group_start(), "ncclGroupStart");
            if (sender)
                for (int d : dsts)
                    for (auto *p : send_planes) {
                        const size_t count = (size_t)p->h * (p->pitch / 4);  // the whole pitched buffer
                        KC_NCCL(r.send(p->dptr, count, NCCL_FLOAT, d, cm.data, cm.data_stream), "ncclSend (plane)");
                        cm.planes_sent++;
                        cm.bytes_sent += count * 4;
                    }
            if (receiver)
                for (auto *p : recv_planes) {
                    KC_NCCL(r.recv(p->dptr, (size_t)p->h * (p->pitch / 4), NCCL_FLOAT, src, cm.data, cm.data_stream), "ncclRecv (plane)");
                    cm.planes_received++;
                }
            KC_NCCL(r.group_end(), "ncclGroupEnd");
        }
        if (sender) {
            PendingSend ps;
            KC_HIP(hipEventCreateWithFlags(&ps.done, hipEventDisableTiming));
            KC_HIP(hipEventRecord(ps.done, cm.data_stream));
            ps.img = img;
            ps.extra = extra;
            cm.pending.push_back(std::move(ps));
        }
        if (receiver) {
            // whatever consumes the slot is enqueued on the compute stream after this wait
            KC_HIP(hipEventRecord(cm.recv_ev, cm.data_stream));
            KC_HIP(hipStreamWaitEvent(c.stream, cm.recv_ev, 0));
            kc_plane *planes[4] = { nullptr, nullptr, nullptr, nullptr };
            std::vector<kc_plane *> consts;
            for (uint32_t p = 0; p < hdr.n_planes; ++p) {
                if (hdr.kind[p] == 1) {
                    planes[p] = plane_new_const(hdr.w, hdr.h, hdr.cval[p]);
                    consts.push_back(planes[p]);
                } else {
                    planes[p] = recv_planes[hdr.idx[p]];
                }
            }
            kc_image *in = image_new((int)hdr.n_planes, planes);  // retains the planes
            for (auto *q : consts) plane_release(q);
            for (auto *q : recv_planes) plane_release(q);
            const int s = lg.import_slot_data(node, slot, in);
            image_release(in);
            KC_TRY(s);
        }
    }
    return KC_OK;
}

int comm_evaluate_partitioned(kc_live_graph &lg, const kc_partition &plan, uint32_t root, kc_image **out)
{
    Comm &cm = comm();
    if (out) *out = nullptr;
    if (!cm.active || plan.world != cm.world) {
        set_error("the plan was made for another world size than the communicator's");
        return KC_ERR_INVALID_ARG;
    }
    KC_TRY(comm_exchange(lg, plan.xfers.data(), (uint32_t)plan.xfers.size()));
    if (cm.rank != plan.home) return KC_OK;
    KC_TRY(lg.await_clean(root));
    if (out) {
        // the root's first output slot
        const SlotData *sd = nullptr;
        for (uint32_t s = 0; s < 4 && !sd; ++s) sd = lg.find_slot(root, s);
        if (!sd) return KC_ERR_NO_SLOT_DATA;
        image_retain(sd->image);
        *out = sd->image;
    }
    return KC_OK;
}

// kc_sync / kc_shutdown
void comm_sync()
{
    Comm &cm = comm();
    if (!cm.active) return;
    (void)hipStreamSynchronize(cm.data_stream);
    reap(cm, true);
}

}  // namespace kc
