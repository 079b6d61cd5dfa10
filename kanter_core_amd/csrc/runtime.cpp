// Context, HBM plane pool, planes, images and the lazy pointwise-chain machinery.
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>

#include <algorithm>

#ifdef KC_HOST_SAMPLE
#include <dlfcn.h>
#include <new>
#include <signal.h>
#include <sys/time.h>
#include <ucontext.h>
#endif

#include "kc_runtime.hpp"

namespace kc {

static const int KC_RETRY_CHAIN = -1000;  // internal: chain_launch made operands resident, rebuild

static thread_local std::string g_last_error;

Context &ctx()
{
    static Context c;
    return c;
}

void set_error(const std::string &msg) { g_last_error = msg; }
const std::string &last_error() { return g_last_error; }

#ifdef KC_HOST_PROFILE
static std::map<std::string, std::pair<unsigned long long, unsigned long long>> &prof_table()
{
    static std::map<std::string, std::pair<unsigned long long, unsigned long long>> t;
    return t;
}
ProfScope::~ProfScope()
{
    auto &e = prof_table()[name];
    e.first += __builtin_ia32_rdtsc() - t0;
    e.second++;
}
void prof_report()
{
    for (auto &kv : prof_table())
        std::fprintf(stderr, "KC_PROF %-28s calls %9llu  cycles %14llu  per call %9.0f\n", kv.first.c_str(), kv.second.second,
                     kv.second.first, (double)kv.second.first / (double)kv.second.second);
}
#endif

#ifdef KC_HOST_SAMPLE
// A sampling profile of the host side for tuning builds (KC_HOST_SAMPLE=1 python -m kanter_core_amd.build --force):
// SIGALRM every 100 us (wall clock: the CPU-time timers tick at 10 ms) records the interrupted PC; kc_shutdown writes "module offset" lines
// to $KC_SAMPLE_OUT, which profiles/host_samples.py turns into a per-function table with this build's symbols.
static constexpr size_t kMaxSamples = 1u << 20;
static void *g_samples[kMaxSamples];
static std::atomic<size_t> g_n_samples{ 0 };
static void on_sigprof(int, siginfo_t *, void *uc)
{
    const size_t i = g_n_samples.fetch_add(1, std::memory_order_relaxed);
    if (i < kMaxSamples) g_samples[i] = (void *)((ucontext_t *)uc)->uc_mcontext.gregs[REG_RIP];
}
// This library's own heap traffic by request size (the definitions below replace operator new / delete for
// this library only: hidden visibility binds them locally).
static std::atomic<unsigned long long> g_new_calls[12];
void sampler_count_new(size_t bytes)
{
    int b = 0;
    while ((16u << b) < bytes && b < 11) ++b;
    g_new_calls[b].fetch_add(1, std::memory_order_relaxed);
}
void sampler_start()
{
    if (!std::getenv("KC_SAMPLE_OUT")) return;
    struct sigaction sa;
    std::memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = on_sigprof;
    sa.sa_flags = SA_SIGINFO | SA_RESTART;
    sigaction(SIGALRM, &sa, nullptr);
    struct itimerval it = { { 0, 100 }, { 0, 100 } };
    setitimer(ITIMER_REAL, &it, nullptr);
}
void sampler_report()
{
    const char *path = std::getenv("KC_SAMPLE_OUT");
    if (!path) return;
    struct itimerval off = { { 0, 0 }, { 0, 0 } };
    setitimer(ITIMER_REAL, &off, nullptr);
    for (int b = 0; b < 12; ++b)
        if (g_new_calls[b].load()) std::fprintf(stderr, "KC_SAMPLE operator new, <= %5u bytes: %llu calls\n", 16u << b, g_new_calls[b].load());
    FILE *f = std::fopen(path, "w");
    if (!f) return;
    const size_t n = std::min(g_n_samples.load(), kMaxSamples);
    for (size_t i = 0; i < n; ++i) {
        Dl_info di;
        if (dladdr(g_samples[i], &di) && di.dli_fname)
            std::fprintf(f, "%s %lx %s\n", di.dli_fname, (unsigned long)((char *)g_samples[i] - (char *)di.dli_fbase),
                         di.dli_sname ? di.dli_sname : "?");
        else
            std::fprintf(f, "? %lx ?\n", (unsigned long)g_samples[i]);
    }
    std::fclose(f);
}
#endif

}  // namespace kc
#ifdef KC_HOST_SAMPLE
void *operator new(size_t bytes)
{
    kc::sampler_count_new(bytes);
    void *p = std::malloc(bytes ? bytes : 1);
    if (!p) throw std::bad_alloc();
    return p;
}
void operator delete(void *p) noexcept { std::free(p); }
void operator delete(void *p, size_t) noexcept { std::free(p); }
#endif
namespace kc {

int hip_fail(hipError_t e, const char *what)
{
    set_error(std::string(what) + ": " + hipGetErrorString(e));
    return KC_ERR_HIP;
}

int need_init()
{
    Context &c = ctx();
    if (!c.inited) {
        set_error("kc_init() has not succeeded: no gfx950 device bound (there is no CPU fallback)");
        return KC_ERR_NO_DEVICE;
    }
    // HIP's current device is per host thread and 0 for a new one: a call from another thread than kc_init's (or after the caller
    // switched devices) must not allocate, launch or load code objects on a different GPU than the library's stream belongs to
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != c.device) {
        hipError_t e = hipSetDevice(c.device);
        if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
    }
    return KC_OK;
}

// ------------------------------------------------------------------------------------------
// Pool: exact-size free lists.  Everything runs on one stream, so a block freed by the host can
// be handed to the next kernel immediately (stream order makes the reuse safe).  Replaces the
// reference's RAM/disk tiering (src/transient_buffer.rs:249-434): 288 GB of HBM hold the planes.
// ------------------------------------------------------------------------------------------
int pool_alloc(size_t bytes, void **out)
{
    Context &c = ctx();
    std::lock_guard<std::recursive_mutex> lk(c.mu);
    auto it = c.free_blocks.find(bytes);
    if (it != c.free_blocks.end()) {
        *out = it->second;
        c.free_blocks.erase(it);
        c.bytes_cached -= bytes;
        c.bytes_in_use += bytes;
        return KC_OK;
    }
    hipError_t e = hipMalloc(out, bytes);
    if (e == hipErrorOutOfMemory) {
        (void)hipGetLastError();
        pool_trim();
        e = hipMalloc(out, bytes);
    }
    if (e == hipErrorOutOfMemory) {
        (void)hipGetLastError();
        set_error("out of HBM");
        return KC_ERR_OUT_OF_MEMORY;
    }
    if (e != hipSuccess) return hip_fail(e, "hipMalloc");
    c.bytes_in_use += bytes;
    return KC_OK;
}

void pool_free(void *p, size_t bytes)
{
    if (!p) return;
    Context &c = ctx();
    std::lock_guard<std::recursive_mutex> lk(c.mu);
    if (!c.inited) return;  // device already torn down
    c.free_blocks.emplace(bytes, p);
    c.bytes_in_use -= bytes;
    c.bytes_cached += bytes;
}

int pool_trim()
{
    Context &c = ctx();
    std::lock_guard<std::recursive_mutex> lk(c.mu);
    if (c.stream) (void)hipStreamSynchronize(c.stream);
    for (auto &kv : c.free_blocks) (void)hipFree(kv.second);
    if (!c.free_blocks.empty()) comm_blocks_freed();  // peers of a communicator must not keep freed blocks mapped
    c.free_blocks.clear();
    c.bytes_cached = 0;
    return KC_OK;
}

// ------------------------------------------------------------------------------------------
// Planes
// ------------------------------------------------------------------------------------------
static size_t pitch_for(uint32_t w)
{
    size_t row = (size_t)w * sizeof(float);
    return (row + 255) / 256 * 256;
}

int plane_new_mem(uint32_t w, uint32_t h, kc_plane **out)
{
    KC_TRY(need_init());
    if (w == 0 || h == 0) {
        set_error("plane with zero extent");
        return KC_ERR_INVALID_ARG;
    }
    kc_plane *p = new kc_plane();
    p->w = w;
    p->h = h;
    p->kind = kc_plane::MEM;
    p->pitch = pitch_for(w);
    p->bytes = p->pitch * h;
    p->owned = true;
    void *d = nullptr;
    int s = pool_alloc(p->bytes, &d);
    if (s != KC_OK) {
        delete p;
        return s;
    }
    p->dptr = (float *)d;
    *out = p;
    return KC_OK;
}

kc_plane *plane_new_const(uint32_t w, uint32_t h, float v)
{
    kc_plane *p = new kc_plane();
    p->w = w;
    p->h = h;
    p->kind = kc_plane::CONST;
    p->cval = v;
    return p;
}

void plane_release(kc_plane *p)
{
    if (!p) return;
    if (--p->refs == 0) {
        if (p->chain) delete p->chain;
        if (p->link) delete p->link;
        if (p->rz_src) plane_release(p->rz_src);
        if (p->view_of) plane_release(p->view_of);
        if (p->owned && p->dptr) pool_free(p->dptr, p->bytes);
        delete p;
    }
}

Chain::~Chain()
{
    plane_release(start);
    for (auto &s : steps) plane_release(s.operand);
    for (auto *j : joined) plane_release(j);
}

// Operand of the step that combines a joined chain's value with the value CH_SAVE_LOAD put aside.  Never freed: its count
// starts far above anything retain / release can take away.
kc_plane *saved_value_marker()
{
    static kc_plane *m = [] {
        kc_plane *p = new kc_plane();
        p->kind = kc_plane::CONST;
        p->refs = 1 << 30;
        return p;
    }();
    return m;
}

ChainLink::~ChainLink()
{
    // Release the prefix iteratively: a long run of otherwise unreferenced lazy planes would recurse
    // once per plane through plane_release -> ~ChainLink.
    plane_release(start);
    plane_release(step.operand);
    kc_plane *q = prev;
    prev = nullptr;
    while (q) {
        if (--q->refs != 0) break;  // still referenced elsewhere
        kc_plane *next = nullptr;
        if (q->link) {
            next = q->link->prev;
            q->link->prev = nullptr;
        }
        q->refs = 1;
        plane_release(q);  // frees q (its link no longer has a prev)
        q = next;
    }
}

// Input planes a chain may gather while it is being built: KC_CHAIN_MAX_IN when programs get kernels of their own (more than
// KC_CHAIN_INTERP_IN run on nothing else; chain_launch splits the chain while such a kernel is not there), else what the
// interpreter handles.
static int chain_in_limit()
{
    const Context &c = ctx();
    return (c.wide && c.fusion && !c.plain_chains && specialize_get_mode() != 0) ? KC_CHAIN_MAX_IN : KC_CHAIN_INTERP_IN;
}

// Identity of a chain input, as input_index() sees it, as a value (see ChainLink::InKey).
static void link_add_input(ChainLink &L, const kc_plane *q)
{
    ChainLink::InKey k;
    if (q->kind == kc_plane::MEM)
        k = { q->dptr, q->pitch };
    else if (q->kind == kc_plane::RESIZE)
        k = { q, ~(size_t)0 };
    else
        return;
    for (int i = 0; i < L.n_in && i <= KC_CHAIN_MAX_IN; ++i)
        if (L.ins[i] == k) return;
    if (L.n_in <= KC_CHAIN_MAX_IN) L.ins[L.n_in] = k;
    L.n_in++;  // may exceed the array: only the count matters beyond KC_CHAIN_MAX_IN
}

static int input_index(std::vector<const kc_plane *> &ins, const kc_plane *p);

// Distinct planes a flat chain reads, as chain_fill will see them NOW.
static int chain_distinct_inputs(const Chain &ch)
{
    std::vector<const kc_plane *> ins;
    auto counts = [](const kc_plane *q) { return q->kind == kc_plane::MEM || q->kind == kc_plane::RESIZE; };
    if (counts(ch.start)) input_index(ins, ch.start);
    for (auto &s : ch.steps)
        if (counts(s.operand)) input_index(ins, s.operand);
    return (int)ins.size();
}

// Records chain_fill will write for a flat chain: a {+, -, *} step on a plane (or a saved value) followed by "constant - acc"
// is one record (CH_*_INV).
static size_t chain_record_count(const Chain &ch)
{
    size_t r = 0;
    for (size_t i = 0; i < ch.steps.size(); ++i, ++r) {
        const ChainStep &st = ch.steps[i];
        uint8_t code = st.code == CH_ADD_R ? (uint8_t)CH_ADD : st.code == CH_MUL_R ? (uint8_t)CH_MUL : st.code;
        const bool plane = st.operand == saved_value_marker() || st.operand->kind != kc_plane::CONST;
        if (plane && code <= CH_MUL && i + 1 < ch.steps.size() && ch.steps[i + 1].code == CH_SUB_R &&
            ch.steps[i + 1].operand->kind == kc_plane::CONST && ch.steps[i + 1].operand != saved_value_marker())
            ++i;
    }
    return r;
}

// Builds p->chain (start + steps in order) from the links; planes forced meanwhile end the walk.  A step whose operand is a
// chain that has not run (a join, plane_mix) becomes: CH_SAVE_LOAD(that chain's start), that chain's steps, the step itself
// with the saved value as its operand and the running value on the other side.
static void chain_flatten(kc_plane *p)
{
    KC_PROF("chain_flatten");
    if (p->chain) return;
    Chain *c = new Chain();
    c->steps.reserve(p->link->length);
    kc_plane *q = p;
    for (;;) {
        ChainLink *L = q->link;
        kc_plane *opnd = L->step.operand;
        if (opnd->kind == kc_plane::LAZY) {
            chain_flatten(opnd);  // (may hold joins of its own)
            const Chain &sub = *opnd->chain;
            // collected back to front: the combining step, the joined chain's steps, its start
            uint8_t code = L->step.code;  // written for "acc op operand"; the operand's value is now the running one
            switch (code) {
            case CH_SUB_L: code = CH_SUB_R; break;
            case CH_SUB_R: code = CH_SUB_L; break;
            case CH_DIV_L: code = CH_DIV_R; break;
            case CH_DIV_R: code = CH_DIV_L; break;
            case CH_POW_L: code = CH_POW_R; break;
            case CH_POW_R: code = CH_POW_L; break;
            default: break;  // + and * commute (ADD_R / MUL_R are canonicalised by chain_fill)
            }
            c->steps.push_back({ code, saved_value_marker(), 0 });
            for (size_t i = sub.steps.size(); i-- > 0;) {
                ChainStep st = sub.steps[i];
                // the joined chain's own joins happen while this one's saved value is aside: one level up
                if (st.code == CH_SAVE_LOAD || st.operand == saved_value_marker()) st.level++;
                c->steps.push_back(st);
            }
            c->steps.push_back({ (uint8_t)CH_SAVE_LOAD, sub.start, 0 });
            c->joined.push_back(opnd);
            plane_retain(opnd);
        } else {
            c->steps.push_back(L->step);
        }
        if (L->prev && L->prev->kind == kc_plane::LAZY) {
            q = L->prev;
            continue;
        }
        c->start = L->prev ? L->prev : L->start;
        break;
    }
    std::reverse(c->steps.begin(), c->steps.end());
    plane_retain(c->start);
    for (auto &s : c->steps) plane_retain(s.operand);
    p->chain = c;
}

Operand plane_operand(const kc_plane *p)
{
    Operand o;
    if (p->kind == kc_plane::CONST) {
        o.ptr = nullptr;
        o.pitch = 0;
        o.c = p->cval;
    } else {
        o.ptr = p->dptr;
        o.pitch = (uint32_t)(p->pitch / sizeof(float));
        o.c = 0.0f;
    }
    return o;
}

// ---- building the kernel program for one lazy plane -------------------------------------
struct BuiltChain {
    ChainProgram prog;
    uint32_t in_refs[KC_CHAIN_MAX_IN] = {};  // reference counts of channel 0's resident inputs when the program was built
    int mode = 0;  // 0 = {+,-,*}, 1 = + divide, 2 = + pow
    kc_plane *sampled[KC_CHAIN_MAX_BATCH] = { nullptr, nullptr, nullptr, nullptr };  // RESIZE operand per channel
    bool joins = false;  // the program holds CH_SAVE_LOAD: only its own compiled kernel can run it
};

static int input_index(std::vector<const kc_plane *> &ins, const kc_plane *p)
{
    for (size_t i = 0; i < ins.size(); ++i)
        if (ins[i] == p || (p->kind == kc_plane::MEM && ins[i]->kind == kc_plane::MEM && ins[i]->dptr == p->dptr &&
                            ins[i]->pitch == p->pitch))
            return (int)i;
    ins.push_back(p);
    return (int)ins.size() - 1;
}

// Fills entry `b` of the program from plane `p`'s chain.  When b > 0 the codes / operand
// pattern must equal entry 0's (returns false otherwise).  Resident planes take input slots
// 0 .. KM-1 in order of appearance; a RESIZE operand (at most one distinct per chain, see
// chain_prepare) takes slot KM and is recorded in bc.sampled[b].
static bool chain_fill(BuiltChain &bc, int b, const kc_plane *p)
{
    KC_PROF("chain_fill");
    const Chain &ch = *p->chain;
    ChainProgram &P = bc.prog;
    std::vector<const kc_plane *> ins;
    kc_plane *samp = nullptr;
    // pass 1: slots of the resident planes
    if (ch.start->kind == kc_plane::MEM) input_index(ins, ch.start);
    for (auto &st : ch.steps)
        if (st.operand->kind == kc_plane::MEM) input_index(ins, st.operand);
    auto slot = [&](kc_plane *q, float *c) -> int {
        if (q == saved_value_marker()) return KC_CHAIN_SRC_SAVED - 1;
        if (q->kind == kc_plane::MEM) return input_index(ins, q);
        if (q->kind == kc_plane::RESIZE) {
            samp = q;
            return -2;
        }
        *c = q->cval;
        return -1;
    };
    const int km = (int)ins.size();
    if (km > KC_CHAIN_MAX_IN) return false;  // planes_force splits such chains before they get here
    float c0 = 0.0f;
    int start_src = slot(ch.start, &c0);
    if (start_src == -2) start_src = km;
    if (b == 0) {
        std::memset(&P, 0, sizeof(P));
        P.start_src = start_src;
    } else if (P.start_src != start_src) {
        return false;
    }
    P.start_c[b] = c0;
    size_t r = 0;  // records written: a {+, -, *} step on a plane followed by "constant - acc" is one (CH_*_INV)
    for (size_t i = 0; i < ch.steps.size(); ++i, ++r) {
        const ChainStep &st = ch.steps[i];
        float c = 0.0f;
        int src = slot(st.operand, &c);
        if (src == -2) src = km;
        uint8_t code = st.code;
        if (code == CH_ADD_R) code = CH_ADD;  // commutative: the same IEEE result
        if (code == CH_MUL_R) code = CH_MUL;
        if (code == CH_SAVE_LOAD) bc.joins = true;
        if (src >= 0 && code <= CH_MUL && i + 1 < ch.steps.size() && ch.steps[i + 1].code == CH_SUB_R &&
            ch.steps[i + 1].operand->kind == kc_plane::CONST && ch.steps[i + 1].operand != saved_value_marker()) {
            static const uint8_t inv[4] = { CH_ADD_INV, CH_SUBL_INV, CH_SUBR_INV, CH_MUL_INV };
            code = inv[code];
            c = ch.steps[i + 1].operand->cval;
            ++i;
        }
        uint32_t word = chain_op_word(code, src);
        if (st.code == CH_SAVE_LOAD || st.operand == saved_value_marker()) word |= (uint32_t)st.level << 16;
        auto rec = [&](int ch_) -> ChainStepRec & { return (r & 1) ? P.step[ch_][r / 2].b : P.step[ch_][r / 2].a; };
        if (b == 0) {
            rec(0).word = word;
            if (code == CH_DIV_L || code == CH_DIV_R) bc.mode = bc.mode < 1 ? 1 : bc.mode;
            if (code == CH_POW_L || code == CH_POW_R) bc.mode = 2;
        } else if (r >= P.n_ops || rec(0).word != word) {
            return false;
        }
        rec(b).word = word;
        rec(b).c = c;
    }
    if (b == 0)
        P.n_ops = (uint32_t)r;
    else if (P.n_ops != r)
        return false;
    const uint32_t n_in = (uint32_t)km + (samp ? 1u : 0u);
    if (b == 0)
        P.n_in = n_in;
    else if (P.n_in != n_in || (samp != nullptr) != (bc.sampled[0] != nullptr))
        return false;
    if (samp && b > 0) {
        const kc_plane *s0 = bc.sampled[0];
        if (samp->rz_filter != s0->rz_filter || samp->rz_src->w != s0->rz_src->w || samp->rz_src->h != s0->rz_src->h)
            return false;
    }
    bc.sampled[b] = samp;
    for (int k = 0; k < km; ++k) {
        P.in[b][k] = ins[k]->dptr;
        P.in_pitch[b][k] = (uint32_t)(ins[k]->pitch / 16);
        if (b == 0) bc.in_refs[k] = (uint32_t)ins[k]->refs;
    }
    if (samp) {
        P.samp_src[b] = samp->rz_src->dptr;
        P.samp_pitch[b] = (uint32_t)(samp->rz_src->pitch / 4);
    }
    return true;
}

// A chain can carry at most ONE distinct resampled operand into the fused resize+chain kernel;
// any further RESIZE operands are run through the plain resize kernel first.
static int chain_prepare(kc_plane *p)
{
    Chain &ch = *p->chain;
    kc_plane *first = nullptr;
    auto visit = [&](kc_plane *q) -> int {
        if (q->kind != kc_plane::RESIZE) return KC_OK;
        if (!first) {
            first = q;
            return KC_OK;
        }
        return q == first ? KC_OK : resize_force(q);
    };
    KC_TRY(visit(ch.start));
    for (auto &st : ch.steps) KC_TRY(visit(st.operand));
    return KC_OK;
}

// MI355X has a 256 MB memory-side Infinity Cache in front of HBM.  A launch that streams more than that through it
// (the BASELINE graphs: 604 MB per evaluation at 4096^2) evicts everything before it is used again, and pays for the
// allocations: the same 1-in-1-out stream runs at 5.9 TB/s with plain accesses and at 6.4-7.1 TB/s when the streams that
// cannot stay are marked nontemporal (profiles/tilecopy.hip, r03_tilecopy3.txt).  Policy: a launch whose streams all fit
// is left alone; otherwise its full-size inputs are read nontemporal, and its results are stored normally while they fit
// by themselves -- the consumer of a result (the next node, an export, the next evaluation writing the same pool block)
// then finds it on chip.
// The same for a chain launch, input by input.  A plane that nobody but the chain holds (refs == 1: an intermediate that
// is freed as soon as the launch is enqueued) will never be read again; a plane somebody else holds as well -- a source
// embedded in the graph, cached slot data, a caller's handle -- is read again by the next node or the next evaluation.  As
// many of those as fit stay cacheable (most-referenced first), everything else is streamed; the result stays cacheable
// only if it fits beside them.  Measured on config #1's traffic (profiles/tilecopy.hip add2, r03_tilecopy4.txt): plain
// 102.9 us, everything streamed 92.6 us, one input kept 84.2 us -- and keeping the RESULT instead is 88.7 us when it goes
// to the same block every time but 98.9 us when a graph still holds its previous result, which is the usual case.
uint32_t chain_cache_policy(const uint32_t *refs, uint32_t n, uint64_t stream_bytes, uint64_t out_bytes)
{
    if (!ctx().cache_policy) return 0;
    static const long forced = std::getenv("KC_NT_FORCE") ? std::strtol(std::getenv("KC_NT_FORCE"), nullptr, 0) : -1;  // tuning: this mask for every launch
    uint32_t all = 0;
    for (uint32_t k = 0; k < n; ++k) all |= KC_CHAIN_NT_BIT(k);
    if (forced >= 0) return (uint32_t)forced & (all | 0x100u);
    const uint64_t budget = (uint64_t)ctx().cache_budget_mb << 20;
    if (n * stream_bytes + out_bytes <= budget) return 0;
    uint32_t mask = all;
    uint64_t used = 0;
    for (;;) {
        int best = -1;
        for (uint32_t k = 0; k < n; ++k)
            if ((mask & KC_CHAIN_NT_BIT(k)) && refs[k] >= 2 && (best < 0 || refs[k] > refs[best])) best = (int)k;
        if (best < 0 || used + stream_bytes > budget) break;
        used += stream_bytes;
        mask &= ~KC_CHAIN_NT_BIT(best);
    }
    if (used + out_bytes > budget) mask |= 0x100u;
    return mask;
}

uint32_t cache_policy_mask(uint64_t in_bytes, uint64_t out_bytes, uint32_t n_resident)
{
    if (!ctx().cache_policy) return 0;
    static const long forced = std::getenv("KC_NT_FORCE") ? std::strtol(std::getenv("KC_NT_FORCE"), nullptr, 0) : -1;  // tuning: this mask for every launch
    if (forced >= 0) return (uint32_t)forced & (((1u << n_resident) - 1u) | 0x100u);
    const uint64_t budget = (uint64_t)ctx().cache_budget_mb << 20;
    if (in_bytes + out_bytes <= budget) return 0;
    uint32_t mask = (1u << n_resident) - 1u;
    if (out_bytes > budget) mask |= 0x100u;
    return mask;
}

// Launches a (non-resampling) chain program whose inputs and outputs are set: the ahead-of-time kernel of a one-step
// program, else the kernel compiled for the program at run time if there is one, else the interpreter.  Shared by
// chain_launch and by the replay of a recorded evaluation (replay.cpp).
hipError_t chain_dispatch(ChainProgram &P, int batch, int mode, uint32_t w, uint32_t h, size_t out_pitch_bytes)
{
    Context &c = ctx();
    const uint32_t row_units = (w + 3) / 4;
    bool launched = false;
    // Rows can be flattened into one run when every plane is dense (pitch == 16 * row_units).
    bool dense = (size_t)row_units * 16 == out_pitch_bytes;
    for (int b = 0; b < batch && dense; ++b)
        for (uint32_t k = 0; k < P.n_in; ++k)
            if (P.in_pitch[b][k] != row_units) dense = false;
    if (dense) {
        P.rows = 1;
        P.row_units = row_units * h;
    } else {
        P.rows = h;
        P.row_units = row_units;
    }
    if (!launched && P.n_ops == 1 && c.chain1) {
        // a single Mix step: its ahead-of-time straight-line kernel (chain1.hip)
        Chain1Args a{};
        const uint32_t w0 = P.step[0][0].a.word, from = w0 >> 8;
        for (int b = 0; b < batch; ++b) {
            if (P.start_src >= 0) {
                a.start[b] = P.in[b][P.start_src];
                a.start_pitch[b] = P.in_pitch[b][P.start_src];
            }
            a.start_c[b] = P.start_c[b];
            if (from) {
                a.operand[b] = P.in[b][from - 1];
                a.operand_pitch[b] = P.in_pitch[b][from - 1];
            }
            a.operand_c[b] = a.c[b] = P.step[b][0].a.c;
            a.out[b] = P.out[b];
            a.out_pitch[b] = P.out_pitch[b];
        }
        a.rows = P.rows;
        a.row_units = P.row_units;
        const unsigned nt = (P.start_src >= 0 && (P.nt_mask >> P.start_src & 1u) ? 1u : 0u) | (from && (P.nt_mask >> (from - 1) & 1u) ? 2u : 0u) |
                            (P.nt_mask & 0x100u ? 4u : 0u);
        hipError_t e = launch_chain1(a, batch, (int)(w0 & 0xffu), nt, c.stream);
        if (e != hipSuccess) return e;
        c.counters["chain1_launches"]++;
        launched = true;
    }
    if (!launched) {
        // a program-specialised straight-line kernel if one has been compiled (specialize.cpp) ...
        hipError_t e = launch_chain_specialized(P, batch, c.stream, &launched);
        // ... otherwise the interpreter -- which does not know the codes of a program that joins two chains (chain_launch
        // never sends it one; a replay of such a launch after kc_set_specialize(0) can)
        if (e == hipSuccess && !launched) {
            if (P.n_in > (uint32_t)KC_CHAIN_INTERP_IN) return hipErrorNotReady;  // as below: compiled kernels only
            for (uint32_t i = 0; i < P.n_ops; ++i)
                if ((((i & 1u) ? P.step[0][i / 2].b.word : P.step[0][i / 2].a.word) & 0xffu) == CH_SAVE_LOAD) return hipErrorNotReady;
            e = launch_chain(P, batch, mode, c.max_blocks, c.chain_unroll, c.stream);
        }
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

static int chain_flatten_to_fit(kc_plane *p, int limit);
static_assert(sizeof(ChainProgram) <= 4096, "a launch passes at most 4 KB of kernel arguments");

static int chain_launch(BuiltChain &bc, kc_plane *const *planes, int batch)
{
    KC_PROF("chain_launch");
    Context &c = ctx();
    ChainProgram &P = bc.prog;
    const kc_plane *p0 = planes[0];
    // Outputs are fresh pool planes, all with the pitch of their width.
    std::vector<kc_plane *> outs(batch, nullptr);
    for (int b = 0; b < batch; ++b) {
        int s = plane_new_mem(p0->w, p0->h, &outs[b]);
        if (s != KC_OK) {
            for (auto *o : outs) plane_release(o);
            return s;
        }
        P.out[b] = outs[b]->dptr;
        P.out_pitch[b] = (uint32_t)(outs[b]->pitch / 16);
    }
    {
        const uint64_t px4 = 4ull * p0->w * p0->h;
        const uint32_t resident = P.n_in - (bc.sampled[0] ? 1u : 0u);
        P.nt_mask = chain_cache_policy(bc.in_refs, resident, px4 * batch, px4 * batch);
    }
    bool launched = false;
    if (bc.sampled[0]) {
        int s = chain_resize_launch(P, batch, bc.mode, bc.sampled, &launched);
        if (s != KC_OK || !launched) {
            for (auto *o : outs) plane_release(o);
            if (s != KC_OK) return s;
            // not eligible: run the resamples on their own, the caller rebuilds the program
            KC_TRY(resize_force_many(bc.sampled, batch));
            return KC_RETRY_CHAIN;
        }
    }
    if (!launched) {
        hipError_t e = chain_dispatch(P, batch, bc.mode, p0->w, p0->h, outs[0]->pitch);
        if (e == hipErrorNotReady && (bc.joins || P.n_in > (uint32_t)KC_CHAIN_INTERP_IN)) {
            // The program joins chains or reads more planes than the interpreter handles, and its own kernel has not been
            // compiled (yet): the joined chains run on their own and the chain is cut to the interpreter's input count, as
            // both were before such programs existed; the caller rebuilds the program around the results.
            for (auto *o : outs) plane_release(o);
            std::vector<kc_plane *> subs;  // channel by channel for each joined chain: planes_force batches neighbours
            for (size_t j = 0; j < planes[0]->chain->joined.size(); ++j)
                for (int b = 0; b < batch; ++b)
                    if (j < planes[b]->chain->joined.size()) subs.push_back(planes[b]->chain->joined[j]);
            if (!subs.empty()) KC_TRY(planes_force(subs.data(), (int)subs.size()));
            for (int b = 0; b < batch; ++b) {
                if (planes[b]->kind != kc_plane::LAZY) continue;  // ran as one of the joined chains' inputs (planes_force skips it)
                delete planes[b]->chain;
                planes[b]->chain = nullptr;
                KC_TRY(chain_flatten_to_fit(planes[b], KC_CHAIN_INTERP_IN));
            }
            c.counters["join_fallbacks"]++;
            if (c.capture) c.capture->ok = false;  // not what this evaluation will do once the kernel is there: do not record it
            return KC_RETRY_CHAIN;
        }
        if (e != hipSuccess) {
            for (auto *o : outs) plane_release(o);
            return hip_fail(e, "launch_chain");
        }
        if (bc.joins) c.counters["join_launches"]++;
        if (P.n_in > (uint32_t)KC_CHAIN_INTERP_IN) c.counters["wide_launches"]++;
    }
    c.launches++;
    {
        // algorithmic bytes of this launch: every resident input plane once, the result once, per channel; a
        // resampled operand costs its (small) source instead of a full-size plane
        uint64_t px = (uint64_t)p0->w * p0->h, bytes = 0;
        for (int b = 0; b < batch; ++b) {
            const uint32_t resident = P.n_in - (bc.sampled[b] ? 1u : 0u);
            bytes += 4 * px * (resident + 1);
            if (bc.sampled[b]) bytes += 4 * (uint64_t)bc.sampled[b]->rz_src->w * bc.sampled[b]->rz_src->h;
        }
        c.alg_bytes += bytes;
    }
    if (c.capture) {  // an evaluation is being recorded for replay (replay.cpp)
        ReplayCapture &cap = *c.capture;
        if (bc.sampled[0] || cap.launches.size() >= 32) cap.ok = false;  // plain chain launches are what a recording may hold
        else {
            cap.launches.emplace_back();
            ReplayLaunch &L = cap.launches.back();
            L.prog = P;
            L.batch = batch;
            L.mode = bc.mode;
            std::memcpy(L.in_refs, bc.in_refs, sizeof L.in_refs);
            L.w = p0->w;
            L.h = p0->h;
            for (int b = 0; b < batch; ++b) L.planes[b] = planes[b];
        }
    }
    for (int b = 0; b < batch; ++b) {
        kc_plane *p = planes[b];
        Chain *old = p->chain;
        ChainLink *old_link = p->link;
        p->chain = nullptr;
        p->link = nullptr;
        p->kind = kc_plane::MEM;
        p->dptr = outs[b]->dptr;
        p->pitch = outs[b]->pitch;
        p->bytes = outs[b]->bytes;
        p->owned = true;
        outs[b]->owned = false;  // ownership of the block moved into p
        outs[b]->dptr = nullptr;
        plane_release(outs[b]);
        delete old;  // drops the operand references (after the launch is enqueued)
        delete old_link;
    }
    return KC_OK;
}

// Flattens p's chain.  The input count was bounded when the chain was built, but a constant operand can have been
// materialised since (kc_plane_materialize, a resize of it ...) and now occupies an input slot, and joined chains that had to
// run on their own (chain_launch) each became one.  Then the prefix is run on its own and the chain restarts from its result.
static int chain_flatten_to_fit(kc_plane *p, int limit)
{
    chain_flatten(p);
    // (ChainLink::length counts a fusable pair as one record; a constant that has been materialised since no longer fuses)
    while (chain_distinct_inputs(*p->chain) > limit || chain_record_count(*p->chain) > (size_t)KC_CHAIN_MAX_OPS) {
        if (!p->chain->joined.empty()) {
            // Joined chains bring their own inputs along.  The bound plane_mix checked (join_ok) held for the chains as they
            // were then; a prefix that has been run since counts as ONE new input beside them.  Run the joined chains: each
            // becomes one input.
            std::vector<kc_plane *> subs(p->chain->joined);
            for (auto *q : subs) plane_retain(q);
            const int fs = planes_force(subs.data(), (int)subs.size());
            for (auto *q : subs) plane_release(q);
            KC_TRY(fs);
            if (p->kind != kc_plane::LAZY) return KC_OK;  // (ran as an input of one of them)
            delete p->chain;
            p->chain = nullptr;
            chain_flatten(p);
            continue;
        }
        kc_plane *prev = p->link->prev;
        if (!prev || prev->kind != kc_plane::LAZY) {
            set_error("chain with more inputs than a program holds cannot be split");
            return KC_ERR_UNSUPPORTED;
        }
        // The longest prefix that fits -- not simply the chain minus its last step, which would run every further step as a
        // launch of its own once the first cut has been made.  Walk the links forward from the chain's start, counting input
        // planes as chain_fill will see them.
        {
            std::vector<kc_plane *> path;  // the lazy planes from the first link to p's predecessor
            for (kc_plane *q = prev; q && q->kind == kc_plane::LAZY; q = q->link->prev) path.push_back(q);
            std::reverse(path.begin(), path.end());
            std::vector<ChainLink::InKey> keys;
            auto add = [&](const kc_plane *q) {
                ChainLink::InKey k;
                if (q->kind == kc_plane::MEM) k = { q->dptr, q->pitch };
                else if (q->kind == kc_plane::RESIZE) k = { q, ~(size_t)0 };
                else return;
                if (std::find(keys.begin(), keys.end(), k) == keys.end()) keys.push_back(k);
            };
            const ChainLink &first = *path[0]->link;
            add(first.prev ? first.prev : first.start);
            kc_plane *cut = nullptr;
            size_t steps = 0;
            for (kc_plane *q : path) {
                const kc_plane *o = q->link->step.operand;
                if (o->kind == kc_plane::LAZY) {  // a joined chain brings its inputs along
                    for (int i = 0; i < o->link->n_in && i <= KC_CHAIN_MAX_IN; ++i)
                        if (std::find(keys.begin(), keys.end(), o->link->ins[i]) == keys.end()) keys.push_back(o->link->ins[i]);
                    steps += o->link->length + 2u;
                    if (o->link->n_in > KC_CHAIN_MAX_IN) break;
                } else {
                    add(o);
                    ++steps;
                }
                if ((int)keys.size() > limit || steps > (size_t)KC_CHAIN_MAX_OPS) break;
                cut = q;
            }
            if (cut) prev = cut;
        }
        KC_TRY(plane_force(prev));
        delete p->chain;
        p->chain = nullptr;
        chain_flatten(p);
    }
    return KC_OK;
}

int planes_force(kc_plane *const *planes, int n)
{
    KC_PROF("planes_force");
    std::lock_guard<std::recursive_mutex> lk(ctx().mu);
    std::vector<kc_plane *> todo;
    KC_TRY(resize_force_many(planes, n));
    for (int i = 0; i < n; ++i) {
        kc_plane *p = planes[i];
        if (!p) continue;
        if (p->kind != kc_plane::LAZY) continue;
        bool dup = false;
        for (auto *q : todo) dup |= (q == p);
        if (!dup) todo.push_back(p);
    }
    if (todo.empty()) return KC_OK;  // constants / resident planes: nothing to launch
    KC_TRY(need_init());
    for (auto *p : todo) {
        // forcing a prefix or an operand below can run another plane of this very list (an image may hold a chain and
        // its own prefix): it is resident then and has nothing left to flatten
        if (p->kind != kc_plane::LAZY) continue;
        // (a resampled operand goes through the fused resize + chain kernels: the interpreter's input count)
        KC_TRY(chain_flatten_to_fit(p, chain_in_limit()));
        if (p->kind != kc_plane::LAZY) continue;
        {
            bool resampled = p->chain->start->kind == kc_plane::RESIZE;
            for (auto &st : p->chain->steps) resampled |= st.operand->kind == kc_plane::RESIZE;
            if (resampled && chain_distinct_inputs(*p->chain) > KC_CHAIN_INTERP_IN) {
                delete p->chain;
                p->chain = nullptr;
                KC_TRY(chain_flatten_to_fit(p, KC_CHAIN_INTERP_IN));
                if (p->kind != kc_plane::LAZY) continue;
            }
        }
        KC_TRY(chain_prepare(p));
    }
    todo.erase(std::remove_if(todo.begin(), todo.end(), [](kc_plane *q) { return q->kind != kc_plane::LAZY; }), todo.end());
    size_t i = 0;
    while (i < todo.size()) {
        // (a launch further up may have had to run this plane first: a joined chain's fallback, a split prefix)
        if (todo[i]->kind != kc_plane::LAZY) {
            ++i;
            continue;
        }
        if (!todo[i]->chain) {
            KC_TRY(chain_flatten_to_fit(todo[i], KC_CHAIN_INTERP_IN));
            if (todo[i]->kind != kc_plane::LAZY) continue;
        }
        BuiltChain bc;
        kc_plane *group[KC_CHAIN_MAX_BATCH];
        int batch = 0;
        if (!chain_fill(bc, 0, todo[i])) {
            set_error("internal: chain does not fit one program");
            return KC_ERR_UNSUPPORTED;
        }
        group[batch++] = todo[i];
        size_t j = i + 1;
        while (j < todo.size() && batch < KC_CHAIN_MAX_BATCH && todo[j]->kind == kc_plane::LAZY && todo[j]->chain &&
               todo[j]->w == todo[i]->w && todo[j]->h == todo[i]->h && chain_fill(bc, batch, todo[j])) {
            group[batch++] = todo[j];
            ++j;
        }
        int s = chain_launch(bc, group, batch);
        if (s == KC_RETRY_CHAIN) continue;  // resampled operands are resident now: rebuild this group
        KC_TRY(s);
        i = j;
    }
    return KC_OK;
}

int plane_force(kc_plane *p) { return planes_force(&p, 1); }

int plane_materialize(kc_plane *p)
{
    KC_TRY(need_init());
    std::lock_guard<std::recursive_mutex> lk(ctx().mu);
    if (p->kind == kc_plane::LAZY || p->kind == kc_plane::RESIZE) return plane_force(p);
    if (p->kind == kc_plane::CONST) {
        kc_plane *m = nullptr;
        KC_TRY(plane_new_mem(p->w, p->h, &m));
        hipError_t e = launch_fill(m->dptr, (uint32_t)(m->pitch / 4), p->w, p->h, p->cval, ctx().stream);
        if (e != hipSuccess) {
            plane_release(m);
            return hip_fail(e, "launch_fill");
        }
        ctx().launches++;
        ctx().alg_bytes += (uint64_t)p->w * p->h * 4;
        p->kind = kc_plane::MEM;
        p->dptr = m->dptr;
        p->pitch = m->pitch;
        p->bytes = m->bytes;
        p->owned = true;
        m->owned = false;
        m->dptr = nullptr;
        plane_release(m);
    }
    return KC_OK;
}

// ---- l op r ------------------------------------------------------------------------------
static float fold_const(int mix, float l, float r)
{
    switch (mix) {
    case KC_MIX_ADD: return l + r;
    case KC_MIX_SUBTRACT: return l - r;
    case KC_MIX_MULTIPLY: return l * r;
    default: return l / r;  // KC_MIX_DIVIDE (Pow is never folded on the host, see plane_mix)
    }
}

static uint8_t code_for(int mix, bool acc_is_left)
{
    switch (mix) {
    case KC_MIX_ADD: return acc_is_left ? CH_ADD : CH_ADD_R;
    case KC_MIX_SUBTRACT: return acc_is_left ? CH_SUB_L : CH_SUB_R;
    case KC_MIX_MULTIPLY: return acc_is_left ? CH_MUL : CH_MUL_R;
    case KC_MIX_DIVIDE: return acc_is_left ? CH_DIV_L : CH_DIV_R;
    default: return acc_is_left ? CH_POW_L : CH_POW_R;
    }
}

// Would continuing `acc` with a step on `opnd` exceed one program (steps or distinct input planes)?
static bool chain_full_with(const kc_plane *acc, const kc_plane *opnd)
{
    if (acc->link->length >= (uint32_t)KC_CHAIN_MAX_OPS) return true;
    int limit = chain_in_limit();
    if (opnd->kind == kc_plane::RESIZE) limit = KC_CHAIN_INTERP_IN;  // the fused resize + chain kernels
    for (int i = 0; i < acc->link->n_in && i <= KC_CHAIN_MAX_IN; ++i)
        if (acc->link->ins[i].q == ~(size_t)0) limit = KC_CHAIN_INTERP_IN;
    if (acc->link->n_in < limit) return false;  // one more operand adds at most one input
    ChainLink probe;
    probe.n_in = acc->link->n_in;
    for (int i = 0; i < probe.n_in && i <= KC_CHAIN_MAX_IN; ++i) probe.ins[i] = acc->link->ins[i];
    link_add_input(probe, opnd);
    return probe.n_in > limit;
}

// Which of two lazy operands plane_mix runs first: the shorter chain (the right one on a tie).
static kc_plane *lazy_pair_victim(kc_plane *l, kc_plane *r)
{
    return (l != r && l->link->length >= r->link->length) ? r : l;
}

// Both inputs of a Mix are chains that have not run.  One of them used to be run on the spot -- a plane written and read
// again, a launch -- because a program has one running value.  With one more register it need not be: the longer chain goes on
// (`acc`), the shorter one is kept as the step's operand and evaluated INSIDE the same program (CH_SAVE_LOAD, chain_flatten).
// BASELINE config #4's add tree over eight branches: 9 launches and 33 plane passes per channel become 5 and 25.
// Conditions: no resampled operand on either side, the saved values needed at once fit (KC_CHAIN_MAX_SAVED: the shorter
// chain may hold joins of its own), the two together fit one program, and
// the longer one has room for the shorter one's RESULT as an input, which is what it becomes whenever the program's own kernel
// is not available (chain_launch).
static bool join_ok(const kc_plane *acc, const kc_plane *sub)
{
    Context &c = ctx();
    if (!c.join || !c.fusion || c.plain_chains || acc == sub || specialize_get_mode() == 0) return false;
    const ChainLink &A = *acc->link, &B = *sub->link;
    const int limit = chain_in_limit();
    if (std::max<int>(A.saved, B.saved + 1) > KC_CHAIN_MAX_SAVED || A.n_in > limit - 1 || B.n_in > limit) return false;
    if (A.length + B.length + 2u > (uint32_t)KC_CHAIN_MAX_OPS) return false;
    ChainLink probe;
    probe.n_in = A.n_in;
    for (int i = 0; i < A.n_in; ++i) {
        if (A.ins[i].q == ~(size_t)0) return false;  // a resampled operand: the fused resize + chain kernels keep their programs plain
        probe.ins[i] = A.ins[i];
    }
    for (int i = 0; i < B.n_in; ++i) {
        if (B.ins[i].q == ~(size_t)0) return false;
        bool seen = false;
        for (int k = 0; k < probe.n_in && k <= KC_CHAIN_MAX_IN; ++k) seen |= probe.ins[k] == B.ins[i];
        if (!seen) {
            if (probe.n_in >= limit) return false;
            probe.ins[probe.n_in++] = B.ins[i];
        }
    }
    return true;
}

// The planes plane_mix(l[c], r[c]) would have to run before it can extend a chain, forced TOGETHER: the R, G
// and B chains of an RGBA operand share one program and leave as one launch (blockIdx.y) instead of three.
int planes_mix_prepare(kc_plane *const *ls, kc_plane *const *rs, int n)
{
    std::lock_guard<std::recursive_mutex> lk(ctx().mu);
    std::vector<kc_plane *> todo;
    for (int c = 0; c < n; ++c)
        if (ls[c]->kind == kc_plane::LAZY && rs[c]->kind == kc_plane::LAZY) {
            kc_plane *sub = lazy_pair_victim(ls[c], rs[c]);
            if (!join_ok(sub == ls[c] ? rs[c] : ls[c], sub)) todo.push_back(sub);
        }
    if (!todo.empty()) KC_TRY(planes_force(todo.data(), (int)todo.size()));
    todo.clear();
    for (int c = 0; c < n; ++c) {
        if (ls[c]->kind == kc_plane::LAZY && rs[c]->kind == kc_plane::LAZY) continue;  // a join (join_ok has checked the room)
        kc_plane *acc = ls[c]->kind == kc_plane::LAZY ? ls[c] : rs[c]->kind == kc_plane::LAZY ? rs[c] : nullptr;
        if (acc && chain_full_with(acc, acc == ls[c] ? rs[c] : ls[c])) todo.push_back(acc);
    }
    if (!todo.empty()) KC_TRY(planes_force(todo.data(), (int)todo.size()));
    return KC_OK;
}

int plane_mix(int mix, kc_plane *l, kc_plane *r, kc_plane **out)
{
    KC_PROF("plane_mix");
    if (mix < KC_MIX_ADD || mix > KC_MIX_POW) {
        set_error("invalid MixType");
        return KC_ERR_INVALID_ARG;
    }
    if (l->w != r->w || l->h != r->h) {
        set_error("Mix operands differ in size (resize_buffers must run first)");
        return KC_ERR_INVALID_ARG;
    }
    std::lock_guard<std::recursive_mutex> lk(ctx().mu);
    // Constants fold on the host -- except Pow: powf is evaluated by the device's own routine (pow_positive,
    // <= 1 ulp from libm), and a folded constant must be the value a pixel of the same operands would get,
    // so constant ^ constant stays a (zero-input) chain and is computed by the kernel.
    if (l->kind == kc_plane::CONST && r->kind == kc_plane::CONST && mix != KC_MIX_POW) {
        *out = plane_new_const(l->w, l->h, fold_const(mix, l->cval, r->cval));
        return KC_OK;
    }
    KC_TRY(need_init());
    // Pick the running value: a lazy operand is continued, the other side becomes the step operand.
    kc_plane *acc = nullptr, *opnd = nullptr;
    bool acc_is_left = true;
    bool join = false;
    if (l->kind == kc_plane::LAZY && r->kind == kc_plane::LAZY) {
        kc_plane *sub = lazy_pair_victim(l, r);  // keep the longer chain lazy
        if (join_ok(sub == l ? r : l, sub)) join = true;
        else KC_TRY(plane_force(sub));
    }
    if (join) {
        kc_plane *sub = lazy_pair_victim(l, r);
        acc = sub == l ? r : l;
        opnd = sub;
        acc_is_left = acc == l;
    } else if (l->kind == kc_plane::LAZY) {
        acc = l;
        opnd = r;
        acc_is_left = true;
    } else if (r->kind == kc_plane::LAZY) {
        acc = r;
        opnd = l;
        acc_is_left = false;
    }
    // A chain that is full (steps or distinct inputs) is run first and restarted from its result.
    if (acc && !join && chain_full_with(acc, opnd)) {
        KC_TRY(plane_force(acc));
        acc = nullptr;
    }
    ChainLink *L = new ChainLink();
    if (acc) {
        L->n_in = acc->link->n_in;
        for (int i = 0; i < L->n_in && i <= KC_CHAIN_MAX_IN; ++i) L->ins[i] = acc->link->ins[i];
        if (join) {
            const ChainLink &B = *opnd->link;
            for (int i = 0; i < B.n_in && i <= KC_CHAIN_MAX_IN; ++i) {
                bool seen = false;
                for (int k = 0; k < L->n_in && k <= KC_CHAIN_MAX_IN; ++k) seen |= L->ins[k] == B.ins[i];
                if (!seen) {
                    if (L->n_in <= KC_CHAIN_MAX_IN) L->ins[L->n_in] = B.ins[i];
                    L->n_in++;
                }
            }
        } else {
            link_add_input(*L, opnd);
        }
    }
    if (acc) {
        L->prev = acc;
        plane_retain(acc);
        L->step = { code_for(mix, acc_is_left), opnd };
        plane_retain(opnd);
        // records, as chain_fill will write them: "constant - acc" right after a {+, -, *} step on a plane is part of that
        // step's record (CH_*_INV)
        const ChainStep &last = acc->link->step;
        const uint8_t lc = last.code == CH_ADD_R ? (uint8_t)CH_ADD : last.code == CH_MUL_R ? (uint8_t)CH_MUL : last.code;
        const bool fuses = !join && L->step.code == CH_SUB_R && opnd->kind == kc_plane::CONST && lc <= CH_MUL &&
                           last.operand->kind != kc_plane::CONST && !acc->link->fused;
        L->fused = fuses;
        L->length = acc->link->length + (join ? opnd->link->length + 2u : fuses ? 0u : 1u);
        L->saved = join ? (uint8_t)std::max<int>(acc->link->saved, opnd->link->saved + 1) : acc->link->saved;
    } else {
        L->prev = nullptr;
        L->start = l;
        plane_retain(l);
        L->step = { code_for(mix, true), r };
        plane_retain(r);
        L->length = 1;
        L->n_in = 0;
        link_add_input(*L, l);
        link_add_input(*L, r);
    }
    kc_plane *res = new kc_plane();
    res->w = l->w;
    res->h = l->h;
    res->kind = kc_plane::LAZY;
    res->link = L;
    *out = res;
    return KC_OK;
}

// ------------------------------------------------------------------------------------------
// Images
// ------------------------------------------------------------------------------------------
kc_image *image_new(int n, kc_plane *const *planes)
{
    kc_image *img = new kc_image();
    img->n = n;
    for (int i = 0; i < n; ++i) {
        img->planes[i] = planes[i];
        plane_retain(planes[i]);
    }
    return img;
}

void image_release(kc_image *img)
{
    if (!img) return;
    if (--img->refs == 0) {
        for (int i = 0; i < img->n; ++i) plane_release(img->planes[i]);
        delete img;
    }
}

int image_force(kc_image *img) { return planes_force(img->planes, img->n); }

}  // namespace kc
