// tests/simt_emu/hip/hip_runtime.h — TEST INFRASTRUCTURE, never part of the product: a functional stand-in for <hip/hip_runtime.h> that lets
// g++ compile spades_amd/csrc/smx_api.hip (the whole library: C ABI, host pipeline AND its gfx950 kernels as written) into a host library,
// tests/simt_emu/_build/libspades_emu.so, so that kernel LOGIC can be run where there is no GPU (this container has none). It says nothing
// about speed, memory ordering or anything else the hardware decides: the GPU tier stays the authority, and nothing in spades_amd/ ever loads
// this library (tests/test_abi_cpu.py enforces it) — the product still fails loudly without a GPU.
//
// Execution model. A workgroup's threads are fibers on one host thread. A fiber runs until it reaches a synchronisation point:
//   * a workgroup barrier (__syncthreads, the library's lds_barrier): released when every thread that has not returned has arrived;
//   * a wave collective (__ballot, __shfl*, wave barrier): 64 consecutive threads form a wave, as on gfx950. A collective completes when
//     every lane of the wave that has not returned is blocked — the lanes waiting at this collective (same call site) are its active
//     lanes, the others wait at a workgroup barrier further on (a loop they have left, a branch they did not take): exactly the exec-mask
//     semantics of structured code. Lanes of one wave blocked at DIFFERENT collectives at once are reported as an error (no kernel of
//     this library does that).
// Workgroups run one after the other in index order (a look-back scan only ever waits for lower-numbered groups), atomics are plain
// read-modify-writes, `__shared__` is static storage (one group at a time), dynamic LDS one 160 KB buffer.
// The HIP runtime calls the host code makes are synchronous host equivalents: device memory is host memory (the virtual-memory
// management calls of the arena map anonymous pages into a reserved range), streams and events are ordering-free.
#pragma once
#include <sys/mman.h>
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <string>
#include <tuple>
#include <type_traits>
#include <vector>

// ---------------------------------------------------------------- language
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static

struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint2 { unsigned x, y; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
struct alignas(16) ulonglong2 { unsigned long long x, y; };
struct alignas(32) ulonglong4 { unsigned long long x, y, z, w; };
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
static inline ulonglong2 make_ulonglong2(unsigned long long x, unsigned long long y) { return ulonglong2{x, y}; }
static inline ulonglong4 make_ulonglong4(unsigned long long x, unsigned long long y, unsigned long long z, unsigned long long w) { return ulonglong4{x, y, z, w}; }

namespace emu {

constexpr unsigned WAVE = 64;
constexpr size_t LDS_BYTES = 160 * 1024, STACK_BYTES = 256 * 1024;
enum State { RUN, AT_BARRIER, AT_COLLECTIVE, DONE };
enum Op { BALLOT, SHFL_IDX, SHFL_UP, SHFL_DOWN, SHFL_XOR, WAVE_SYNC };

struct Lane {
    void *sp = nullptr;
    char *stack = nullptr;
    State state = DONE;
    const void *site = nullptr;
    Op op = BALLOT;
    uint64_t val = 0;   // predicate / value offered
    int arg = 0;        // source lane / delta / lane mask
    uint64_t res = 0;   // what the collective returns to this lane
};

struct Ctx {
    std::vector<Lane> lanes;
    std::vector<unsigned> order;  // the order in which the runnable threads of a workgroup take their turns (identity, or shuffled: below)
    unsigned pos = 0;             // position of the running thread in `order`
    uint64_t rng = 0;             // SMX_EMU_SHUFFLE=seed: xorshift state, 0 = off
    unsigned cur = 0, nthreads = 0;
    void *sched_sp = nullptr;
    const std::function<void()> *body = nullptr;
    dim3 tid, bid, bdim, gdim;
    size_t lds_used = 0;  // dynamic LDS the launch asked for (what SMX_EMU_POISON fills)
    // dynamic LDS: the bytes a launch asked for END at a no-access page, so a kernel that uses more than it requested faults here instead of
    // trampling its neighbours' LDS on the hardware (base is 16-byte aligned: an overrun of less than that goes unseen)
    unsigned char *lds_area = nullptr, *lds = nullptr;
    void place_lds(size_t bytes) {
        if (!lds_area) {
            lds_area = (unsigned char *)mmap(nullptr, LDS_BYTES + 2 * 4096, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            if (lds_area == (unsigned char *)MAP_FAILED || mprotect(lds_area + LDS_BYTES + 4096, 4096, PROT_NONE) != 0) {
                perror("emu: LDS area");
                abort();
            }
        }
        lds_used = bytes;
        lds = lds_area + LDS_BYTES + 4096 - ((bytes + 15) & ~(size_t)15);
    }
};
inline Ctx *g_ctx = new Ctx;  // (one host thread drives the library in the tests)
inline Ctx &ctx() { return *g_ctx; }

extern "C" void emu_ctx_switch(void **save_sp, void *load_sp);
#ifdef EMU_DEFINE_SWITCH
asm(R"(
.text
.globl emu_ctx_switch
.type emu_ctx_switch,@function
emu_ctx_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size emu_ctx_switch,.-emu_ctx_switch
)");
#endif

inline void set_tid(unsigned t) {
    Ctx &c = ctx();
    if (c.bdim.y == 1 && c.bdim.z == 1) {
        c.tid.x = t;
        return;
    }
    c.tid.x = t % c.bdim.x;
    c.tid.y = (t / c.bdim.x) % c.bdim.y;
    c.tid.z = t / (c.bdim.x * c.bdim.y);
}
// The running fiber has just blocked (or finished): on to the next thread that can run, directly — the scheduler is entered only when
// nobody can (one context switch per thread and synchronisation point instead of two).
inline void yield_to_scheduler() {
    Ctx &c = ctx();
    const unsigned me = c.cur;
    for (unsigned p = c.pos + 1; p < c.nthreads; ++p) {
        const unsigned t = c.order[p];
        if (c.lanes[t].state == RUN) {
            c.pos = p;
            c.cur = t;
            set_tid(t);
            emu_ctx_switch(&c.lanes[me].sp, c.lanes[t].sp);
            return;  // resumed: whoever switched here has set cur and the thread index again
        }
    }
    emu_ctx_switch(&c.lanes[me].sp, c.sched_sp);
}
inline void fiber_main() {
    Ctx &c = ctx();
    (*c.body)();
    c.lanes[c.cur].state = DONE;
    yield_to_scheduler();
    abort();  // a finished fiber is never resumed
}
inline void prepare(Lane &l) {
    if (!l.stack) {
        l.stack = (char *)mmap(nullptr, STACK_BYTES, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (l.stack == (char *)MAP_FAILED) {
            perror("emu: fiber stack");
            abort();
        }
    }
    uintptr_t top = ((uintptr_t)l.stack + STACK_BYTES) & ~(uintptr_t)15;
    void **s = (void **)top;
    *--s = nullptr;                // the trampoline's (never used) return address: keeps rsp % 16 == 8 at its entry
    *--s = (void *)&fiber_main;    // where the first switch "returns" to
    for (int i = 0; i < 6; ++i) *--s = nullptr;  // rbp rbx r12 r13 r14 r15
    l.sp = (void *)s;
    l.state = RUN;
}

[[noreturn]] inline void die(const char *what) {
    fprintf(stderr, "simt_emu: %s\n", what);
    abort();
}

inline void resolve(Ctx &c, unsigned w0, unsigned w1, const void *site) {
    // active lanes of this collective: the lanes of the wave that wait at `site`
    uint64_t active = 0, pred = 0;
    for (unsigned t = w0; t < w1; ++t)
        if (c.lanes[t].state == AT_COLLECTIVE && c.lanes[t].site == site) {
            active |= 1ull << (t - w0);
            if (c.lanes[t].val) pred |= 1ull << (t - w0);
        }
    for (unsigned t = w0; t < w1; ++t) {
        Lane &l = c.lanes[t];
        if (!(l.state == AT_COLLECTIVE && l.site == site)) continue;
        const int lane = (int)(t - w0);
        int src = lane;
        switch (l.op) {
            case BALLOT: l.res = pred; break;
            case WAVE_SYNC: l.res = 0; break;
            case SHFL_IDX: src = l.arg & 63; break;
            case SHFL_UP: src = lane - l.arg; break;
            case SHFL_DOWN: src = lane + l.arg; break;
            case SHFL_XOR: src = lane ^ l.arg; break;
        }
        if (l.op != BALLOT && l.op != WAVE_SYNC) {
            if (src < 0 || src >= (int)(w1 - w0) || !((active >> src) & 1)) src = lane;  // out of range / inactive source: the lane's own value
            l.res = c.lanes[w0 + src].val;
        }
    }
    for (unsigned t = w0; t < w1; ++t)
        if (c.lanes[t].state == AT_COLLECTIVE && c.lanes[t].site == site) c.lanes[t].state = RUN;
}

inline void run_block(const std::function<void()> &body) {
    Ctx &c = ctx();
    c.body = &body;
    const unsigned n = c.nthreads;
    if (c.lanes.size() < n) c.lanes.resize(n);
    for (unsigned t = 0; t < n; ++t) prepare(c.lanes[t]);
    // SMX_EMU_POISON=1: a workgroup finds garbage in its dynamic LDS, as on the hardware (here the previous group's bytes would still be
    // there, and a kernel that counts on them — or on zeros — would pass by accident)
    static const bool poison = getenv("SMX_EMU_POISON") != nullptr;
    if (poison && c.lds_used) memset(c.lds, 0xCD, c.lds_used);
    if (c.order.size() != n) {
        c.order.resize(n);
        for (unsigned t = 0; t < n; ++t) c.order[t] = t;
    }
    for (;;) {
        // SMX_EMU_SHUFFLE=seed: between two synchronisation points the threads take their turns in a fresh random order each time — the
        // hardware promises no order either, so whatever a kernel's RESULT owes to the order in which its lanes reach an atomic shows up as
        // a difference between runs (the library claims there is none: the order of the survivors never reaches the output)
        if (c.rng) {
            for (unsigned i = n - 1; i > 0; --i) {
                c.rng ^= c.rng << 13;
                c.rng ^= c.rng >> 7;
                c.rng ^= c.rng << 17;
                std::swap(c.order[i], c.order[c.rng % (i + 1)]);
            }
        }
        bool ran = false;
        for (unsigned p = 0; p < n; ++p) {
            const unsigned t = c.order[p];
            if (c.lanes[t].state != RUN) continue;
            ran = true;
            c.pos = p;
            c.cur = t;
            set_tid(t);
            emu_ctx_switch(&c.sched_sp, c.lanes[t].sp);
        }
        if (ran) continue;
        // everybody is blocked or done
        bool all_done = true, any = false;
        for (unsigned t = 0; t < n; ++t) all_done = all_done && c.lanes[t].state == DONE;
        if (all_done) break;
        for (unsigned w0 = 0; w0 < n; w0 += WAVE) {
            const unsigned w1 = std::min(n, w0 + WAVE);
            const void *site = nullptr;
            bool mixed = false;
            for (unsigned t = w0; t < w1; ++t)
                if (c.lanes[t].state == AT_COLLECTIVE) {
                    if (!site) site = c.lanes[t].site;
                    else if (site != c.lanes[t].site) mixed = true;
                }
            if (!site) continue;
            if (mixed) die("lanes of one wave wait at different wave collectives at once (divergent collectives): not modelled");
            resolve(c, w0, w1, site);
            any = true;
        }
        if (any) continue;
        // a workgroup barrier: every thread that has not returned waits at it
        for (unsigned t = 0; t < n; ++t)
            if (c.lanes[t].state == AT_BARRIER) c.lanes[t].state = RUN;
    }
}

inline void barrier() {
    Ctx &c = ctx();
    c.lanes[c.cur].state = AT_BARRIER;
    yield_to_scheduler();
}
inline uint64_t collective(Op op, uint64_t val, int arg, const void *site) {
    Ctx &c = ctx();
    Lane &l = c.lanes[c.cur];
    l.state = AT_COLLECTIVE;
    l.op = op;
    l.val = val;
    l.arg = arg;
    l.site = site;
    yield_to_scheduler();
    return ctx().lanes[ctx().cur].res;
}

// SMX_EMU_STATS=1: per kernel (as spelled at the launch) launches, workgroups and host seconds, printed at exit
struct Stats {
    struct Row { uint64_t launches = 0, blocks = 0; double s = 0; };
    std::map<std::string, Row> rows;
    bool on = getenv("SMX_EMU_STATS") != nullptr;
    ~Stats() {
        if (!on) return;
        std::vector<std::pair<double, std::string>> v;
        for (auto &r : rows) v.push_back({r.second.s, r.first});
        std::sort(v.rbegin(), v.rend());
        for (size_t i = 0; i < v.size() && i < 25; ++i)
            fprintf(stderr, "[emu] %8.2f s %8llu launches %10llu groups  %s\n", v[i].first, (unsigned long long)rows[v[i].second].launches,
                    (unsigned long long)rows[v[i].second].blocks, v[i].second.c_str());
    }
};
inline Stats g_stats;

template <class F>
inline void launch(dim3 grid, dim3 block, const F &f, const char *name = "") {
    const auto t0 = std::chrono::steady_clock::now();
    struct Done {
        const char *name; dim3 grid; std::chrono::steady_clock::time_point t0;
        ~Done() {
            if (!g_stats.on) return;
            auto &r = g_stats.rows[name];
            r.launches++;
            r.blocks += (uint64_t)grid.x * grid.y * grid.z;
            r.s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        }
    } done{name, grid, t0};
    Ctx &c = ctx();
    static const uint64_t seed = getenv("SMX_EMU_SHUFFLE") ? (uint64_t)atoll(getenv("SMX_EMU_SHUFFLE")) * 0x9E3779B97F4A7C15ull + 1 : 0;
    if (seed && !c.rng) c.rng = seed;
    c.gdim = grid;
    c.bdim = block;
    c.nthreads = block.x * block.y * block.z;
    if (c.nthreads == 0 || c.nthreads > 1024) die("bad block size");
    const std::function<void()> body = f;
    // SMX_EMU_SHUFFLE_GROUPS=1 (with SMX_EMU_SHUFFLE): the workgroups of a launch run in a random order as well — the hardware dispatches them
    // in no promised order, and everything the library numbers by atomics (chunk ids, places in the output, node ids of route 0) comes out
    // differently; the RESULTS must not. (No kernel of the library waits for another workgroup, so any order completes.)
    static const bool shuffle_groups = getenv("SMX_EMU_SHUFFLE_GROUPS") != nullptr;
    const uint64_t ngroups = (uint64_t)grid.x * grid.y * grid.z;
    std::vector<uint32_t> perm;
    if (shuffle_groups && c.rng && ngroups > 1 && ngroups < (1ull << 31)) {
        perm.resize(ngroups);
        for (uint64_t i = 0; i < ngroups; ++i) perm[i] = (uint32_t)i;
        for (uint64_t i = ngroups - 1; i > 0; --i) {
            c.rng ^= c.rng << 13;
            c.rng ^= c.rng >> 7;
            c.rng ^= c.rng << 17;
            std::swap(perm[i], perm[c.rng % (i + 1)]);
        }
    }
    for (uint64_t g = 0; g < ngroups; ++g) {
        const uint64_t id = perm.empty() ? g : perm[g];
        c.bid = dim3((unsigned)(id % grid.x), (unsigned)((id / grid.x) % grid.y), (unsigned)(id / ((uint64_t)grid.x * grid.y)));
        run_block(body);
    }
}

template <class T>
inline T shfl(Op op, T v, int arg, const void *site) {
    static_assert(sizeof(T) <= 8, "shuffles of up to 64 bits");
    uint64_t bits = 0;
    memcpy(&bits, &v, sizeof(T));
    bits = collective(op, bits, arg, site);
    T r;
    memcpy(&r, &bits, sizeof(T));
    return r;
}

}  // namespace emu

#define threadIdx (emu::g_ctx->tid)
#define blockIdx (emu::g_ctx->bid)
#define blockDim (emu::g_ctx->bdim)
#define gridDim (emu::g_ctx->gdim)
#define warpSize 64

// a collective's call site: the address of a static of its own (one per textual use; template instantiations share the line, which is fine)
#define EMU_SITE ([]() -> const void * { static const char s = 0; return &s; }())
#define __syncthreads() emu::barrier()
#define __ballot(p) emu::collective(emu::BALLOT, (p) ? 1 : 0, 0, EMU_SITE)
#define __shfl(v, src, ...) emu::shfl(emu::SHFL_IDX, (v), (int)(src), EMU_SITE)
#define __shfl_up(v, d, ...) emu::shfl(emu::SHFL_UP, (v), (int)(d), EMU_SITE)
#define __shfl_down(v, d, ...) emu::shfl(emu::SHFL_DOWN, (v), (int)(d), EMU_SITE)
#define __shfl_xor(v, m, ...) emu::shfl(emu::SHFL_XOR, (v), (int)(m), EMU_SITE)
#define EMU_WAVE_BARRIER() ((void)emu::collective(emu::WAVE_SYNC, 0, 0, EMU_SITE))
#define EMU_LDS_BARRIER() emu::barrier()

// HIP's device-side min / max (global namespace, mixed integer types)
template <class A, class B> static inline typename std::common_type<A, B>::type min(A a, B b) { using T = typename std::common_type<A, B>::type; return (T)a < (T)b ? (T)a : (T)b; }
template <class A, class B> static inline typename std::common_type<A, B>::type max(A a, B b) { using T = typename std::common_type<A, B>::type; return (T)a < (T)b ? (T)b : (T)a; }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __ffs(int v) { return __builtin_ffs(v); }
static inline int __ffs(unsigned v) { return __builtin_ffs((int)v); }
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
static inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
static inline int __clz(unsigned v) { return v ? __builtin_clz(v) : 32; }
static inline int __clzll(unsigned long long v) { return v ? __builtin_clzll(v) : 64; }
template <typename T>
static inline void emu_nt_store(T v, T *p) { *p = v; }
template <typename T>
static inline T emu_nt_load(const T *p) { return *p; }
static inline unsigned __brev(unsigned v) {
    v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
    v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
    v = ((v >> 4) & 0x0F0F0F0Fu) | ((v & 0x0F0F0F0Fu) << 4);
    return __builtin_bswap32(v);
}
static inline unsigned long long __brevll(unsigned long long v) { return ((unsigned long long)__brev((unsigned)v) << 32) | __brev((unsigned)(v >> 32)); }
static inline unsigned long long __umul64hi(unsigned long long a, unsigned long long b) { return (unsigned long long)(((unsigned __int128)a * b) >> 64); }
static inline unsigned emu_rotl32(unsigned v, unsigned r) { r &= 31; return r ? (v << r) | (v >> (32 - r)) : v; }  // (the build replaces __builtin_rotateleft32)
static inline unsigned long long wall_clock64() { return (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count() / 10; }

// atomics: the fibers of a workgroup take turns and workgroups run one after the other
template <class T, class U> static inline T atomicAdd(T *p, U v) { T o = *p; *p = (T)(o + (T)v); return o; }
template <class T, class U> static inline T atomicOr(T *p, U v) { T o = *p; *p = (T)(o | (T)v); return o; }
template <class T, class U> static inline T atomicAnd(T *p, U v) { T o = *p; *p = (T)(o & (T)v); return o; }
template <class T, class U> static inline T atomicMax(T *p, U v) { T o = *p; if ((T)v > o) *p = (T)v; return o; }
template <class T, class U> static inline T atomicMin(T *p, U v) { T o = *p; if ((T)v < o) *p = (T)v; return o; }
template <class T, class U> static inline T atomicExch(T *p, U v) { T o = *p; *p = (T)v; return o; }
template <class T, class U, class V> static inline T atomicCAS(T *p, U cmp, V v) { T o = *p; if (o == (T)cmp) *p = (T)v; return o; }

// ---------------------------------------------------------------- runtime
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorInvalidValue = 1 };
typedef void *hipStream_t;
typedef void *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum { hipHostMallocDefault = 0, hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
typedef void *hipDeviceptr_t;
struct EmuAlloc { size_t size; };
typedef EmuAlloc *hipMemGenericAllocationHandle_t;
enum { hipMemAllocationTypePinned = 1, hipMemLocationTypeDevice = 1, hipMemAccessFlagsProtReadWrite = 3 };
struct hipMemLocation { int type; int id; };
struct hipMemAllocationProp { int type; int requestedHandleTypes; hipMemLocation location; void *win32HandleMetaData; struct { unsigned char c, g; unsigned short u; unsigned char r[4]; } allocFlags; };
struct hipMemAccessDesc { hipMemLocation location; int flags; };

namespace emu {
inline size_t device_bytes() {
    if (const char *e = getenv("SMX_EMU_DEVICE_MB")) return (size_t)atoll(e) << 20;
    return (size_t)6 << 30;  // the "device memory" the arena sees
}
inline size_t &in_use() { static size_t v = 0; return v; }
}  // namespace emu

static inline const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : e == hipErrorOutOfMemory ? "out of memory (emulated)" : "error (emulated)"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipMemGetInfo(size_t *fr, size_t *tot) { *tot = emu::device_bytes(); *fr = *tot > emu::in_use() ? *tot - emu::in_use() : 0; return hipSuccess; }
static inline hipError_t hipStreamCreate(hipStream_t *s) { *s = (void *)1; return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = (void *)1; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
struct EmuEvent { double t; };
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = new EmuEvent{0}; return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete (EmuEvent *)e; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr) { ((EmuEvent *)e)->t = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(((EmuEvent *)b)->t - ((EmuEvent *)a)->t); return hipSuccess; }
template <class T> static inline hipError_t hipMalloc(T **p, size_t n) {
    if (emu::in_use() + n > emu::device_bytes()) { *p = nullptr; return hipErrorOutOfMemory; }
    void *q = nullptr;
    if (posix_memalign(&q, 256, std::max<size_t>(n, 1)) != 0) { *p = nullptr; return hipErrorOutOfMemory; }
    if (getenv("SMX_EMU_POISON")) memset(q, 0xCD, std::max<size_t>(n, 1));
    *p = (T *)q;
    return hipSuccess;
}
static inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
template <class T> static inline hipError_t hipHostMalloc(T **p, size_t n, unsigned = 0) { *p = (T *)malloc(std::max<size_t>(n, 1)); return *p ? hipSuccess : hipErrorOutOfMemory; }
static inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { if (n) memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { if (n) memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemset(void *d, int v, size_t n) { if (n) memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t = nullptr) { if (n) memset(d, v, n); return hipSuccess; }
static inline hipError_t hipFuncSetAttribute(const void *, int, int) { return hipSuccess; }
// virtual memory management (the arena of smx_ctx.hpp): a reserved range of no-access pages, chunks mapped into it on demand
static inline hipError_t hipMemAddressReserve(void **p, size_t n, size_t, void *, unsigned long long) {
    void *q = mmap(nullptr, n, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (q == MAP_FAILED) return hipErrorOutOfMemory;
    *p = q;
    return hipSuccess;
}
static inline hipError_t hipMemAddressFree(void *p, size_t n) { munmap(p, n); return hipSuccess; }
static inline hipError_t hipMemCreate(hipMemGenericAllocationHandle_t *h, size_t n, const hipMemAllocationProp *, unsigned long long) {
    if (emu::in_use() + n > emu::device_bytes()) return hipErrorOutOfMemory;
    emu::in_use() += n;
    *h = new EmuAlloc{n};
    return hipSuccess;
}
static inline hipError_t hipMemRelease(hipMemGenericAllocationHandle_t h) { emu::in_use() -= h->size; delete h; return hipSuccess; }
static inline hipError_t hipMemMap(void *p, size_t n, size_t, hipMemGenericAllocationHandle_t, unsigned long long) {
    void *q = mmap(p, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_FIXED | MAP_NORESERVE, -1, 0);
    if (q == MAP_FAILED) return hipErrorOutOfMemory;
    // SMX_EMU_POISON=1: freshly mapped device memory holds garbage, as VRAM does (anonymous pages would be zeros: a kernel that reads what
    // nobody wrote would get away with it here). Use small arena chunks (SMX_ARENA_CHUNK_MB=2) with it: every mapped byte is touched.
    static const bool poison = getenv("SMX_EMU_POISON") != nullptr;
    if (poison) memset(q, 0xCD, n);
    return hipSuccess;
}
static inline hipError_t hipMemUnmap(void *p, size_t n) {
    void *q = mmap(p, n, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_FIXED | MAP_NORESERVE, -1, 0);
    return q == MAP_FAILED ? hipErrorInvalidValue : hipSuccess;
}
static inline hipError_t hipMemSetAccess(void *, size_t, const hipMemAccessDesc *, size_t) { return hipSuccess; }

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...)                                   \
    do {                                                                                              \
        if ((size_t)(shmem) > emu::LDS_BYTES) emu::die("dynamic LDS request beyond 160 KB");          \
        emu::g_ctx->place_lds((size_t)(shmem));                                                       \
        auto emu_args_ = std::make_tuple(__VA_ARGS__);                                                \
        emu::launch(dim3(grid), dim3(block), [&]() { std::apply([&](auto &...a_) { kernel(a_...); }, emu_args_); }, #kernel); \
    } while (0)
