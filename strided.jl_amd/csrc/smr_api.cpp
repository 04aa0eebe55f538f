// smr_api.cpp -- the C ABI of libstrided_hip.so (include/strided_hip.h): error reporting,
// device/memory helpers, plan cache, execution dispatch, sharding.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <list>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "smr_dispatch.h"

namespace smr {

static thread_local std::string g_last_error;

int set_error(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}
int hip_error(hipError_t e, const char* what) {
    g_last_error = std::string(what) + ": " + hipGetErrorString(e);
    return e == hipErrorOutOfMemory ? SMR_ENOMEM : (e == hipErrorNoDevice ? SMR_ENODEVICE : SMR_EHIP);
}

// ---- per-type dispatch ---------------------------------------------------------------------------
#define SMR_DISPATCH_CT(fn)                                                   \
    switch (plan.c.bitcopy ? SMR_F32 : plan.c.ct) {                           \
        case SMR_F32: return fn##_ct<SMR_F32>(plan, bases, s);                \
        case SMR_F64: return fn##_ct<SMR_F64>(plan, bases, s);                \
        case SMR_C32: return fn##_ct<SMR_C32>(plan, bases, s);                \
        case SMR_C64: return fn##_ct<SMR_C64>(plan, bases, s);                \
        case SMR_I64: return fn##_ct<SMR_I64>(plan, bases, s);                \
    }                                                                         \
    return set_error(SMR_EINVAL, "bad compute class");

int launch_generic_map(const Plan& plan, void* const* bases, hipStream_t s) { SMR_DISPATCH_CT(launch_generic_map) }
int launch_stream_map(const Plan& plan, void* const* bases, hipStream_t s) { SMR_DISPATCH_CT(launch_stream_map) }
int launch_tiled_map(const Plan& plan, void* const* bases, hipStream_t s) { SMR_DISPATCH_CT(launch_tiled_map) }
int launch_reduce_all(const Plan& plan, void* const* bases, hipStream_t s) { SMR_DISPATCH_CT(launch_reduce_all) }
int launch_reduce_part(const Plan& plan, void* const* bases, hipStream_t s) { SMR_DISPATCH_CT(launch_reduce_part) }
int launch_orbit_map(const Plan& plan, void* const* bases, hipStream_t s) { SMR_DISPATCH_CT(launch_orbit_map) }
int launch_flat_map(const Plan& plan, void* const* bases, hipStream_t s) { SMR_DISPATCH_CT(launch_flat_map) }

// ---- overlap window: dependency-aware launch order --------------------------------------------------------------------
// The reference runs the independent halves of a problem as concurrent tasks and waits only where it must
// (src/mapreduce.jl:203-223: spawn one half, run the other, wait).  On the GPU the analogous unit is the LAUNCH: a 16 MiB
// launch here is a span of 1.6-3.2 us followed by a 1.6-1.9 us boundary (drain, write-back, next dispatch -- profiles/
// r03_device_span.txt) that a stream-ordered successor pays even when it touches none of the predecessor's data.  Inside an
// overlap window (smr_overlap_begin / _end, or any stream made by smr_stream_create) the library therefore keeps, per stream,
// the byte ranges read and written by the launches that went out since the last ORDERED one; a new launch whose ranges do
// not conflict with them (no read-after-write, write-after-write or write-after-read) is dispatched without the AQL barrier
// bit (hipExtAnyOrderLaunch) -- its waves start while the earlier launches are still running or draining -- anything else
// goes out in order and becomes the new window base.  Closing the window (and, on library-owned streams, every copy /
// synchronisation the library performs) first issues one ordered empty kernel, so that "the stream's last command has
// completed" again implies "everything before it has" for whatever follows (events, copies, hipStreamSynchronize).
namespace {
struct Span {
    uintptr_t lo, hi;
};
struct Window {
    int depth = 0;        // nesting of smr_overlap_begin
    bool owned = false;   // smr_stream_create: permanently open
    bool loose = false;   // the last launch went out without the barrier bit: a fence is due before foreign work
    int unordered = 0;    // launches since the window base
    std::vector<Span> reads, writes;
};
struct Windows {
    std::mutex mu;
    std::unordered_map<void*, Window> map;
    long stat_any = 0, stat_ordered = 0, stat_fences = 0;
};
Windows& windows() {
    static Windows* w = new Windows();
    return *w;
}
thread_local unsigned tl_launch_flags = 0;
constexpr int WINDOW_CAP = 32;  // launches tracked before the window is re-based by an ordered launch

__global__ void k_window_fence() {}

bool overlaps(const std::vector<Span>& v, const Span& x) {
    for (const Span& y : v)
        if (x.lo < y.hi && y.lo < x.hi) return true;
    return false;
}

// byte ranges one execution of `plan` reads and writes (every operand's bounding range; the plan's own partials)
void footprint(const Plan& plan, void* const* bases, std::vector<Span>& rd, std::vector<Span>& wr) {
    const Canon& c = plan.c;
    for (int k = 0; k < c.M; ++k) {
        i64 lo = c.offsets[k], hi = c.offsets[k];
        for (int d = 0; d < c.N; ++d) {
            const i64 ext = (c.dims[d] - 1) * c.strides[k][d];
            (ext < 0 ? lo : hi) += ext;
        }
        const uintptr_t b = (uintptr_t)(bases ? bases[c.orig[k]] : c.base[k]);
        const Span sp{b + (uintptr_t)(lo * (i64)c.esize[k]), b + (uintptr_t)((hi + 1) * (i64)c.esize[k])};
        if (k == 0) {
            wr.push_back(sp);
            if (c.redop != SMR_RED_NONE) rd.push_back(sp);  // reductions accumulate into the destination
        } else {
            rd.push_back(sp);
        }
    }
    if (plan.scratch) {
        const Span sp{(uintptr_t)plan.scratch, (uintptr_t)plan.scratch + plan.counter_off + RED_COUNTERS * sizeof(unsigned)};
        wr.push_back(sp);
        rd.push_back(sp);
    }
}

// decides the ordering of the execution about to be launched on `s`
void window_admit(const Plan& plan, void* const* bases, hipStream_t s) {
    tl_launch_flags = 0;
    // HIP accepts hipExtAnyOrderLaunch and IGNORES it on gfx9 (hip_ext.h says so; device stamps confirm it, profiles/r04_overlap.txt):
    // on gfx942 / gfx950 a window on a stream that launches through HIP cannot change anything.  The analysis below -- two mutex
    // acquisitions, a footprint with vector allocations per launch, a fence kernel -- therefore runs only on request (option
    // "overlap_window_hip" = 1, for a runtime that honours the flag); by default the window API is a no-op for HIP launches, and
    // independent launches overlap where the library dispatches itself: library-owned streams and recorded sequences (smr_seq.cpp).
    if (!options().overlap_window_hip) return;
    Windows& W = windows();
    std::lock_guard<std::mutex> g(W.mu);
    if (W.map.empty()) return;
    auto it = W.map.find((void*)s);
    if (it == W.map.end() || (it->second.depth == 0 && !it->second.owned)) return;
    Window& w = it->second;
    std::vector<Span> rd, wr;
    footprint(plan, bases, rd, wr);
    bool free_ = w.unordered > 0 && w.unordered < WINDOW_CAP;
    for (size_t i = 0; free_ && i < wr.size(); ++i) free_ = !overlaps(w.reads, wr[i]) && !overlaps(w.writes, wr[i]);
    for (size_t i = 0; free_ && i < rd.size(); ++i) free_ = !overlaps(w.writes, rd[i]);
    if (free_) {
        tl_launch_flags = hipExtAnyOrderLaunch;
        w.loose = true;
        ++W.stat_any;
    } else {
        w.reads.clear();
        w.writes.clear();
        w.unordered = 0;
        w.loose = false;
        ++W.stat_ordered;
    }
    ++w.unordered;
    w.reads.insert(w.reads.end(), rd.begin(), rd.end());
    w.writes.insert(w.writes.end(), wr.begin(), wr.end());
}

// one ordered empty kernel: everything launched before it has completed when it has
int window_fence_locked(Windows& W, Window& w, hipStream_t s) {
    w.reads.clear();
    w.writes.clear();
    w.unordered = 0;
    if (!w.loose) return SMR_OK;
    w.loose = false;
    ++W.stat_fences;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_window_fence, dim3(1), dim3(64), 0, s);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? SMR_OK : hip_error(e, "overlap window fence");
}
int window_fence(hipStream_t s) {
    bool owned = false;
    int rc = SMR_OK;
    {
        Windows& W = windows();
        std::lock_guard<std::mutex> g(W.mu);
        auto it = W.map.find((void*)s);
        if (it == W.map.end()) return SMR_OK;
        owned = it->second.owned;
        rc = window_fence_locked(W, it->second, s);
    }
    if (owned && eager_available(s)) {  // what the library submitted directly completes before whatever follows on the stream through HIP ...
        const int rc2 = eager_fence_all();
        eager_note_hip_work(s);        // ... and that HIP work completes before the next direct launch ON THIS STREAM
        if (rc == SMR_OK) rc = rc2;
    }
    return rc;
}
}  // namespace

// for the other translation units (smr_comm.cpp: the RCCL collective is foreign work on the stream)
int fence_for_foreign_work(hipStream_t s) { return window_fence(s); }

static std::atomic<long> g_launches{0};
void count_launch() { g_launches.fetch_add(1, std::memory_order_relaxed); }

unsigned take_launch_flags() {
    const unsigned f = tl_launch_flags;
    tl_launch_flags = 0;
    return f;
}

static int execute_family(const Plan& plan, void* const* bases, hipStream_t s);

static bool stream_is_owned(hipStream_t s) {
    Windows& W = windows();
    std::lock_guard<std::mutex> g(W.mu);
    if (W.map.empty()) return false;
    auto it = W.map.find((void*)s);
    return it != W.map.end() && it->second.owned;
}

static int execute(const Plan& plan, void* const* bases, hipStream_t s) {
    if (!jit_no_launch() && !recorder()) {
        // a library-owned stream: the library submits the launch itself (smr_seq.cpp: eager direct dispatch), on the hardware queue its
        // data dependencies select -- independent executions run concurrently, host cost ~1 us instead of HIP's 3.6-4 us
        hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
        if (options().eager_direct && stream_is_owned(s) && eager_available(s) &&
            (hipStreamIsCapturing(s, &capturing) != hipSuccess || capturing == hipStreamCaptureStatusNone)) {  // (a capture records HIP launches)
            std::vector<Span> rd, wr;
            footprint(plan, bases, rd, wr);
            std::vector<RecLaunch> rec;
            // self-released launches (write-through stores, no release fence on the packet: smr_seq.cpp) while what this stream has
            // been writing lately fits the caches
            set_recorder(&rec, options().eager_self_release != 0 && !wr.empty() && eager_recent_writes_fit(wr[0].lo, wr[0].hi), true);
            int rc = execute_family(plan, bases, s);
            set_recorder(nullptr);
            if (rc) return rc;
            if (!__atomic_exchange_n(&plan.eager_seen, true, __ATOMIC_RELAXED)) eager_request_sys_acquire(s);  // (two threads may execute one plan)
            std::vector<std::pair<uintptr_t, uintptr_t>> r2, w2;
            for (const Span& x : rd) r2.emplace_back(x.lo, x.hi);
            for (const Span& x : wr) w2.emplace_back(x.lo, x.hi);
            rc = rec.empty() ? SMR_EUNSUPPORTED : eager_submit(plan, rec, r2, w2, s);
            if (rc != SMR_EUNSUPPORTED) return rc;
            // this execution goes through HIP (a kernel that needs scratch memory, ...): everything submitted directly comes first
            if (int rc3 = eager_fence_all()) return rc3;
            eager_note_hip_work(s);
        }
        window_admit(plan, bases, s);
    }
    const int rc = execute_family(plan, bases, s);
    tl_launch_flags = 0;  // an execution that launched nothing must not leak its flag to the next one
    return rc;
}

static int execute_family(const Plan& plan, void* const* bases, hipStream_t s) {
    switch (plan.family) {
        case FAM_GENERIC: return launch_generic_map(plan, bases, s);
        case FAM_STREAM: return launch_stream_map(plan, bases, s);
        case FAM_TILED: {
            std::lock_guard<std::mutex> g(*plan.build_mu);
            return launch_tiled_map(plan, bases, s);
        }
        case FAM_ORBIT: {
            std::lock_guard<std::mutex> g(*plan.build_mu);
            return launch_orbit_map(plan, bases, s);
        }
        case FAM_FLAT: return launch_flat_map(plan, bases, s);
        case FAM_REDUCE_ALL: return launch_reduce_all(plan, bases, s);
        case FAM_REDUCE_PART: return launch_reduce_part(plan, bases, s);
    }
    return set_error(SMR_EINVAL, "plan has no kernel family");
}

unsigned long long* stamp_next(size_t waves) {
    Options& o = options();
    if (!o.stamp_base || o.stamp_used + (i64)(2 * waves) > o.stamp_cap) return nullptr;
    unsigned long long* p = reinterpret_cast<unsigned long long*>((uintptr_t)o.stamp_base) + o.stamp_used;
    o.stamp_used += (i64)(2 * waves);
    return p;
}

static bool g_device_checked = false;
static int ensure_device() {
    if (g_device_checked) return SMR_OK;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return set_error(SMR_ENODEVICE, "libstrided_hip: no HIP device available (the HIP kernels are the only compute path; there is no CPU fallback)");
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess) {
            if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
                return set_error(SMR_ENODEVICE, std::string("libstrided_hip is built for gfx950 only, found ") + prop.gcnArchName);
        }
    }
    g_device_checked = true;
    return SMR_OK;
}

}  // namespace smr

using namespace smr;

struct smr_plan {
    Plan plan;
    int nops = 0;  // M of the originating problem (length of a `bases` rebinding array)
    void* stream = nullptr;
    void* obase[SMR_MAXM] = {nullptr};  // the base pointers the plan was created with (aliasing pattern)
    // plans with reduction scratch (partials + arrival counters) must not run twice at once: executions on different streams are
    // chained through an event
    std::mutex scratch_mu;
    bool ran = false;
    hipStream_t last_stream = nullptr;
    hipEvent_t order_ev = nullptr;
};

// Executes a plan that may own reduction scratch.  The one-launch fold (smr_k_reduce.hip: arrive_last) relies on the arrival
// counters being zero when a launch starts and on the partials belonging to ONE execution at a time:
//   * an execution on another stream than the previous one first waits (on the device) for that one -- two streams never run
//     the same plan's launches concurrently (skipped while either stream is being captured: a capture cannot wait for foreign work;
//     captured graphs replay in the order their streams impose);
//   * an execution that failed to launch leaves nothing half-done on the device, but the counters are re-zeroed on the stream all the
//     same, so that no earlier, aborted launch can leave a later one waiting for arrivals that never come.
static int execute_owned(smr_plan* h, void* const* bases, hipStream_t s) {
    if (!h->plan.scratch || jit_no_launch() || recorder()) return execute(h->plan, bases, s);
    std::lock_guard<std::mutex> g(h->scratch_mu);
    if (h->ran && h->last_stream != s) {
        hipStreamCaptureStatus c1 = hipStreamCaptureStatusNone, c2 = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(s, &c1);
        (void)hipStreamIsCapturing(h->last_stream, &c2);
        // launches the library submitted itself (a library-owned stream) are not HIP work: no event covers them, and the direct queues do
        // not honour hipStreamWaitEvent.  With such a stream on either side the previous execution is waited for on the host.
        if (c1 == hipStreamCaptureStatusNone && c2 == hipStreamCaptureStatusNone && (stream_is_owned(s) || stream_is_owned(h->last_stream))) {
            (void)eager_fence_if_active();
            (void)hipStreamSynchronize(h->last_stream);
        }
        if (c1 == hipStreamCaptureStatusNone && c2 == hipStreamCaptureStatusNone) {
            if (!h->order_ev) (void)hipEventCreateWithFlags(&h->order_ev, hipEventDisableTiming);
            if (h->order_ev && hipEventRecord(h->order_ev, h->last_stream) == hipSuccess) (void)hipStreamWaitEvent(s, h->order_ev, 0);
        }
        (void)hipGetLastError();
    }
    const int rc = execute(h->plan, bases, s);
    h->ran = true;
    h->last_stream = s;
    if (rc != SMR_OK) {
        (void)hipGetLastError();
        (void)hipMemsetAsync((char*)h->plan.scratch + h->plan.counter_off, 0, RED_COUNTERS * sizeof(unsigned), s);
        (void)hipGetLastError();
    }
    return rc;
}

static int plan_build(const smr_problem* p, smr_plan** out) {
    if (!p || !out) return set_error(SMR_EINVAL, "null argument");
    smr_plan* h = new (std::nothrow) smr_plan();
    if (!h) return set_error(SMR_ENOMEM, "out of host memory");
    int rc = make_plan(p, h->plan);
    if (rc) {
        delete h;
        return rc;
    }
    h->nops = p->M;
    h->stream = p->stream;
    for (int k = 0; k < p->M; ++k) h->obase[k] = p->ops[k].base;
    *out = h;
    return SMR_OK;
}

// reduction partials are allocated on first execution (planning itself needs no device)
static int ensure_scratch(smr_plan* h) {
    std::lock_guard<std::mutex> g(*h->plan.build_mu);
    if (h->plan.scratch || h->plan.scratch_bytes == 0 || h->plan.red_blocks <= 1) return SMR_OK;
    const size_t coff = (h->plan.scratch_bytes + 255) & ~(size_t)255;
    void* buf = nullptr;
    hipError_t e = hipMalloc(&buf, coff + RED_COUNTERS * sizeof(unsigned));
    if (e != hipSuccess) return hip_error(e, "hipMalloc(reduction partials)");
    e = hipMemset((char*)buf + coff, 0, RED_COUNTERS * sizeof(unsigned));  // the kernels leave the counters at zero
    if (e != hipSuccess) {
        (void)hipFree(buf);
        return hip_error(e, "hipMemset(reduction counters)");
    }
    h->plan.counter_off = coff;
    h->plan.scratch = buf;
    return SMR_OK;
}

static void plan_free(smr_plan* h) {
    if (!h) return;
    (void)eager_fence_if_active();  // launches submitted directly may still read the plan's tables
    if (h->plan.scratch) (void)hipFree(h->plan.scratch);
    for (void*& p : h->plan.lanetab)
        if (p) {
            (void)hipFree(p);
            p = nullptr;
        }
    if (h->plan.ordtab) (void)hipFree(h->plan.ordtab);
    if (h->order_ev) (void)hipEventDestroy(h->order_ev);
    delete h;
}

// ---- hooks for smr_seq.cpp (the smr_plan struct is private to this file) ------------------------------------------------
namespace smr {
int seq_execute_plan(smr_plan* plan, void* const* bases, hipStream_t s, bool prepare_only) {
    if (prepare_only) return smr_plan_prepare(plan);
    return smr_plan_execute(plan, bases, (void*)s);
}
void seq_footprint(smr_plan* plan, void* const* bases, std::vector<std::pair<uintptr_t, uintptr_t>>& rd, std::vector<std::pair<uintptr_t, uintptr_t>>& wr) {
    std::vector<Span> r, w;
    footprint(plan->plan, bases, r, w);
    for (const Span& x : r) rd.emplace_back(x.lo, x.hi);
    for (const Span& x : w) wr.emplace_back(x.lo, x.hi);
}
int seq_nops(smr_plan* plan) { return plan->nops; }
bool seq_stream_is_owned(hipStream_t s) { return stream_is_owned(s); }
}  // namespace smr

// ---- plan cache for the one-shot entry point ---------------------------------------------------------
// Plans are held by shared_ptr: a thread that found a plan keeps it alive while it executes, eviction only
// drops the cache's reference (the last owner frees the device tables -- after draining the plan's stream,
// outside the cache lock).  The key describes the problem's SHAPE, not its addresses: operands are
// identified by alias class (which operands share a base pointer) and 256-byte alignment, and the
// actual base pointers are bound at launch -- a loop over freshly allocated arrays hits one entry.
namespace {
struct PlanDeleter {
    void operator()(smr_plan* h) const {
        if (!h) return;
        if (h->plan.scratch || h->plan.ordtab || h->plan.lanetab[0] || h->plan.lanetab[1] || h->plan.lanetab[2] || h->plan.lanetab[3])
        {
            (void)window_fence((hipStream_t)h->stream);
            (void)hipStreamSynchronize((hipStream_t)h->stream);  // queued kernels may still read the tables
        }
        plan_free(h);
    }
};
typedef std::shared_ptr<smr_plan> PlanRef;

struct Cache {
    std::mutex mu;
    std::list<std::pair<std::string, PlanRef>> lru;
    std::unordered_map<std::string, std::list<std::pair<std::string, PlanRef>>::iterator> map;
    static constexpr size_t CAP = 512;
    ~Cache() {}
};
Cache& cache() {
    static Cache* c = new Cache();  // intentionally leaked: no HIP calls at process exit
    return *c;
}

std::string signature(const smr_problem* p) {
    std::string s;
    auto put = [&](const void* d, size_t n) { s.append((const char*)d, n); };
    put(&p->N, sizeof p->N);
    put(&p->M, sizeof p->M);
    put(p->dims, sizeof(int64_t) * (size_t)p->N);
    for (int k = 0; k < p->M; ++k) {
        // alias class = first operand with the same base; byte distance to it (views of one buffer may sit
        // at different offsets); alignment of the base itself
        int32_t cls = k;
        for (int j = 0; j < k; ++j)
            if (p->ops[j].base == p->ops[k].base) {
                cls = j;
                break;
            }
        const uint32_t align = (uint32_t)((uintptr_t)p->ops[k].base & 255u);
        // address class = first operand whose FIRST ELEMENT sits at the same address: the ORBIT / tile-order planners
        // recognise "views of one buffer" by that address, whatever the base pointers are
        int32_t acls = k;
        const char* ak = (const char*)p->ops[k].base + p->ops[k].offset * (int64_t)dtype_size(p->ops[k].dtype);
        for (int j = 0; j < k; ++j)
            if ((const char*)p->ops[j].base + p->ops[j].offset * (int64_t)dtype_size(p->ops[j].dtype) == ak) {
                acls = j;
                break;
            }
        put(&acls, sizeof acls);
        put(&cls, sizeof cls);
        put(&align, sizeof align);
        put(&p->ops[k].offset, sizeof(int64_t));
        put(p->ops[k].strides, sizeof(int64_t) * (size_t)p->N);
        put(&p->ops[k].dtype, sizeof(int32_t));
        put(&p->ops[k].conj, sizeof(int32_t));
    }
    put(&p->fprog_len, sizeof(int32_t));
    if (p->fprog && p->fprog_len > 0) put(p->fprog, (size_t)(2 * p->fprog_len));
    put(&p->nconsts, sizeof(int32_t));
    if (p->fconsts && p->nconsts > 0) put(p->fconsts, sizeof(double) * (size_t)(2 * p->nconsts));
    put(&p->redop, sizeof(int32_t));
    put(&p->initop, sizeof(int32_t));
    put(p->initarg, sizeof p->initarg);
    put(&p->stream, sizeof(void*));
    return s;
}

void cache_clear_locked(Cache& c, std::vector<PlanRef>& dropped) {
    for (auto& kv : c.lru) dropped.push_back(std::move(kv.second));
    c.lru.clear();
    c.map.clear();
}
}  // namespace

extern "C" {

int smr_abi_version(void) { return SMR_ABI_VERSION; }

const char* smr_last_error(void) { return g_last_error.c_str(); }

int smr_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int smr_init(int device) {
    int n = smr_device_count();
    if (n <= 0) return set_error(SMR_ENODEVICE, "smr_init: no HIP device");
    if (device < 0 || device >= n) return set_error(SMR_EINVAL, "smr_init: device index out of range");
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return hip_error(e, "hipSetDevice");
    g_device_checked = false;
    return ensure_device();
}

int smr_shutdown(void) {
    Cache& c = cache();
    std::vector<PlanRef> dropped;
    {
        std::lock_guard<std::mutex> g(c.mu);
        cache_clear_locked(c, dropped);
    }
    dropped.clear();  // frees outside the lock
    return SMR_OK;
}

int smr_malloc(size_t bytes, void** out) {
    if (!out) return set_error(SMR_EINVAL, "null out");
    int rc = ensure_device();
    if (rc) return rc;
    hipError_t e = hipMalloc(out, bytes ? bytes : 1);
    return e == hipSuccess ? SMR_OK : hip_error(e, "hipMalloc");
}
int smr_free(void* p) {
    const int rc = eager_fence_if_active();  // hipFree waits for HIP's queues only
    hipError_t e = hipFree(p);
    if (e != hipSuccess) return hip_error(e, "hipFree");
    return rc;
}
int smr_memcpy_h2d(void* dst, const void* src, size_t bytes, void* stream) {
    if (int rc = window_fence((hipStream_t)stream)) return rc;
    hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream);
    return e == hipSuccess ? SMR_OK : hip_error(e, "hipMemcpyAsync(h2d)");
}
int smr_memcpy_d2h(void* dst, const void* src, size_t bytes, void* stream) {
    if (int rc = window_fence((hipStream_t)stream)) return rc;
    hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream);
    return e == hipSuccess ? SMR_OK : hip_error(e, "hipMemcpyAsync(d2h)");
}
int smr_stream_sync(void* stream) {
    int rc = window_fence((hipStream_t)stream);
    if (rc) return rc;
    hipError_t e = hipStreamSynchronize((hipStream_t)stream);
    return e == hipSuccess ? SMR_OK : hip_error(e, "hipStreamSynchronize");
}

int smr_overlap_begin(void* stream) {
    int rc = ensure_device();
    if (rc) return rc;
    Windows& W = windows();
    std::lock_guard<std::mutex> g(W.mu);
    Window& w = W.map[stream];
    if (w.depth++ == 0 && !w.owned) {  // the first launch of a window is always ordered (after whatever the caller queued before)
        w.reads.clear();
        w.writes.clear();
        w.unordered = 0;
        w.loose = false;
    }
    return SMR_OK;
}

int smr_overlap_end(void* stream) {
    Windows& W = windows();
    std::lock_guard<std::mutex> g(W.mu);
    auto it = W.map.find(stream);
    if (it == W.map.end() || it->second.depth == 0) return set_error(SMR_EINVAL, "smr_overlap_end without smr_overlap_begin on this stream");
    Window& w = it->second;
    if (--w.depth > 0) return SMR_OK;
    const int rc = window_fence_locked(W, w, (hipStream_t)stream);
    if (!w.owned) W.map.erase(it);
    return rc;
}

int smr_overlap_fence(void* stream) { return window_fence((hipStream_t)stream); }

int smr_stream_create(void** out) {
    if (!out) return set_error(SMR_EINVAL, "null out");
    int rc = ensure_device();
    if (rc) return rc;
    hipStream_t s = nullptr;
    hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    if (e != hipSuccess) return hip_error(e, "hipStreamCreateWithFlags");
    Windows& W = windows();
    std::lock_guard<std::mutex> g(W.mu);
    W.map[(void*)s].owned = true;
    *out = (void*)s;
    return SMR_OK;
}

int smr_stream_destroy(void* stream) {
    if (!stream) return set_error(SMR_EINVAL, "null stream");
    {
        Windows& W = windows();
        std::lock_guard<std::mutex> g(W.mu);
        auto it = W.map.find(stream);
        if (it == W.map.end() || !it->second.owned) return set_error(SMR_EINVAL, "smr_stream_destroy: not a stream made by smr_stream_create");
        (void)window_fence_locked(W, it->second, (hipStream_t)stream);
        W.map.erase(it);
    }
    (void)eager_fence_if_active();
    eager_forget_stream((hipStream_t)stream);
    {   // cached one-shot plans remember the stream they were created on and drain it when they are dropped: none may outlive this one
        // (round 4: a later smr_set_option -- which clears the cache -- synchronised a destroyed stream and crashed)
        Cache& c = cache();
        std::vector<PlanRef> dropped;
        {
            std::lock_guard<std::mutex> g(c.mu);
            cache_clear_locked(c, dropped);
        }
        dropped.clear();
    }
    hipError_t e = hipStreamSynchronize((hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamDestroy((hipStream_t)stream);
    return e == hipSuccess ? SMR_OK : hip_error(e, "hipStreamDestroy");
}

int smr_plan_create(const smr_problem* problem, smr_plan** out) {
    // planning is pure host arithmetic; the device is only required to execute (or to
    // allocate reduction scratch)
    return plan_build(problem, out);
}

int smr_plan_execute(smr_plan* plan, void* const* bases, void* stream) {
    if (!plan) return set_error(SMR_EINVAL, "null plan");
    if (bases) {
        // the planner merged operands that were the same view of one buffer and co-located views that shared a
        // buffer: a rebinding must keep exactly that aliasing pattern
        for (int k = 0; k < plan->nops; ++k)
            for (int j = 0; j < k; ++j)
                if ((plan->obase[j] == plan->obase[k]) != (bases[j] == bases[k]))
                    return set_error(SMR_EINVAL, "smr_plan_execute: the rebound base pointers change which operands share a buffer; create a new plan");
    }
    int rc = ensure_device();
    if (rc) return rc;
    rc = ensure_scratch(plan);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)(stream ? stream : plan->stream);
    return execute_owned(plan, bases, s);
}

int smr_plan_prepare(smr_plan* plan) {
    if (!plan) return set_error(SMR_EINVAL, "null plan");
    int rc = ensure_device();
    if (rc) return rc;
    rc = ensure_scratch(plan);
    if (rc) return rc;
    jit_set_prepare(true);
    rc = execute(plan->plan, nullptr, (hipStream_t)plan->stream);
    jit_set_prepare(false);
    return rc;
}

int smr_plan_destroy(smr_plan* plan) {
    plan_free(plan);
    return SMR_OK;
}

int smr_plan_describe(const smr_plan* plan, char* buf, size_t buflen) {
    if (!plan || !buf || buflen == 0) return set_error(SMR_EINVAL, "null argument");
    std::snprintf(buf, buflen, "%s", plan->plan.desc.c_str());
    return SMR_OK;
}

// the f-program the kernels will run, after canonicalisation (host-only: no device, no launch)
int smr_debug_canon_prog(const smr_problem* problem, uint8_t* code, int cap, int* nwraps, int* compute_class, int32_t* orig) {
    Canon c;
    const int rc = canonicalise(problem, c);
    if (rc) return rc;
    if (nwraps) *nwraps = c.int_wraps;
    if (compute_class) *compute_class = c.bitcopy ? -1 : c.ct;
    if (orig)
        for (int j = 0; j < SMR_MAXM; ++j) orig[j] = j < c.M ? c.orig[j] : -1;
    if (code)
        for (int i = 0; i < 2 * c.prog.len && i < cap; ++i) code[i] = c.prog.code[i];
    return c.prog.len;
}

int64_t smr_plan_algorithmic_bytes(const smr_plan* plan) { return plan ? plan->plan.c.algbytes : 0; }

int smr_mapreduce_scalar(const smr_problem* problem, void* host_result) {
    if (!problem || !host_result) return set_error(SMR_EINVAL, "null argument");
    if (problem->N < 1 || problem->N > SMR_MAXN) return set_error(SMR_EINVAL, "rank out of range");
    for (int i = 0; i < problem->N; ++i)
        if (problem->dims[i] != 1 && problem->ops[0].strides[i] != 0)
            return set_error(SMR_EINVAL, "smr_mapreduce_scalar: the destination must be a single element");
    int rc = smr_mapreduce(problem);
    if (rc) return rc;
    const smr_operand& d = problem->ops[0];
    const size_t es = (size_t)dtype_size(d.dtype);
    rc = window_fence((hipStream_t)problem->stream);
    if (rc) return rc;
    hipError_t e = hipMemcpyAsync(host_result, (const char*)d.base + d.offset * (int64_t)es, es, hipMemcpyDeviceToHost,
                                  (hipStream_t)problem->stream);
    if (e != hipSuccess) return hip_error(e, "hipMemcpyAsync(scalar result)");
    e = hipStreamSynchronize((hipStream_t)problem->stream);
    return e == hipSuccess ? SMR_OK : hip_error(e, "hipStreamSynchronize");
}

int smr_plan_jit_compile(smr_plan* plan, size_t* code_size) {
    if (!plan) return set_error(SMR_EINVAL, "null plan");
    if (code_size) *code_size = 0;
    if (plan->plan.c.bitcopy) return SMR_OK;
    jit_set_dry_run(true);
    const int rc = execute(plan->plan, nullptr, nullptr);
    const size_t n = jit_dry_code_size();
    jit_set_dry_run(false);
    if (rc == SMR_OK && code_size) *code_size = n;
    return rc;
}

int smr_plan_jit_source(const smr_plan* plan, char* buf, size_t buflen) {
    if (!plan || !buf || buflen == 0) return set_error(SMR_EINVAL, "null argument");
    static const char* names[] = {"float", "double", "smr::c32", "smr::c64"};
    const Canon& c = plan->plan.c;
    std::snprintf(buf, buflen, "%s", jit_functor_source(c, c.ct == SMR_I64 ? "smr::ix64" : names[c.ct & 3]).c_str());
    return SMR_OK;
}

int64_t smr_plan_tile_order(const smr_plan* plan, uint32_t* out, size_t cap) {
    if (!plan || (plan->plan.family != FAM_TILED && plan->plan.family != FAM_ORBIT)) return 0;
    const std::vector<uint32_t>& ord = plan->plan.family == FAM_ORBIT ? plan->plan.orbit.wtile : plan->plan.tile.ord;
    if (out)
        for (size_t i = 0; i < ord.size() && i < cap; ++i) out[i] = ord[i];
    return (int64_t)ord.size();
}

int64_t smr_plan_orbit_pairs(const smr_plan* plan, uint32_t* out, size_t cap) {
    if (!plan || plan->plan.family != FAM_ORBIT || !plan->plan.orbit.pair_ok) return 0;
    const std::vector<uint32_t>& pt = plan->plan.orbit.ptile;
    if (out)
        for (size_t i = 0; i < pt.size() && i < cap; ++i) out[i] = pt[i];
    return (int64_t)pt.size();
}

int64_t smr_plan_flat_runs(const smr_plan* plan, int64_t* out, size_t cap) {
    if (!plan || plan->plan.family != FAM_FLAT || !plan->plan.flat2.on || plan->plan.flatb.on) return 0;
    const Canon& c = plan->plan.c;
    const Flat2Plan& f = plan->plan.flat2;
    std::vector<int64_t> v = {f.kt, f.shared ? 1 : 0, c.N, f.R[0], f.R[1], f.TP[0], f.TP[1], f.p[0], f.p[1]};
    for (int d = 0; d < c.N; ++d) v.push_back(c.dims[d]);
    for (int d = 0; d < c.N; ++d) v.push_back(c.strides[0][d]);
    for (int d = 0; d < c.N; ++d) v.push_back(c.strides[f.kt][d]);
    for (int s = 0; s < 2; ++s)
        for (int d = 0; d < c.N; ++d) v.push_back(f.ingroup[s][d] ? 1 : 0);
    for (int s = 0; s < 2; ++s)
        for (int r = 0; r < f.R[s]; ++r) v.push_back(f.roff[s][r]);
    if (out)
        for (size_t i = 0; i < v.size() && i < cap; ++i) out[i] = v[i];
    return (int64_t)v.size();
}

int64_t smr_plan_flat_side(const smr_plan* plan, int64_t* out, size_t cap) {
    if (!plan || plan->plan.family != FAM_FLAT || plan->plan.flat2.on || plan->plan.flatb.on) return 0;
    const Canon& c = plan->plan.c;
    const FlatPlan& f = plan->plan.flat;
    std::vector<int64_t> v = {f.dir, f.R, f.tplog, f.tqlog, f.p, f.q, f.lshare ? 1 : 0, f.fuse ? 1 : 0, f.kt, c.N};
    for (int d = 0; d < c.N; ++d) v.push_back(c.dims[d]);
    for (int d = 0; d < c.N; ++d) v.push_back(c.strides[0][d]);
    for (int d = 0; d < c.N; ++d) v.push_back(c.strides[f.kt][d]);
    for (int d = 0; d < c.N; ++d) v.push_back(f.ingroup[d] ? 1 : 0);
    for (int r = 0; r < f.R; ++r) v.push_back(f.roff[r]);
    if (out)
        for (size_t i = 0; i < v.size() && i < cap; ++i) out[i] = v[i];
    return (int64_t)v.size();
}

int64_t smr_plan_flat_batched(const smr_plan* plan, int64_t* out, size_t cap) {
    if (!plan || plan->plan.family != FAM_FLAT || !plan->plan.flatb.on) return 0;
    const Canon& c = plan->plan.c;
    const FlatBPlan& f = plan->plan.flatb;
    std::vector<int64_t> v = {f.g, f.P, f.K, c.N};
    for (int d = 0; d < c.N; ++d) v.push_back(c.dims[d]);
    for (int d = 0; d < c.N; ++d) v.push_back(c.strides[0][d]);
    for (int d = 0; d < c.N; ++d) v.push_back(c.strides[1][d]);
    for (int r = 0; r < f.P; ++r) v.push_back(f.srcoff[r]);
    if (out)
        for (size_t i = 0; i < v.size() && i < cap; ++i) out[i] = v[i];
    return (int64_t)v.size();
}

int smr_mapreduce(const smr_problem* problem) {
    if (!problem) return set_error(SMR_EINVAL, "null problem");
    if (problem->N < 1 || problem->N > SMR_MAXN || problem->M < 1 || problem->M > SMR_MAXM)
        return set_error(SMR_EINVAL, "rank / operand count out of range");
    int rc = ensure_device();
    if (rc) return rc;
    Cache& c = cache();
    std::string key = signature(problem);
    PlanRef h, evicted;
    {
        std::lock_guard<std::mutex> g(c.mu);
        auto it = c.map.find(key);
        if (it != c.map.end()) {
            c.lru.splice(c.lru.begin(), c.lru, it->second);
            h = it->second->second;
        }
    }
    if (!h) {
        smr_plan* raw = nullptr;
        rc = plan_build(problem, &raw);
        if (rc) return rc;
        PlanRef fresh(raw, PlanDeleter());
        std::lock_guard<std::mutex> g(c.mu);
        auto it = c.map.find(key);  // another thread may have inserted the same key meanwhile: use its plan
        if (it != c.map.end()) {
            c.lru.splice(c.lru.begin(), c.lru, it->second);
            h = it->second->second;
        } else {
            c.lru.emplace_front(key, fresh);
            c.map[key] = c.lru.begin();
            h = fresh;
            if (c.lru.size() > Cache::CAP) {
                evicted = std::move(c.lru.back().second);  // released after the lock is gone
                c.map.erase(c.lru.back().first);
                c.lru.pop_back();
            }
        }
    }
    evicted.reset();
    rc = ensure_scratch(h.get());
    if (rc) return rc;
    // the plan was built for another set of base pointers with the same aliasing / alignment: bind the
    // caller's at launch
    void* bases[SMR_MAXM];
    for (int k = 0; k < problem->M; ++k) bases[k] = problem->ops[k].base;
    return execute_owned(h.get(), bases, (hipStream_t)problem->stream);
}

int smr_shard(const smr_problem* p, int nshards, int shard, smr_problem* out, int* needs_allreduce) {
    return smr_shard_ex(p, nshards, shard, 0u, out, needs_allreduce, nullptr, nullptr, nullptr);
}

int smr_shard_ex(const smr_problem* p, int nshards, int shard, uint32_t local_ops, smr_problem* out, int* needs_allreduce, int* split_dim,
                 int64_t* start_out, int64_t* stop_out) {
    if (!p || !out) return set_error(SMR_EINVAL, "null argument");
    if (nshards < 1 || shard < 0 || shard >= nshards) return set_error(SMR_EINVAL, "bad shard index");
    if (p->N < 1 || p->N > SMR_MAXN || p->M < 1 || p->M > SMR_MAXM) return set_error(SMR_EINVAL, "bad N/M");
    *out = *p;
    if (needs_allreduce) *needs_allreduce = 0;
    if (split_dim) *split_dim = -1;
    if (start_out) *start_out = 0;
    if (stop_out) *stop_out = 0;
    if (nshards == 1) return SMR_OK;
    // Split dim: the slowest-varying destination dim that is long enough (destination slabs
    // are then disjoint and contiguous-ish); for reductions prefer a kept dim so that no
    // exchange is needed -- the same race-freedom rule as src/mapreduce.jl:172-177.  Only when
    // no kept dim is long enough split a reduced dim and combine partials afterwards, like
    // the reference's complete-reduction branch (:153-170).
    int best = -1;
    int64_t beststride = -1;
    for (int i = 0; i < p->N; ++i) {
        int64_t s = p->ops[0].strides[i] < 0 ? -p->ops[0].strides[i] : p->ops[0].strides[i];
        if (s != 0 && p->dims[i] >= nshards && s > beststride) {
            best = i;
            beststride = s;
        }
    }
    int allred = 0;
    if (best < 0) {
        if (p->redop == SMR_RED_NONE) {
            // tiny map: give everything to shard 0, empty work elsewhere is not expressible
            // (dims >= 1), so fall back to the longest dim
            for (int i = 0; i < p->N; ++i)
                if (best < 0 || p->dims[i] > p->dims[best]) best = i;
            if (p->dims[best] < nshards) return set_error(SMR_EUNSUPPORTED, "box too small to shard");
        } else {
            int64_t bs = -1;
            for (int i = 0; i < p->N; ++i) {
                if (p->ops[0].strides[i] != 0 || p->dims[i] < nshards) continue;
                int64_t s = 0;
                for (int k = 1; k < p->M; ++k) {
                    int64_t v = p->ops[k].strides[i] < 0 ? -p->ops[k].strides[i] : p->ops[k].strides[i];
                    s = std::max(s, v);
                }
                if (s > bs) {
                    bs = s;
                    best = i;
                }
            }
            if (best < 0) return set_error(SMR_EUNSUPPORTED, "box too small to shard");
            allred = 1;
        }
    }
    const int64_t d = p->dims[best];
    const int64_t start = d * shard / nshards, end = d * (shard + 1) / nshards;
    out->dims[best] = end - start;
    // an operand flagged in local_ops is already this shard's slab: its base/offset address box index
    // `start` of the split dim (block-partitioned inputs: nothing is replicated); the others live in the whole
    // parent and get the reference's offset shift (src/mapreduce.jl:217-219)
    for (int k = 0; k < p->M; ++k)
        if (!((local_ops >> k) & 1u)) out->ops[k].offset = p->ops[k].offset + start * p->ops[k].strides[best];
    if (split_dim) *split_dim = best;
    if (start_out) *start_out = start;
    if (stop_out) *stop_out = end;
    if (allred && shard != 0) out->initop = SMR_INIT_NONE;  // initop is applied once, on shard 0
    if (needs_allreduce) *needs_allreduce = allred;
    return SMR_OK;
}

int smr_set_option(const char* name, int64_t value) {
    if (!name) return set_error(SMR_EINVAL, "null option name");
    Options& o = options();
    std::string n(name);
    bool ok = true;
    if (n == "force_family") o.force_family = value;
    else if (n == "tile_log2") o.tile_log2 = value;
    else if (n == "tile_order") o.tile_order = value;
    else if (n == "tile_block") o.tile_block = value;
    else if (n == "tile_block_xcd") o.tile_block_xcd = value;
    else if (n == "tile_block_min_axes") o.tile_block_min_axes = value;
    else if (n == "reduce_blocks") o.reduce_blocks = value;
    else if (n == "jit") o.jit = value;
    else if (n == "tiled_persist") o.tiled_persist = value;
    else if (n == "stream_u") o.stream_u = value;
    else if (n == "stream_pack_rows") o.stream_pack_rows = value;
    else if (n == "flatb") o.flatb = value;
    else if (n == "flat2_long") o.flat2_long = value;
    else if (n == "flat2_pair") o.flat2_pair = value;
    else if (n == "eager_direct") o.eager_direct = value;
    else if (n == "reduce_tree") o.reduce_tree = value;
    else if (n == "tiled_persist_wpc") o.tiled_persist_wpc = value;
    else if (n == "tiled_persist_min") o.tiled_persist_min = value;
    else if (n == "reduce_part_kind") o.reduce_part_kind = value;
    else if (n == "reduce_col_txlog") o.reduce_col_txlog = value;
    else if (n == "reduce_part_wgs") o.reduce_part_wgs = value;
    else if (n == "reduce_single") o.reduce_single = value;
    else if (n == "reduce_col_narrow") o.reduce_col_narrow = value;
    else if (n == "reduce_col_exact") o.reduce_col_exact = value;
    else if (n == "reduce_row_floor") o.reduce_row_floor = value;
    else if (n == "reduce_row_dense") o.reduce_row_dense = value;
    else if (n == "flat2") o.flat2 = value;
    else if (n == "flat_wide") o.flat_wide = value;
    else if (n == "flat2_bytes") o.flat2_bytes = value;
    else if (n == "flat2_lead_bytes") o.flat2_lead_bytes = value;
    else if (n == "tiled_vec") o.tiled_vec = value;
    else if (n == "tiled_uavec") o.tiled_uavec = value;
    else if (n == "tiled_force_edge") o.tiled_force_edge = value;
    else if (n == "tiled_edge_first") o.tiled_edge_first = value;
    else if (n == "nt_stream_min") o.nt_stream_min = value;
    else if (n == "nt_store") o.nt_store = value;
    else if (n == "seq_self_release") o.seq_self_release = value;
    else if (n == "eager_self_release") o.eager_self_release = value;
    else if (n == "overlap_window_hip") o.overlap_window_hip = value;
    else if (n == "tiled_gorder") o.tiled_gorder = value;
    else if (n == "tiled_xpose") o.tiled_xpose = value;
    else if (n == "stream_ua") o.stream_ua = value;
    else if (n == "allreduce_f64") o.allreduce_f64 = value;
    else if (n == "self_release_max_bytes") o.self_release_max_bytes = value;
    else if (n == "self_release_max_total") o.self_release_max_total = value;
    else if (n == "nt_load") o.nt_load = value;
    else if (n == "orbit_min") o.orbit_min = value;
    else if (n == "orbit_few") o.orbit_few = value;
    else if (n == "orbit_pack") o.orbit_pack = value;
    else if (n == "orbit_pair") o.orbit_pair = value;
    else if (n == "orbit_pipe") o.orbit_pipe = value;
    else if (n == "orbit_lds_min") o.orbit_lds_min = value;
    else if (n == "orbit_group") o.orbit_group = value;
    else if (n == "orbit_wgs") o.orbit_wgs = value;
    else if (n == "orbit_minrun") o.orbit_minrun = value;
    else if (n == "orbit_skew") o.orbit_skew = value;
    else if (n == "orbit_deal") o.orbit_deal = value;
    else if (n == "flat") o.flat = value;
    else if (n == "stamp_base" || n == "stamp_cap" || n == "stamp_used") {  // no plan depends on these: keep the cache
        (n == "stamp_base" ? o.stamp_base : (n == "stamp_cap" ? o.stamp_cap : o.stamp_used)) = value;
        return SMR_OK;
    }
    else if (n == "orbit_lg") o.orbit_lg = value;
    else if (n == "orbit") o.orbit = value;
    else if (n == "max_lds_bytes") o.max_lds_bytes = value;
    else if (n.rfind("tile_lg", 0) == 0 && n.size() == 8 && n[7] >= '0' && n[7] <= '7') o.tile_lg[n[7] - '0'] = value;
    else ok = false;
    if (!ok) return set_error(SMR_EINVAL, "unknown option " + n);
    Cache& c = cache();
    std::vector<PlanRef> dropped;
    {
        std::lock_guard<std::mutex> g(c.mu);
        cache_clear_locked(c, dropped);
    }
    dropped.clear();
    return SMR_OK;
}

int64_t smr_get_option(const char* name) {
    if (!name) return -1;
    const Options& o = options();
    std::string n(name);
    if (n == "force_family") return o.force_family;
    if (n == "tile_log2") return o.tile_log2;
    if (n == "tile_order") return o.tile_order;
    if (n == "tile_block") return o.tile_block;
    if (n == "tile_block_xcd") return o.tile_block_xcd;
    if (n == "tile_block_min_axes") return o.tile_block_min_axes;
    if (n == "reduce_blocks") return o.reduce_blocks;
    if (n == "jit") return o.jit;
    if (n == "tiled_persist") return o.tiled_persist;
    if (n == "stream_u") return o.stream_u;
    if (n == "stream_pack_rows") return o.stream_pack_rows;
    if (n == "flatb") return o.flatb;
    if (n == "flat2_long") return o.flat2_long;
    if (n == "flat2_pair") return o.flat2_pair;
    if (n == "eager_direct") return o.eager_direct;
    if (n == "eager_launches") return eager_stat(0);
    if (n == "eager_free") return eager_stat(1);
    if (n == "eager_same") return eager_stat(2);
    if (n == "eager_cross") return eager_stat(3);
    if (n == "eager_fallback") return eager_stat(4);
    if (n == "eager_kernarg_device") return eager_stat(5);
    if (n == "eager_arg_hits") return eager_stat(7);
    if (n == "eager_gpu_only_signals") return eager_stat(6);
    if (n == "reduce_tree") return o.reduce_tree;
    if (n == "tiled_persist_wpc") return o.tiled_persist_wpc;
    if (n == "tiled_persist_min") return o.tiled_persist_min;
    if (n == "reduce_part_kind") return o.reduce_part_kind;
    if (n == "reduce_col_txlog") return o.reduce_col_txlog;
    if (n == "reduce_part_wgs") return o.reduce_part_wgs;
    if (n == "reduce_single") return o.reduce_single;
    if (n == "reduce_col_narrow") return o.reduce_col_narrow;
    if (n == "reduce_col_exact") return o.reduce_col_exact;
    if (n == "reduce_row_floor") return o.reduce_row_floor;
    if (n == "reduce_row_dense") return o.reduce_row_dense;
    if (n == "flat2") return o.flat2;
    if (n == "flat_wide") return o.flat_wide;
    if (n == "flat2_bytes") return o.flat2_bytes;
    if (n == "flat2_lead_bytes") return o.flat2_lead_bytes;
    if (n == "overlap_any" || n == "overlap_ordered" || n == "overlap_fences") {
        Windows& W = windows();
        std::lock_guard<std::mutex> g(W.mu);
        return n == "overlap_any" ? W.stat_any : (n == "overlap_ordered" ? W.stat_ordered : W.stat_fences);
    }
    if (n == "jit_compiles") return jit_stats().compiles;
    if (n == "jit_hits") return jit_stats().hits;
    if (n == "jit_failures") return jit_stats().failures;
    if (n == "jit_compile_ms") return (int64_t)jit_stats().compile_ms;
    if (n == "tiled_vec") return o.tiled_vec;
    if (n == "tiled_uavec") return o.tiled_uavec;
    if (n == "tiled_force_edge") return o.tiled_force_edge;
    if (n == "tiled_edge_first") return o.tiled_edge_first;
    if (n == "nt_stream_min") return o.nt_stream_min;
    if (n == "nt_store") return o.nt_store;
    if (n == "seq_self_release") return o.seq_self_release;
    if (n == "eager_self_release") return o.eager_self_release;
    if (n == "overlap_window_hip") return o.overlap_window_hip;
    if (n == "tiled_gorder") return o.tiled_gorder;
    if (n == "tiled_xpose") return o.tiled_xpose;
    if (n == "stream_ua") return o.stream_ua;
    if (n == "allreduce_f64") return o.allreduce_f64;
    if (n == "launches") return g_launches.load();
    if (n == "allreduces") return comm_stat(0);
    if (n == "allreduces_inplace") return comm_stat(1);
    if (n == "self_release_max_bytes") return o.self_release_max_bytes;
    if (n == "self_release_max_total") return o.self_release_max_total;
    if (n == "nt_load") return o.nt_load;
    if (n == "orbit_min") return o.orbit_min;
    if (n == "orbit_few") return o.orbit_few;
    if (n == "orbit_pack") return o.orbit_pack;
    if (n == "orbit_pair") return o.orbit_pair;
    if (n == "orbit_pipe") return o.orbit_pipe;
    if (n == "orbit_lds_min") return o.orbit_lds_min;
    if (n == "orbit_group") return o.orbit_group;
    if (n == "orbit_wgs") return o.orbit_wgs;
    if (n == "orbit_minrun") return o.orbit_minrun;
    if (n == "orbit_skew") return o.orbit_skew;
    if (n == "orbit_deal") return o.orbit_deal;
    if (n == "flat") return o.flat;
    if (n == "stamp_base") return o.stamp_base;
    if (n == "stamp_cap") return o.stamp_cap;
    if (n == "stamp_used") return o.stamp_used;
    if (n == "stamp_build") return SMR_STAMP;
    if (n == "orbit_lg") return o.orbit_lg;
    if (n == "orbit") return o.orbit;
    if (n == "max_lds_bytes") return o.max_lds_bytes;
    if (n.rfind("tile_lg", 0) == 0 && n.size() == 8 && n[7] >= '0' && n[7] <= '7') return o.tile_lg[n[7] - '0'];
    return -1;
}

}  // extern "C"
