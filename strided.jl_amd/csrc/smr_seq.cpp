// smr_seq.cpp -- recorded sequences of plan executions, replayed as hand-built AQL packets on the library's own HSA queue.
//
// Why.  The reference runs independent pieces of work as concurrent tasks and waits only where it must (src/mapreduce.jl:203-223).
// On MI355X the unit is the LAUNCH, and a stream-ordered launch of 16 MiB is a span of 1.6-3.2 us plus a 1.6-1.9 us boundary (drain,
// write-back, next dispatch) even when its successor touches none of its data.  What round 4 measured (profiles/r04_overlap.txt):
//   * HIP accepts hipExtAnyOrderLaunch but ignores it on gfx9 (hip_ext.h says so; the device stamps confirm it): through HIP every
//     packet of a stream carries the AQL barrier bit;
//   * two HIP streams do overlap the two kernels of the bench step (8.6 -> 6.2 us per step) -- when two host threads feed them: HIP's
//     eager launch path costs 3.6-4 us of host time per launch, one thread cannot feed two queues;
//   * a hipGraph with two branches is executed node by node through that same eager path (8.2-8.8 us per step), only single-chain
//     graphs get the batched submission.
// So the library submits its own packets: a sequence is recorded ONCE (every launch's kernel object, grid and kernarg block, resident
// in device memory), the dependency analysis of the overlap window (smr_api.cpp: byte ranges read / written) decides per launch whether
// its packet carries the barrier bit, and a replay is N x 64-byte stores into the queue rings plus a doorbell per queue -- ~0.1 us of
// host time per launch, no host thread in the loop.  Independent launches go to DIFFERENT hardware queues (one per dependency
// component): inside one queue the packet processor runs consecutive dispatches one after the other on their agent-scope fences even
// with the barrier bit clear (measured with device stamps, profiles/r04_overlap.txt); two queues do overlap.
//
// Kernel objects come from the code objects HIP itself has loaded: host stub -> kernel name (hipKernelNameRefByPtr) -> "<name>.kd" looked
// up in the process's HSA executables (loader extension 1.03: hsa_ven_amd_loader_iterate_executables).  Nothing is loaded twice.
// Ordering against the caller's HIP stream: smr_seq_run waits (on the host) for `stream` to drain when it is busy, and makes it wait for
// the replay with hipStreamWaitValue64 on the completion signal's value -- "as if the sequence had been launched in order on stream".
#include <dlfcn.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <hsa/hsa_ven_amd_loader.h>
#include <link.h>
#include <sched.h>
#include <sys/mman.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#include "smr_internal.h"
#include "smr_kmeta.h"

namespace smr {

static thread_local std::vector<RecLaunch>* tl_recorder = nullptr;
std::vector<RecLaunch>* recorder() { return tl_recorder; }
static thread_local bool tl_rec_seq = false, tl_self_released = false;
static thread_local bool tl_rec_eager = false;  // the recorder belongs to an eager call: `allow` was decided by the eager option, not the sequence's
void set_recorder(std::vector<RecLaunch>* r, bool allow_self_release, bool eager) {
    tl_recorder = r;
    tl_rec_seq = r != nullptr && allow_self_release;
    tl_rec_eager = r != nullptr && eager;
}
void mark_self_released() {
    if (tl_recorder) tl_self_released = true;
}
bool want_self_release(const Plan& plan) {
    if (!tl_recorder || !tl_rec_seq) return false;
    const Options& o = options();
    // the two knobs are independent: a sequence asks seq_self_release, an eager call was admitted by eager_self_release
    if (!(tl_rec_eager ? o.eager_self_release : o.seq_self_release)) return false;
    const Canon& c = plan.c;
    i64 lo = 0, hi = 0;  // extent of the destination in elements
    for (int d = 0; d < c.N; ++d) {
        const i64 ext = (c.dims[d] - 1) * c.strides[0][d];
        (ext < 0 ? lo : hi) += ext;
    }
    return (hi - lo + 1) * (i64)c.esize[0] <= o.self_release_max_bytes;
}
// Eager path: is the recent write set of this process small enough to stay in the Infinity Cache?  (Write-through stores pay off
// while the destinations are cache-resident and lose on partial lines that go to HBM: profiles/r05_bench_n1.json, cold 4-way sum.)
// A 16-slot direct-mapped table of recently written destinations (base address -> bytes); O(1) per call.
bool eager_recent_writes_fit(uintptr_t dest_lo, uintptr_t dest_hi) {
    static std::mutex mu;
    static uintptr_t key[16] = {};
    static size_t bytes[16] = {}, total = 0;
    std::lock_guard<std::mutex> g(mu);
    const unsigned slot = (unsigned)((dest_lo >> 12) * 0x9E3779B1u >> 28) & 15u;
    if (key[slot] != dest_lo || bytes[slot] != dest_hi - dest_lo) {
        total -= bytes[slot];
        key[slot] = dest_lo;
        bytes[slot] = dest_hi - dest_lo;
        total += bytes[slot];
    }
    return (i64)total <= options().self_release_max_total;
}
static thread_local int tl_slice_kind = 0;
static thread_local unsigned tl_slice_off = 0, tl_slice_row = 0;
void mark_sliceable(int kind, unsigned off, unsigned row) {
    if (!tl_recorder) return;
    tl_slice_kind = kind;
    tl_slice_off = off;
    tl_slice_row = row;
}
void take_slice_mark(RecLaunch& r) {
    r.slice_kind = tl_slice_kind;
    r.slice_off = tl_slice_off;
    r.slice_row = tl_slice_row;
    tl_slice_kind = 0;
    r.self_released = tl_self_released;
    tl_self_released = false;
}

// smr_api.cpp
int seq_execute_plan(smr_plan* plan, void* const* bases, hipStream_t s, bool prepare_only);
void seq_footprint(smr_plan* plan, void* const* bases, std::vector<std::pair<uintptr_t, uintptr_t>>& rd, std::vector<std::pair<uintptr_t, uintptr_t>>& wr);
int seq_nops(smr_plan* plan);
bool seq_stream_is_owned(hipStream_t s);

namespace {

// ---- the HSA runtime HIP already loaded (never a second copy) ------------------------------------------------------------------
struct Hsa {
    void* lib = nullptr;
    decltype(&hsa_init) init = nullptr;
    decltype(&hsa_iterate_agents) iterate_agents = nullptr;
    decltype(&hsa_agent_get_info) agent_get_info = nullptr;
    decltype(&hsa_queue_create) queue_create = nullptr;
    decltype(&hsa_queue_destroy) queue_destroy = nullptr;
    decltype(&hsa_queue_load_read_index_scacquire) load_read_index = nullptr;
    decltype(&hsa_queue_add_write_index_relaxed) add_write_index = nullptr;
    decltype(&hsa_signal_create) signal_create = nullptr;
    decltype(&hsa_signal_destroy) signal_destroy = nullptr;
    decltype(&hsa_signal_store_relaxed) signal_store_relaxed = nullptr;
    decltype(&hsa_signal_store_screlease) signal_store_screlease = nullptr;
    decltype(&hsa_signal_load_scacquire) signal_load = nullptr;
    decltype(&hsa_signal_wait_scacquire) signal_wait = nullptr;
    decltype(&hsa_amd_signal_value_pointer) signal_value_pointer = nullptr;
    decltype(&hsa_system_get_major_extension_table) get_ext_table = nullptr;
    decltype(&hsa_executable_get_symbol_by_name) get_symbol_by_name = nullptr;
    decltype(&hsa_executable_symbol_get_info) symbol_get_info = nullptr;
    decltype(&hsa_status_string) status_string = nullptr;
    // optional (eager dispatch): signals without an interrupt, argument blocks in device memory written through the BAR
    decltype(&hsa_amd_signal_create) amd_signal_create = nullptr;
    decltype(&hsa_amd_agent_iterate_memory_pools) iterate_pools = nullptr;
    decltype(&hsa_amd_memory_pool_get_info) pool_get_info = nullptr;
    decltype(&hsa_amd_agent_memory_pool_get_info) agent_pool_get_info = nullptr;
    decltype(&hsa_amd_memory_pool_allocate) pool_allocate = nullptr;
    decltype(&hsa_amd_agents_allow_access) allow_access = nullptr;
    hsa_ven_amd_loader_1_03_pfn_t loader;
    bool ok = false;
    std::string why;
};

int find_hsa(struct dl_phdr_info* info, size_t, void* data) {
    if (info->dlpi_name && std::strstr(info->dlpi_name, "libhsa-runtime64")) {
        *(std::string*)data = info->dlpi_name;
        return 1;
    }
    return 0;
}

Hsa& hsa() {
    static Hsa* h = [] {
        Hsa* x = new Hsa();
        // A profiler (rocprofv3, HSA_TOOLS_LIB) intercepts queue creation and hands out proxy queues whose rings are not meant to be
        // written by the application directly (measured: rocprofv3 --kernel-trace crashes inside the first ring write).  Under such a
        // tool -- or with SMR_SEQ_DIRECT=0 -- sequences replay through HIP, in recorded order, and the profiler sees ordinary launches.
        {
            const char* force = std::getenv("SMR_SEQ_DIRECT");
            bool tool = false;
            for (const char* v : {"HSA_TOOLS_LIB", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_FORCE_LOAD", "ROCPROF_OUTPUT_PATH"})
                if (const char* e = std::getenv(v)) tool = tool || *e;
            if (const char* e = std::getenv("LD_PRELOAD")) tool = tool || std::strstr(e, "rocprof") != nullptr;
            if ((force && force[0] == '0') || (tool && !(force && force[0] == '1'))) {
                x->why = force && force[0] == '0' ? "SMR_SEQ_DIRECT=0" : "an HSA tools library (profiler) intercepts the queues";
                return x;
            }
        }
        std::string path;
        dl_iterate_phdr(find_hsa, &path);
        if (path.empty()) {
            x->why = "libhsa-runtime64 is not loaded in this process (HIP not initialised?)";
            return x;
        }
        x->lib = dlopen(path.c_str(), RTLD_NOW | RTLD_NOLOAD);
        if (!x->lib) {
            x->why = std::string("dlopen(RTLD_NOLOAD) of ") + path + " failed";
            return x;
        }
        bool all = true;
#define SMR_HSA_SYM(field, name)                                   \
    x->field = (decltype(x->field))dlsym(x->lib, #name);            \
    if (!x->field) {                                                \
        all = false;                                                \
        x->why += std::string(" missing ") + #name;                 \
    }
        SMR_HSA_SYM(init, hsa_init)
        SMR_HSA_SYM(iterate_agents, hsa_iterate_agents)
        SMR_HSA_SYM(agent_get_info, hsa_agent_get_info)
        SMR_HSA_SYM(queue_create, hsa_queue_create)
        SMR_HSA_SYM(queue_destroy, hsa_queue_destroy)
        SMR_HSA_SYM(load_read_index, hsa_queue_load_read_index_scacquire)
        SMR_HSA_SYM(add_write_index, hsa_queue_add_write_index_relaxed)
        SMR_HSA_SYM(signal_create, hsa_signal_create)
        SMR_HSA_SYM(signal_destroy, hsa_signal_destroy)
        SMR_HSA_SYM(signal_store_relaxed, hsa_signal_store_relaxed)
        SMR_HSA_SYM(signal_store_screlease, hsa_signal_store_screlease)
        SMR_HSA_SYM(signal_load, hsa_signal_load_scacquire)
        SMR_HSA_SYM(signal_wait, hsa_signal_wait_scacquire)
        SMR_HSA_SYM(signal_value_pointer, hsa_amd_signal_value_pointer)
        SMR_HSA_SYM(get_ext_table, hsa_system_get_major_extension_table)
        SMR_HSA_SYM(get_symbol_by_name, hsa_executable_get_symbol_by_name)
        SMR_HSA_SYM(symbol_get_info, hsa_executable_symbol_get_info)
        SMR_HSA_SYM(status_string, hsa_status_string)
#undef SMR_HSA_SYM
        x->amd_signal_create = (decltype(x->amd_signal_create))dlsym(x->lib, "hsa_amd_signal_create");
        x->iterate_pools = (decltype(x->iterate_pools))dlsym(x->lib, "hsa_amd_agent_iterate_memory_pools");
        x->pool_get_info = (decltype(x->pool_get_info))dlsym(x->lib, "hsa_amd_memory_pool_get_info");
        x->agent_pool_get_info = (decltype(x->agent_pool_get_info))dlsym(x->lib, "hsa_amd_agent_memory_pool_get_info");
        x->pool_allocate = (decltype(x->pool_allocate))dlsym(x->lib, "hsa_amd_memory_pool_allocate");
        x->allow_access = (decltype(x->allow_access))dlsym(x->lib, "hsa_amd_agents_allow_access");
        if (!all) return x;
        if (x->init() != HSA_STATUS_SUCCESS) {  // reference-counted: HIP holds the first reference
            x->why = "hsa_init failed";
            return x;
        }
        std::memset(&x->loader, 0, sizeof x->loader);
        if (x->get_ext_table(HSA_EXTENSION_AMD_LOADER, 1, sizeof x->loader, &x->loader) != HSA_STATUS_SUCCESS ||
            !x->loader.hsa_ven_amd_loader_iterate_executables) {
            x->why = "HSA loader extension 1.03 (iterate_executables) is unavailable";
            return x;
        }
        x->ok = true;
        return x;
    }();
    return *h;
}

// ---- per-device direct queue --------------------------------------------------------------------------------------------------
constexpr int SEQ_MAXQ = 8;
struct KernelRef {
    uint64_t object = 0;
    uint32_t kernarg_size = 0, group_static = 0, private_size = 0;
    std::string name;
    KernargLayout layout;        // where the hidden arguments live: from the code object's metadata, else the code-object-v5 rule
    bool from_metadata = false;
};
struct Direct {
    int dev = 0;
    hsa_agent_t agent{};
    // A queue error (the HSA callback) or a wait that ran out of time marks the device's direct path as failed: every wait returns,
    // the call that noticed returns SMR_EHIP, and later executions go through HIP (direct().ok is false from then on).
    std::atomic<bool> failed{false};
    std::string fail_why;
    std::mutex why_mu;                  // guards `why` / `fail_why` (written by whichever thread sees the failure, read by smr_seq_info and the planners of other threads: ADVICE r5)
    // kernarg layouts by code object (storage base address of the loaded image) and kernel-descriptor symbol
    std::map<uint64_t, std::map<std::string, KernargLayout>> code_objects;
    int n_meta = 0, n_v5 = 0;    // kernels resolved with a layout from metadata / from the v5 rule
    bool agent_by_pci = false;   // the HSA agent was found by the HIP device's PCI address (false: the only GPU agent there is)
    bool probe_ok = false;       // the self-test launch saw the blockDim / gridDim / LDS it was given
    bool probe_meta = false;     // ... with a layout read from metadata
    // SEQ_MAXQ hardware queues of the library's own (created on first use): launches that belong to different dependency
    // components of a sequence go to different queues, where neither the barrier bit nor the acquire / release fences of one
    // chain hold up the other (inside ONE queue the packet processor serialises consecutive dispatches on their fences even
    // when the barrier bit is clear: profiles/r04_overlap.txt)
    hsa_queue_t* q[SEQ_MAXQ] = {};
    hsa_signal_t done[SEQ_MAXQ] = {};   // completion signal of the last packet a replay put on queue k
    volatile int64_t* done_ptr[SEQ_MAXQ] = {};
    bool armed[SEQ_MAXQ] = {};          // done[k] belongs to a replay nobody waited for yet
    double t_submit = 0, last_us = 0;   // host clock at the first doorbell of the replay in flight; doorbell -> completion observed
    std::mutex mu;                      // one replay is written at a time
    std::map<const void*, KernelRef> kernels;
    bool wait_value_ok = false;
    bool ok = false;
    std::string why;
};
struct AgentPick {
    Hsa* h;
    uint32_t bdf, domain;
    hsa_agent_t found{};
    bool have = false;
    hsa_agent_t first{};
    int ngpu = 0;
};
hsa_status_t pick_agent(hsa_agent_t a, void* data) {
    AgentPick* p = (AgentPick*)data;
    hsa_device_type_t t;
    if (p->h->agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t) != HSA_STATUS_SUCCESS || t != HSA_DEVICE_TYPE_GPU) return HSA_STATUS_SUCCESS;
    if (p->ngpu++ == 0) p->first = a;
    uint32_t bdf = 0, dom = 0;
    p->h->agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_BDFID, &bdf);
    p->h->agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_DOMAIN, &dom);
    if (bdf == p->bdf && dom == p->domain) {
        p->found = a;
        p->have = true;
    }
    return HSA_STATUS_SUCCESS;
}

void queue_error(hsa_status_t st, hsa_queue_t*, void* data) {
    std::fprintf(stderr, "libstrided_hip: the direct-dispatch HSA queue reported error 0x%x; direct dispatch is switched off\n", (unsigned)st);
    if (Direct* d = (Direct*)data) d->failed.store(true);
}

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// how long a wait on the direct queues may last before the path is declared dead ($SMR_DIRECT_TIMEOUT_MS, default 30 s: far above
// any replay this library issues, far below "forever")
double direct_timeout_s() {
    static const double t = [] {
        const char* e = std::getenv("SMR_DIRECT_TIMEOUT_MS");
        const double ms = e ? std::atof(e) : 0;
        return ms > 0 ? ms * 1e-3 : 30.0;
    }();
    return t;
}

void direct_fail(Direct& d, const std::string& why) {
    std::string msg;
    {
        std::lock_guard<std::mutex> g(d.why_mu);
        const bool first = !d.failed.exchange(true) || d.fail_why.empty();
        if (first) d.fail_why = why;
        d.ok = false;
        d.why = "direct dispatch failed earlier: " + d.fail_why;
        msg = d.fail_why;
    }
    // a holding kernel on some HIP stream may be polling the replay's completion signals (asynchronous smr_seq_run): let it go
    for (int k = 0; k < SEQ_MAXQ; ++k)
        if (d.q[k] && d.done_ptr[k]) hsa().signal_store_relaxed(d.done[k], 0);
    set_error(SMR_EHIP, "direct dispatch: " + msg);
}

// Bounded wait for a completion signal.  false: the queue reported an error or NOTHING MOVED for the time limit -- the device's
// direct path is marked failed (the caller unwinds; nothing waits on these queues again).  The limit is on the absence of progress,
// not on the wait: a replay of a million launches is allowed to take its time, the clock restarts whenever any of the device's
// queues has consumed packets since the last look.
bool wait_signal(Direct& d, hsa_signal_t sig) {
    Hsa& h = hsa();
    if (h.signal_load(sig) == 0) return true;
    if (d.failed.load()) {
        direct_fail(d, "the HSA queue reported an error");
        return false;
    }
    double t0 = now_s();
    const double limit = direct_timeout_s();
    uint64_t seen[SEQ_MAXQ] = {};
    for (int k = 0; k < SEQ_MAXQ; ++k)
        if (d.q[k]) seen[k] = h.load_read_index(d.q[k]);
    for (;;) {
        // timeout hint in timestamp ticks (100 MHz on this part: ~2 ms per slice)
        if (h.signal_wait(sig, HSA_SIGNAL_CONDITION_EQ, 0, 200000, HSA_WAIT_STATE_ACTIVE) == 0) return true;
        if (d.failed.load()) {
            direct_fail(d, "the HSA queue reported an error");
            return false;
        }
        for (int k = 0; k < SEQ_MAXQ; ++k)
            if (d.q[k]) {
                const uint64_t r = h.load_read_index(d.q[k]);
                if (r != seen[k]) {
                    seen[k] = r;
                    t0 = now_s();
                }
            }
        if (now_s() - t0 > limit) {
            direct_fail(d, "no hardware queue made progress and the completion signal did not arrive within the time limit ($SMR_DIRECT_TIMEOUT_MS)");
            return false;
        }
    }
}

std::mutex g_direct_mu;
std::map<int, Direct*> g_direct;

// queue k of the device (created on first use)
int direct_queue(Direct& d, int k) {
    if (d.q[k]) return SMR_OK;
    Hsa& h = hsa();
    hsa_status_t st = h.queue_create(d.agent, 16384, HSA_QUEUE_TYPE_SINGLE, queue_error, &d, UINT32_MAX, UINT32_MAX, &d.q[k]);
    if (st != HSA_STATUS_SUCCESS) {
        d.q[k] = nullptr;
        d.why = "hsa_queue_create failed";
        return set_error(SMR_EHIP, d.why);
    }
    if (h.signal_create(0, 0, nullptr, &d.done[k]) != HSA_STATUS_SUCCESS) {
        d.why = "hsa_signal_create failed";
        return set_error(SMR_EHIP, d.why);
    }
    volatile hsa_signal_value_t* vp = nullptr;
    if (h.signal_value_pointer(d.done[k], &vp) == HSA_STATUS_SUCCESS) d.done_ptr[k] = (volatile int64_t*)vp;
    return SMR_OK;
}

int current_device() {
    int dev = 0;
    (void)hipGetDevice(&dev);
    return dev;
}
// the device a stream belongs to (not the calling thread's current device: a process that drives several GPUs may submit to a
// stream of another one); the null stream belongs to the current device
int device_of(hipStream_t s) {
    if (s) {
        hipDevice_t dv = 0;
        if (hipStreamGetDevice(s, &dv) == hipSuccess) return (int)dv;
        (void)hipGetLastError();
    }
    return current_device();
}

int direct_selftest(Direct& d);

Direct& direct_of(int dev) {
    std::lock_guard<std::mutex> g(g_direct_mu);
    auto it = g_direct.find(dev);
    if (it != g_direct.end()) return *it->second;
    Direct* d = new Direct();
    d->dev = dev;
    g_direct[dev] = d;
    Hsa& h = hsa();
    if (!h.ok) {
        d->why = h.why;
        return *d;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        d->why = "hipGetDeviceProperties failed";
        return *d;
    }
    AgentPick pick;
    pick.h = &h;
    pick.bdf = ((uint32_t)prop.pciBusID << 8) | ((uint32_t)prop.pciDeviceID << 3);
    pick.domain = (uint32_t)prop.pciDomainID;
    h.iterate_agents(pick_agent, &pick);
    if (!pick.have) {
        if (pick.ngpu == 1) {
            pick.found = pick.first;  // one GPU visible: nothing to confuse
        } else {
            d->why = "no HSA agent matches the HIP device's PCI address";
            return *d;
        }
    }
    d->agent = pick.found;
    d->agent_by_pci = pick.have;
    if (int rc = direct_queue(*d, 0)) {
        (void)rc;
        return *d;
    }
    int can = 0;
    // stream-side waits (hipStreamWaitValue64 on the completion signals) are opt-in ($SMR_SEQ_STREAM_WAIT=1): the MI355X boxes this was
    // developed on report hipDeviceAttributeCanUseStreamWaitValue = 0, so only the host-side wait has run on hardware
    const char* sw = std::getenv("SMR_SEQ_STREAM_WAIT");
    if (sw && sw[0] == '1' && hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, dev) == hipSuccess && can && d->done_ptr[0]) d->wait_value_ok = true;
    d->ok = true;
    // one packet through queue 0 before anything depends on the path: a probe kernel reports the blockDim / gridDim it sees and
    // writes through its dynamic LDS.  A wrong hidden-argument layout, a ring that is not ours to write, a signal that never
    // arrives -- all end here, with the path switched off and everything going through HIP.
    if (direct_selftest(*d) != SMR_OK) d->ok = false;
    return *d;
}
Direct& direct() { return direct_of(current_device()); }

struct Lookup {
    Hsa* h;
    Direct* d;
    hsa_agent_t agent;
    std::string kd;
    KernelRef out;
    bool have = false;
};

// is [p, p + n) mapped in this process?  (the loader reports where the code object's image was when it was loaded; HIP keeps that
// memory for the life of the module, but a stale address must cost a fallback, not a fault)
bool range_mapped(const void* p, size_t n) {
    const long page = sysconf(_SC_PAGESIZE);
    if (page <= 0 || !p || !n) return false;
    const uintptr_t lo = (uintptr_t)p & ~(uintptr_t)(page - 1), hi = ((uintptr_t)p + n + (uintptr_t)page - 1) & ~(uintptr_t)(page - 1);
    std::vector<unsigned char> vec((hi - lo) / (uintptr_t)page);
    return mincore((void*)lo, hi - lo, vec.data()) == 0;
}

// the loaded code objects of the executable that holds the kernel: parse the metadata of each image once, look the kernel up
hsa_status_t scan_code_object(hsa_executable_t, hsa_loaded_code_object_t lco, void* data) {
    Lookup* l = (Lookup*)data;
    auto get = l->h->loader.hsa_ven_amd_loader_loaded_code_object_get_info;
    if (!get) return HSA_STATUS_INFO_BREAK;
    hsa_ven_amd_loader_code_object_storage_type_t st = HSA_VEN_AMD_LOADER_CODE_OBJECT_STORAGE_TYPE_NONE;
    if (get(lco, HSA_VEN_AMD_LOADER_LOADED_CODE_OBJECT_INFO_CODE_OBJECT_STORAGE_TYPE, &st) != HSA_STATUS_SUCCESS || st != HSA_VEN_AMD_LOADER_CODE_OBJECT_STORAGE_TYPE_MEMORY)
        return HSA_STATUS_SUCCESS;
    uint64_t base = 0, size = 0;
    if (get(lco, HSA_VEN_AMD_LOADER_LOADED_CODE_OBJECT_INFO_CODE_OBJECT_STORAGE_MEMORY_BASE, &base) != HSA_STATUS_SUCCESS ||
        get(lco, HSA_VEN_AMD_LOADER_LOADED_CODE_OBJECT_INFO_CODE_OBJECT_STORAGE_MEMORY_SIZE, &size) != HSA_STATUS_SUCCESS || !base || !size)
        return HSA_STATUS_SUCCESS;
    auto it = l->d->code_objects.find(base);
    if (it == l->d->code_objects.end()) {
        std::map<std::string, KernargLayout> kernels;
        std::string why;
        if (range_mapped((const void*)base, (size_t)size)) (void)kmeta_parse((const void*)base, (size_t)size, kernels, why);
        it = l->d->code_objects.emplace(base, std::move(kernels)).first;  // (an image that did not parse stays as an empty entry)
    }
    auto k = it->second.find(l->kd);
    if (k == it->second.end()) return HSA_STATUS_SUCCESS;
    l->out.layout = k->second;
    l->out.from_metadata = true;
    return HSA_STATUS_INFO_BREAK;
}

hsa_status_t lookup_exec(hsa_executable_t ex, void* data) {
    Lookup* l = (Lookup*)data;
    hsa_executable_symbol_t sym;
    if (l->h->get_symbol_by_name(ex, l->kd.c_str(), &l->agent, &sym) != HSA_STATUS_SUCCESS) return HSA_STATUS_SUCCESS;
    hsa_symbol_kind_t kind;
    if (l->h->symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_TYPE, &kind) != HSA_STATUS_SUCCESS || kind != HSA_SYMBOL_KIND_KERNEL) return HSA_STATUS_SUCCESS;
    l->h->symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &l->out.object);
    l->h->symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &l->out.kernarg_size);
    l->h->symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &l->out.group_static);
    l->h->symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &l->out.private_size);
    l->have = true;
    // the hidden-argument layout, from the metadata of the code object this executable was loaded from
    const char* off = std::getenv("SMR_DIRECT_METADATA");
    if (!(off && off[0] == '0') && l->h->loader.hsa_ven_amd_loader_executable_iterate_loaded_code_objects)
        (void)l->h->loader.hsa_ven_amd_loader_executable_iterate_loaded_code_objects(ex, scan_code_object, l);
    return HSA_STATUS_INFO_BREAK;
}

// kernel name -> kernel descriptor among the executables loaded in this process
int resolve_name(Direct& d, const std::string& name, KernelRef& out) {
    Lookup l;
    l.h = &hsa();
    l.d = &d;
    l.agent = d.agent;
    l.kd = name + ".kd";
    l.h->loader.hsa_ven_amd_loader_iterate_executables(lookup_exec, &l);
    if (!l.have) return set_error(SMR_EUNSUPPORTED, std::string("direct dispatch: kernel descriptor not found for ") + name);
    l.out.name = name;
    if (l.out.from_metadata) {
        // the runtime's view of the kernel and the metadata's must agree, and nothing only HIP could supply may be asked for
        if ((uint32_t)l.out.layout.kernarg_size != l.out.kernarg_size)
            return set_error(SMR_EUNSUPPORTED, "direct dispatch: kernarg size of " + name + " differs between the loader and the code object's metadata");
        if (l.out.layout.needs_runtime) return set_error(SMR_EUNSUPPORTED, "direct dispatch: " + name + " declares hidden arguments only the HIP runtime can supply");
        ++d.n_meta;
    } else {
        ++d.n_v5;  // layout by the code-object-v5 rule once the explicit size is known (kernarg_layout_of); trusted only because the self-test passed with it
    }
    out = l.out;
    return SMR_OK;
}

// The layout to fill for a launch whose explicit arguments occupy `explicit_bytes`: the metadata's when it was found -- and then the
// recorded explicit block must end where the metadata says the explicit arguments end -- else the code-object-v5 rule.
// false: this launch cannot be dispatched directly.
bool kernarg_layout_of(const KernelRef& k, size_t explicit_bytes, KernargLayout& L) {
    if (k.from_metadata) {
        // the recorder packs the explicit arguments the way the kernarg segment holds them: its block must cover every explicit
        // argument the metadata lists and must not reach into the first hidden field
        int32_t first_hidden = k.layout.kernarg_size;
        auto lower = [&](int32_t off) {
            if (off >= 0) first_hidden = std::min(first_hidden, off);
        };
        for (int d3 = 0; d3 < 3; ++d3) {
            lower(k.layout.block_count[d3]);
            lower(k.layout.group_size[d3]);
            lower(k.layout.remainder[d3]);
            lower(k.layout.global_offset[d3]);
        }
        lower(k.layout.grid_dims);
        lower(k.layout.dynamic_lds);
        if ((size_t)k.layout.explicit_end > explicit_bytes || explicit_bytes > (size_t)first_hidden) return false;
        L = k.layout;
        return true;
    }
    L = kmeta_v5_default(explicit_bytes, k.kernarg_size);
    return true;
}

// host stub -> kernel descriptor in the code object HIP has loaded
int resolve_kernel(Direct& d, const void* hostfn, KernelRef& out) {
    auto it = d.kernels.find(hostfn);
    if (it != d.kernels.end()) {
        out = it->second;
        return SMR_OK;
    }
    hipFuncAttributes attr;  // forces HIP to load the code object that holds the kernel (deferred loading)
    hipError_t e = hipFuncGetAttributes(&attr, hostfn);
    if (e != hipSuccess) return hip_error(e, "hipFuncGetAttributes (sequence build)");
    const char* name = hipKernelNameRefByPtr(hostfn, nullptr);
    if (!name || !*name) return set_error(SMR_EUNSUPPORTED, "direct dispatch: HIP does not know the kernel's name");
    const int rc = resolve_name(d, name, out);
    if (rc) return rc;
    d.kernels[hostfn] = out;
    return SMR_OK;
}

// ---- self-test: one hand-built packet before anything relies on the path ------------------------------------------------------------
// out[0] = blockDim.x, out[1] = gridDim.x (both read by the compiler from the HIDDEN arguments this file fills), out[2] = a value that
// went through the dynamic LDS the packet asked for, out[3] = number of workgroups that ran.
__global__ void k_direct_probe(unsigned* out, unsigned magic) {
    extern __shared__ unsigned probe_lds[];
    probe_lds[threadIdx.x] = magic + threadIdx.x;
    __syncthreads();
    if (threadIdx.x == 0) {
        if (blockIdx.x == 0) {
            out[0] = blockDim.x;
            out[1] = gridDim.x;
            out[2] = probe_lds[blockDim.x - 1];
        }
        atomicAdd(&out[3], 1u);
    }
}

// a kernel on the caller's stream that holds the stream back until the replay's completion signals (host memory) read zero: makes
// smr_seq_run asynchronous on devices without hipStreamWaitValue64.  Bounded: after ~`limit` ticks of the 100 MHz device clock it
// gives up and raises *gave_up (the host reports it from smr_seq_wait).
__global__ void k_seq_hold(const volatile long long* const* sigs, int n, unsigned long long limit, unsigned* gave_up) {
    if (threadIdx.x != 0) return;
    const unsigned long long t0 = wall_clock64();
    for (int k = 0; k < n; ++k)
        while (__hip_atomic_load((const long long*)sigs[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) {
            __builtin_amdgcn_s_sleep(64);
            if (wall_clock64() - t0 > limit) {
                __hip_atomic_store(gave_up, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                return;
            }
        }
}

int direct_selftest(Direct& d) {
    const char* off = std::getenv("SMR_DIRECT_SELFTEST");
    if (off && off[0] == '0') return SMR_OK;
    if (off && std::strcmp(off, "fail") == 0) {  // tests: the path a failing self-test takes
        d.why = "direct-dispatch self-test: forced failure ($SMR_DIRECT_SELFTEST=fail)";
        return SMR_EUNSUPPORTED;
    }
    Hsa& h = hsa();
    unsigned* out = nullptr;
    void* kargs = nullptr;
    auto done = [&](int rc, const std::string& why) {
        if (out) (void)hipHostFree(out);
        if (kargs) (void)hipHostFree(kargs);
        (void)hipGetLastError();
        if (rc != SMR_OK) d.why = "direct-dispatch self-test: " + why;
        return rc;
    };
    if (hipHostMalloc((void**)&out, 64, hipHostMallocDefault) != hipSuccess || hipHostMalloc(&kargs, 4096, hipHostMallocDefault) != hipSuccess)
        return done(SMR_EUNSUPPORTED, "hipHostMalloc failed");
    std::memset(out, 0, 64);
    KernelRef k;
    if (resolve_kernel(d, (const void*)k_direct_probe, k) != SMR_OK) return done(SMR_EUNSUPPORTED, smr_last_error());
    const unsigned grid = 5, block = 192, lds = 192 * 4, magic = 0x5eed0000u;
    std::vector<unsigned char> ex(12);  // (unsigned* out, unsigned magic) in kernarg layout
    std::memcpy(ex.data(), &out, 8);
    std::memcpy(ex.data() + 8, &magic, 4);
    KernargLayout lay;
    if (k.private_size != 0 || k.kernarg_size > 4096 || !kernarg_layout_of(k, ex.size(), lay)) return done(SMR_EUNSUPPORTED, "the probe kernel's arguments do not match its metadata");
    std::memset(kargs, 0, 4096);
    std::memcpy(kargs, ex.data(), ex.size());
    kmeta_fill_hidden(lay, (unsigned char*)kargs, grid, block, lds);
    hsa_kernel_dispatch_packet_t pk;
    std::memset(&pk, 0, sizeof pk);
    pk.setup = 1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
    pk.workgroup_size_x = (uint16_t)block;
    pk.workgroup_size_y = pk.workgroup_size_z = 1;
    pk.grid_size_x = grid * block;
    pk.grid_size_y = pk.grid_size_z = 1;
    pk.group_segment_size = k.group_static + lds;
    pk.kernel_object = k.object;
    pk.kernarg_address = kargs;
    pk.completion_signal = d.done[0];
    h.signal_store_relaxed(d.done[0], 1);
    hsa_queue_t* hq = d.q[0];
    const uint64_t idx = h.add_write_index(hq, 1);
    char* slot = (char*)hq->base_address + (idx & (hq->size - 1)) * 64;
    std::memcpy(slot + 4, (const char*)&pk + 4, 60);
    const uint16_t hdr = (uint16_t)((HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) |
                                    (HSA_FENCE_SCOPE_SYSTEM << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (HSA_FENCE_SCOPE_SYSTEM << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE));
    __atomic_store_n((uint32_t*)slot, (uint32_t)hdr | ((uint32_t)pk.setup << 16), __ATOMIC_RELEASE);
    h.signal_store_screlease(hq->doorbell_signal, (hsa_signal_value_t)idx);
    // (a short limit of its own: a first packet that does not come back within 5 s never will)
    const double t0 = now_s();
    while (h.signal_wait(d.done[0], HSA_SIGNAL_CONDITION_EQ, 0, 200000, HSA_WAIT_STATE_ACTIVE) != 0) {
        if (d.failed.load() || now_s() - t0 > std::min(5.0, direct_timeout_s())) {
            d.failed.store(true);
            {
                std::lock_guard<std::mutex> g(d.why_mu);
                d.fail_why = "the self-test packet did not complete";
            }
            // the queue may still hold the packet: the buffers stay allocated (leaked on purpose)
            out = nullptr;
            kargs = nullptr;
            return done(SMR_EUNSUPPORTED, "the probe packet did not complete (queue error or time-out)");
        }
    }
    const unsigned want2 = magic + block - 1;
    if (out[0] != block || out[1] != grid || out[2] != want2 || out[3] != grid) {
        char msg[200];
        std::snprintf(msg, sizeof msg, "the probe kernel saw blockDim %u (want %u), gridDim %u (want %u), LDS word 0x%x (want 0x%x), %u workgroups (want %u)", out[0], block,
                      out[1], grid, out[2], want2, out[3], grid);
        return done(SMR_EUNSUPPORTED, msg);
    }
    d.probe_ok = true;
    d.probe_meta = k.from_metadata;
    return done(SMR_OK, "");
}

// how the hidden-argument layout of this device's direct launches is known
const char* layout_source(const Direct& d) {
    if (!d.probe_ok) return d.n_meta > 0 && d.n_v5 == 0 ? "metadata" : "v5-rule(unverified)";
    if (d.n_v5 == 0) return "metadata+verified";
    return d.n_meta > 0 ? "metadata|v5-rule+verified" : "v5-rule+verified";
}

}  // namespace
}  // namespace smr


// ---- eager direct dispatch: the launches of a library-owned stream (smr_stream_create) ---------------------------------------------
// A host that routes ALL its device work through this library (the Julia shim) pays HIP's 3.6-4 us of host time per launch for kernels
// that last 2-5 us, and HIP orders every launch behind its predecessor.  On a library-owned stream the library submits the launch
// itself: the launchers run in recording mode (SMR_LAUNCH appends instead of launching), the kernel descriptor comes from the code
// object HIP loaded, the argument block goes into a ring of host-coherent slots, and ONE 64-byte packet + a doorbell go to one of up
// to four HSA queues.  Which queue is decided by the data: the bounding byte ranges of the operands (and the plan's partials) are
// compared with what is still in flight on every queue --
//   * no conflict anywhere  -> the queue with the least in flight: the launch runs CONCURRENTLY with its predecessors;
//   * conflicts on one queue -> that queue (the barrier bit orders it behind them);
//   * conflicts on several   -> one of them, behind a barrier-AND packet that waits for the last packet of each of the others.
// Every packet carries a completion signal from a per-queue ring; a signal that has reached 0 retires its launch's ranges.  Results
// are those of in-order execution on the stream (src/mapreduce.jl:203-223: spawn what is independent, wait where it must).
// The library fences by itself -- waits for every queue -- before anything it does on the stream through HIP (copies, synchronisation,
// sequence replays, the scalar result of a complete reduction), and drains HIP work it queued itself before the next direct launch.
namespace smr {
namespace {
typedef std::vector<std::pair<uintptr_t, uintptr_t>> Spans;
bool overlaps(const Spans& v, const std::pair<uintptr_t, uintptr_t>& x) {
    for (const auto& y : v)
        if (x.first < y.second && y.first < x.second) return true;
    return false;
}
bool overlaps(const Spans& v, const Spans& w) {
    for (const auto& x : w)
        if (overlaps(v, x)) return true;
    return false;
}

uint16_t header_of(bool barrier, int acq, int rel) {
    return (uint16_t)((HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | ((barrier ? 1 : 0) << HSA_PACKET_HEADER_BARRIER) |
                      (acq << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (rel << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE));
}

constexpr int EAGER_Q = 4;          // hardware queues of the eager path (the first EAGER_Q of the device's direct queues)
constexpr int EAGER_SIGS = 256;     // launches in flight per queue
constexpr size_t EAGER_SLOT = 8192; // bytes of argument block per launch (TiledArgs<true> + hidden block fit)
struct Inflight {
    int sig;  // index into EagerQueue::sigs
    Spans rd, wr;
};
struct EagerQueue {
    std::vector<hsa_signal_t> sigs;
    std::vector<unsigned char> dep_user;  // bit k: a barrier-AND packet on queue k names this signal (several queues may name the same tail)
    std::vector<Inflight> inflight;  // oldest first
    unsigned next = 0;               // next signal / argument slot
    int tail = -1;                   // signal index of the last packet submitted when it carries one (-1: it does not, or nothing was submitted since the last fence)
    int unsignaled = 0;              // packets at the tail without a completion signal (0 with tail == -1: the queue is idle as far as we know)
    unsigned char* kargs = nullptr;  // EAGER_SIGS slots of EAGER_SLOT bytes, host-coherent
};
struct Eager {
    EagerQueue q[EAGER_Q];
    bool ready = false, failed = false, fail_reported = false;
    bool kargs_device = false, gpu_only_signals = false;
    // resident argument blocks (device memory only): a bump arena behind the per-queue rings; when it is full everything in flight is
    // waited for and the arena starts over (blocks of an older epoch are stale)
    unsigned char* arena = nullptr;
    size_t arena_bytes = 0, arena_used = 0;
    unsigned long long epoch = 1;
    long n_arg_hits = 0;
    std::set<hipStream_t> hip_pending;  // owned streams on which the library queued HIP work (a copy, a fallback launch) since their last drain:
                                        // a direct launch on stream s waits for s's own HIP work only -- streams are ordered in themselves, not among each other
    unsigned sys_acquire = ~0u; // bit k: the next direct launch on queue k follows work of another agent (a copy, a table upload): acquire at system scope
    std::map<std::string, std::pair<KernelRef, std::shared_ptr<void>>> jit;  // runtime-compiled kernels by entry-point name (module pinned)
    long n_launch = 0, n_free = 0, n_same = 0, n_cross = 0, n_fallback = 0;
};
std::mutex g_eager_mu;
std::map<int, Eager*> g_eager;
Eager& eager_of(int dev) {  // per device, like the direct queues it drives (one process per GPU is the usual case)
    std::lock_guard<std::mutex> g(g_eager_mu);
    Eager*& e = g_eager[dev];
    if (!e) e = new Eager();
    return *e;
}
std::vector<int> eager_devices() {
    std::lock_guard<std::mutex> g(g_eager_mu);
    std::vector<int> v;
    for (auto& kv : g_eager) v.push_back(kv.first);
    return v;
}

// a CPU agent (for hsa_amd_agents_allow_access) and a device-local pool the CPU may be given access to (large BAR)
struct PoolPick {
    Hsa* h;
    hsa_agent_t cpu{};
    bool have_cpu = false;
    hsa_amd_memory_pool_t pool{};
    bool have_pool = false;
};
hsa_status_t pick_cpu(hsa_agent_t a, void* data) {
    PoolPick* p = (PoolPick*)data;
    hsa_device_type_t t;
    if (!p->have_cpu && p->h->agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t) == HSA_STATUS_SUCCESS && t == HSA_DEVICE_TYPE_CPU) {
        p->cpu = a;
        p->have_cpu = true;
    }
    return HSA_STATUS_SUCCESS;
}
hsa_status_t pick_pool(hsa_amd_memory_pool_t pool, void* data) {
    PoolPick* p = (PoolPick*)data;
    hsa_amd_segment_t seg;
    uint32_t flags = 0;
    bool alloc = false;
    if (p->h->pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg) != HSA_STATUS_SUCCESS || seg != HSA_AMD_SEGMENT_GLOBAL) return HSA_STATUS_SUCCESS;
    p->h->pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
    p->h->pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_RUNTIME_ALLOC_ALLOWED, &alloc);
    if (!alloc || !(flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_COARSE_GRAINED)) return HSA_STATUS_SUCCESS;
    hsa_amd_memory_pool_access_t acc = HSA_AMD_MEMORY_POOL_ACCESS_NEVER_ALLOWED;
    if (p->h->agent_pool_get_info(p->cpu, pool, HSA_AMD_AGENT_MEMORY_POOL_INFO_ACCESS, &acc) != HSA_STATUS_SUCCESS || acc == HSA_AMD_MEMORY_POOL_ACCESS_NEVER_ALLOWED)
        return HSA_STATUS_SUCCESS;
    p->pool = pool;
    p->have_pool = true;
    return HSA_STATUS_INFO_BREAK;
}

// Argument blocks: device memory the host writes through the BAR (what HIP itself does on this part: a kernel that fetches its
// arguments from host memory starts a PCIe round trip later), when the device-local pool can be mapped for the CPU; else pinned
// host memory.  $SMR_EAGER_KERNARG = host | device forces one.
unsigned char* eager_kernarg_ring(Direct& d, Eager& e, size_t bytes) {
    Hsa& h = hsa();
    const char* force = std::getenv("SMR_EAGER_KERNARG");
    const bool want_dev = !(force && std::strcmp(force, "host") == 0);
    if (want_dev && h.iterate_pools && h.pool_get_info && h.agent_pool_get_info && h.pool_allocate && h.allow_access) {
        PoolPick pp;
        pp.h = &h;
        h.iterate_agents(pick_cpu, &pp);
        if (pp.have_cpu) h.iterate_pools(d.agent, pick_pool, &pp);
        void* p = nullptr;
        if (pp.have_pool && h.pool_allocate(pp.pool, bytes, 0, &p) == HSA_STATUS_SUCCESS && p) {
            hsa_agent_t both[2] = {pp.cpu, d.agent};
            if (h.allow_access(2, both, nullptr, p) == HSA_STATUS_SUCCESS) {
                e.kargs_device = true;
                return (unsigned char*)p;
            }
        }
    }
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return (unsigned char*)p;
}

int eager_init(Direct& d, Eager& e) {
    if (e.ready) return SMR_OK;
    if (e.failed) return SMR_EUNSUPPORTED;
    Hsa& h = hsa();
    const char* sg = std::getenv("SMR_EAGER_SIGNALS");  // "interrupt": ordinary signals (experiments)
    const bool gpu_only = h.amd_signal_create && !(sg && std::strcmp(sg, "interrupt") == 0);
    e.gpu_only_signals = gpu_only;
    constexpr size_t ARENA = (size_t)16 << 20;
    unsigned char* ring = eager_kernarg_ring(d, e, (size_t)EAGER_Q * EAGER_SIGS * EAGER_SLOT + ARENA);
    if (!ring) {
        e.failed = true;
        return SMR_EUNSUPPORTED;
    }
    if (e.kargs_device) {
        e.arena = ring + (size_t)EAGER_Q * EAGER_SIGS * EAGER_SLOT;
        e.arena_bytes = ARENA;
    }
    for (int k = 0; k < EAGER_Q; ++k) {
        if (direct_queue(d, k) != SMR_OK) {
            e.failed = true;
            return SMR_EUNSUPPORTED;
        }
        EagerQueue& q = e.q[k];
        q.sigs.resize(EAGER_SIGS);
        q.dep_user.assign(EAGER_SIGS, 0);
        for (int i = 0; i < EAGER_SIGS; ++i) {
            // completion signals are polled by the host and consumed by barrier-AND packets: no interrupt, no event mailbox write
            const hsa_status_t st = gpu_only ? h.amd_signal_create(0, 0, nullptr, HSA_AMD_SIGNAL_AMD_GPU_ONLY, &q.sigs[i]) : h.signal_create(0, 0, nullptr, &q.sigs[i]);
            if (st != HSA_STATUS_SUCCESS) {
                e.failed = true;
                return SMR_EUNSUPPORTED;
            }
        }
        q.kargs = ring + (size_t)k * EAGER_SIGS * EAGER_SLOT;
    }
    e.ready = true;
    return SMR_OK;
}

// drop the launches whose completion signal has reached 0 (in submission order: a queue completes in order)
bool put_packet(Direct& d, hsa_queue_t* hq, const void* body64, uint16_t header, uint16_t setup);
void eager_wait_queue(Eager& e, Direct& d, int k);

// (a queue completes in order: every packet carries the barrier bit; an entry without a signal of its own retires with the next
// signalled one behind it)
void eager_retire(EagerQueue& q) {
    Hsa& h = hsa();
    size_t done = 0;
    for (size_t i = 0; i < q.inflight.size(); ++i) {
        if (q.inflight[i].sig < 0) continue;
        if (h.signal_load(q.sigs[q.inflight[i].sig]) != 0) break;
        done = i + 1;
    }
    if (done) q.inflight.erase(q.inflight.begin(), q.inflight.begin() + (long)done);
}

int eager_take_signal(Eager& e, Direct& d, EagerQueue& q, int self);

// a marker: an empty barrier packet that completes when everything submitted to the queue before it has; returns its signal index
int eager_marker(Eager& e, Direct& d, EagerQueue& q, hsa_queue_t* hq, int self) {
    const int si = eager_take_signal(e, d, q, self);
    hsa_barrier_and_packet_t bp;
    std::memset(&bp, 0, sizeof bp);
    bp.completion_signal = q.sigs[si];
    const uint16_t hdr = (uint16_t)((HSA_PACKET_TYPE_BARRIER_AND << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER));
    (void)put_packet(d, hq, &bp, hdr, 0);
    Inflight f;
    f.sig = si;
    q.inflight.push_back(std::move(f));
    q.tail = si;
    q.unsignaled = 0;
    return si;
}

// (after a failure nothing is waited for any more: the bookkeeping is dropped, the caller learns about it from d.failed)
void eager_wait_queue(Eager& e, Direct& d, int k) {
    EagerQueue& q = e.q[k];
    if (!d.failed.load()) {
        if (q.unsignaled > 0) (void)eager_marker(e, d, q, d.q[k], k);
        if (q.tail >= 0) (void)wait_signal(d, q.sigs[q.tail]);
    }
    q.inflight.clear();
    q.tail = -1;
    q.unsignaled = 0;
}

// the next completion signal of queue `self` (a ring): the packet that used it EAGER_SIGS signalled submissions ago must have completed,
// and a barrier-AND packet of another queue that names it must have passed, before it is re-armed
int eager_take_signal(Eager& e, Direct& d, EagerQueue& q, int self) {
    Hsa& h = hsa();
    const int si = (int)(q.next % EAGER_SIGS);
    ++q.next;
    (void)wait_signal(d, q.sigs[si]);
    if (const unsigned users = q.dep_user[si]) {  // every queue whose barrier-AND packet names it must have consumed that packet
        q.dep_user[si] = 0;
        for (int k = 0; k < EAGER_Q; ++k)
            if (((users >> k) & 1u) && k != self) eager_wait_queue(e, d, k);
    }
    eager_retire(q);
    h.signal_store_relaxed(q.sigs[si], 1);
    return si;
}

bool conflicts(const EagerQueue& q, const Spans& rd, const Spans& wr) {
    for (const Inflight& f : q.inflight)
        if (overlaps(f.wr, wr) || overlaps(f.wr, rd) || overlaps(f.rd, wr)) return true;
    return false;
}

// false: the ring stayed full until the time limit or the queue failed (the packet is NOT written; the path is marked failed)
bool put_packet(Direct& d, hsa_queue_t* hq, const void* body64, uint16_t header, uint16_t setup) {
    Hsa& h = hsa();
    if (d.failed.load()) return false;
    const uint64_t idx = h.add_write_index(hq, 1);
    if (idx - h.load_read_index(hq) >= hq->size) {  // ring full: spin briefly, then yield the core; bounded
        const double t0 = now_s();
        for (unsigned spins = 0; idx - h.load_read_index(hq) >= hq->size; ++spins) {
            if (spins < 256) {
                __builtin_ia32_pause();
            } else {
                sched_yield();
                if ((spins & 63) == 0 && (d.failed.load() || now_s() - t0 > direct_timeout_s())) {
                    direct_fail(d, "a hardware queue stayed full (the packet processor stopped consuming packets)");
                    return false;
                }
            }
        }
    }
    char* slot = (char*)hq->base_address + (idx & (hq->size - 1)) * 64;
    std::memcpy(slot + 4, (const char*)body64 + 4, 60);
    __atomic_store_n((uint32_t*)slot, (uint32_t)header | ((uint32_t)setup << 16), __ATOMIC_RELEASE);
    h.signal_store_screlease(hq->doorbell_signal, (hsa_signal_value_t)idx);
    return true;
}
}  // namespace

// smr_api.cpp: the launches of one execution, recorded by the caller; rd / wr = its footprint.  SMR_OK, an error, or
// SMR_EUNSUPPORTED when this execution has to go through HIP (the caller fences and launches normally).
int eager_submit(const Plan& plan, std::vector<RecLaunch>& launches, const std::vector<std::pair<uintptr_t, uintptr_t>>& rd,
                 const std::vector<std::pair<uintptr_t, uintptr_t>>& wr, hipStream_t s) {
    const int dev = device_of(s);  // the stream's device, not the calling thread's current one
    Direct& d = direct_of(dev);
    if (!d.ok) return SMR_EUNSUPPORTED;
    std::lock_guard<std::mutex> g(d.mu);
    Eager& e = eager_of(dev);
    if (eager_init(d, e) != SMR_OK) return SMR_EUNSUPPORTED;
    // a sequence replay still in flight on these queues (asynchronous smr_seq_run) comes first
    for (int k = 0; k < SEQ_MAXQ; ++k)
        if (d.armed[k]) {
            if (!wait_signal(d, d.done[k])) return SMR_EHIP;
            d.armed[k] = false;
        }
    // kernels first: anything that cannot be dispatched directly sends the whole execution through HIP
    std::vector<KernelRef> refs(launches.size());
    for (size_t j = 0; j < launches.size(); ++j) {
        RecLaunch& l = launches[j];
        int rc;
        if (l.hostfn) {
            rc = resolve_kernel(d, l.hostfn, refs[j]);
        } else if (!l.kname.empty()) {
            auto it = e.jit.find(l.kname);
            if (it != e.jit.end()) {
                refs[j] = it->second.first;
                rc = SMR_OK;
            } else {
                rc = resolve_name(d, l.kname, refs[j]);
                if (rc == SMR_OK) {
                    if (e.jit.size() > 512) {  // unpins the modules (looked up again on their next use): nothing in flight may still run their code
                        for (int k = 0; k < EAGER_Q; ++k) eager_wait_queue(e, d, k);
                        e.jit.clear();
                    }
                    e.jit[l.kname] = std::make_pair(refs[j], l.keep);
                }
            }
        } else {
            rc = SMR_EUNSUPPORTED;
        }
        KernargLayout lay;
        if (rc != SMR_OK || refs[j].private_size != 0 || !kernarg_layout_of(refs[j], l.args.size(), lay) ||
            std::max<size_t>(refs[j].kernarg_size, l.args.size()) > EAGER_SLOT) {
            ++e.n_fallback;
            return SMR_EUNSUPPORTED;
        }
    }
    if (e.hip_pending.count(s)) {  // what the library queued on THIS stream through HIP (a copy, a fallback launch) comes first
        hipError_t he = hipStreamSynchronize(s);
        if (he != hipSuccess) return hip_error(he, "draining the stream before a direct launch");
        e.hip_pending.erase(s);
        e.sys_acquire = ~0u;
    }
    // which queue
    int nconf = 0, conf[EAGER_Q], target = -1;
    for (int k = 0; k < EAGER_Q; ++k) {
        eager_retire(e.q[k]);
        if (conflicts(e.q[k], rd, wr)) conf[nconf++] = k;
    }
    if (nconf == 0) {
        size_t best = (size_t)-1;
        for (int k = 0; k < EAGER_Q; ++k)
            if (e.q[k].inflight.size() < best) {
                best = e.q[k].inflight.size();
                target = k;
            }
        ++e.n_free;
    } else {
        target = conf[0];
        for (int i = 1; i < nconf; ++i)
            if (e.q[conf[i]].inflight.size() > e.q[target].inflight.size()) target = conf[i];
        if (nconf == 1) ++e.n_same;
        else ++e.n_cross;
    }
    EagerQueue& q = e.q[target];
    hsa_queue_t* hq = d.q[target];
    if (nconf > 1) {  // wait (on the device) for the last packet of every other conflicting queue
        hsa_barrier_and_packet_t bp;
        std::memset(&bp, 0, sizeof bp);
        int nd = 0;
        for (int i = 0; i < nconf; ++i)
            if (conf[i] != target) {
                EagerQueue& o = e.q[conf[i]];
                if (o.unsignaled > 0) (void)eager_marker(e, d, o, d.q[conf[i]], conf[i]);  // its last packet carries no signal: a marker behind it does
                if (o.tail >= 0) {
                    bp.dep_signal[nd++] = o.sigs[o.tail];
                    o.dep_user[o.tail] |= (unsigned char)(1u << target);
                }
            }
        const uint16_t hdr = (uint16_t)((HSA_PACKET_TYPE_BARRIER_AND << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) |
                                        (HSA_FENCE_SCOPE_AGENT << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (HSA_FENCE_SCOPE_AGENT << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE));
        if (!put_packet(d, hq, &bp, hdr, 0)) return SMR_EHIP;
    }
    for (size_t j = 0; j < launches.size(); ++j) {
        const RecLaunch& l = launches[j];
        // completion signals are expensive on the device side (the packet processor updates one in host memory before it goes on: a
        // dependent chain with a signal per packet ran at 4.5 us per launch, 2.9 without): with resident argument blocks only every
        // 8th launch of a queue carries one (it retires its predecessors too; fences and cross-queue waits add a marker on demand);
        // with argument blocks in the per-launch ring slots every launch needs its own
        const bool want_sig = !e.arena || q.unsignaled >= 7;
        const int si = want_sig ? eager_take_signal(e, d, q, target) : -1;
        if (d.failed.load()) return SMR_EHIP;
        // the argument block: a resident one when this plan's launch j was issued with these very bytes before (the hot loop of a
        // host program), else a fresh block -- in the arena when there is one (it becomes resident), in the launch's ring slot otherwise
        unsigned char* b = nullptr;
        bool fresh = true;
        if (e.arena) {
            for (Plan::ArgBlock& ab : plan.eager_args)
                if (ab.launch == (int)j && ab.dev_index == dev && ab.epoch == e.epoch && ab.bytes.size() == l.args.size() && std::memcmp(ab.bytes.data(), l.args.data(), l.args.size()) == 0) {
                    b = (unsigned char*)ab.dev;
                    fresh = false;
                    ++e.n_arg_hits;
                    break;
                }
            if (!b) {
                const size_t need = (std::max<size_t>(refs[j].kernarg_size, l.args.size()) + 255) & ~(size_t)255;
                if (e.arena_used + need > e.arena_bytes) {  // start over: nothing in flight may still read an old block
                    for (int k = 0; k < EAGER_Q; ++k) eager_wait_queue(e, d, k);
                    e.arena_used = 0;
                    ++e.epoch;
                }
                b = e.arena + e.arena_used;
                e.arena_used += need;
                if (plan.eager_args.size() >= 8) plan.eager_args.erase(plan.eager_args.begin());  // a few rebinding patterns per plan
                Plan::ArgBlock ab;
                ab.launch = (int)j;
                ab.bytes = l.args;
                ab.dev = b;
                ab.dev_index = dev;
                ab.epoch = e.epoch;
                plan.eager_args.push_back(std::move(ab));
            }
        } else {
            b = q.kargs + (size_t)si * EAGER_SLOT;
        }
        if (fresh) {
            // explicit arguments, then the hidden ones at the offsets the code object's metadata names (block counts, group sizes,
            // grid dims, dynamic LDS size); staged in host memory: the block itself may be device memory behind the BAR
            KernargLayout lay;
            (void)kernarg_layout_of(refs[j], l.args.size(), lay);
            std::vector<unsigned char> img(std::max<size_t>(refs[j].kernarg_size, l.args.size()), 0);
            std::memcpy(img.data(), l.args.data(), l.args.size());
            if (img.size() >= (size_t)lay.kernarg_size) kmeta_fill_hidden(lay, img.data(), l.grid, l.block, l.lds);
            std::memcpy(b, img.data(), img.size());
            if (e.kargs_device) {  // posted writes through the BAR: a read of the last byte written returns only after they have landed
                const size_t used = std::max<size_t>(l.args.size(), refs[j].kernarg_size);
                __atomic_thread_fence(__ATOMIC_SEQ_CST);
                volatile unsigned char sink = ((volatile unsigned char*)b)[used ? used - 1 : 0];
                (void)sink;
            }
        }
        hsa_kernel_dispatch_packet_t pk;
        std::memset(&pk, 0, sizeof pk);
        pk.setup = 1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
        pk.workgroup_size_x = (uint16_t)l.block;
        pk.workgroup_size_y = pk.workgroup_size_z = 1;
        pk.grid_size_x = l.grid * l.block;
        pk.grid_size_y = pk.grid_size_z = 1;
        pk.group_segment_size = refs[j].group_static + l.lds;
        pk.kernel_object = refs[j].object;
        pk.kernarg_address = b;
        pk.completion_signal = si >= 0 ? q.sigs[si] : hsa_signal_t{0};
        // agent-scope fences like HIP's between kernels (the argument block is host-coherent memory, never cached in L2); the first
        // launch after a copy acquires at system scope
        // (a self-released launch -- write-through stores, acknowledged before its waves end -- leaves nothing dirty in an L2: no release)
        if (!put_packet(d, hq, &pk,
                        header_of(true, ((e.sys_acquire >> target) & 1u) ? HSA_FENCE_SCOPE_SYSTEM : HSA_FENCE_SCOPE_AGENT, l.self_released ? HSA_FENCE_SCOPE_NONE : HSA_FENCE_SCOPE_AGENT),
                        pk.setup))
            return SMR_EHIP;
        e.sys_acquire &= ~(1u << target);
        Inflight f;
        f.sig = si;
        if (j + 1 == launches.size()) {  // the execution's ranges retire with its LAST launch
            f.rd = rd;
            f.wr = wr;
        }
        q.inflight.push_back(std::move(f));
        q.tail = si;
        q.unsignaled = si >= 0 ? 0 : q.unsignaled + 1;
        ++e.n_launch;
        count_launch();
    }
    return SMR_OK;
}

// everything submitted directly -- on every device this process drove that way -- has completed when this returns (host wait).
// SMR_OK, or SMR_EHIP when a device's direct path failed (now or earlier, reported once): its results are then undefined.
// the replay in flight on this device's queues has completed when this returns SMR_OK (SMR_EHIP: the direct path failed)
int wait_all(Direct& d) {
    bool any = false;
    int rc = SMR_OK;
    for (int k = 0; k < SEQ_MAXQ; ++k)
        if (d.armed[k]) {
            if (rc == SMR_OK && !wait_signal(d, d.done[k])) rc = SMR_EHIP;
            d.armed[k] = false;
            any = true;
        }
    if (any && rc == SMR_OK) d.last_us = (now_s() - d.t_submit) * 1e6;
    return rc;
}
int eager_fence_all() {
    int rc = SMR_OK;
    std::vector<int> devs;
    {
        std::lock_guard<std::mutex> g(g_direct_mu);
        for (auto& kv : g_direct) devs.push_back(kv.first);
    }
    for (int dev : devs) {
        Direct& d = direct_of(dev);
        std::lock_guard<std::mutex> g(d.mu);
        Eager& e = eager_of(dev);
        const bool was_failed = d.failed.load();
        if (e.ready)
            for (int k = 0; k < EAGER_Q; ++k) eager_wait_queue(e, d, k);
        (void)wait_all(d);  // a sequence replay submitted asynchronously (smr_seq_run) shares the queues
        if (d.failed.load() && !(was_failed && e.fail_reported)) {
            e.fail_reported = true;
            {
                std::string fw;
                {
                    std::lock_guard<std::mutex> g(d.why_mu);
                    fw = d.fail_why;
                }
                rc = set_error(SMR_EHIP, "direct dispatch: " + (fw.empty() ? std::string("the HSA queue reported an error") : fw));
            }
        }
    }
    return rc;
}
void eager_note_hip_work(hipStream_t s) {
    const int dev = device_of(s);
    Direct& d = direct_of(dev);
    if (!d.ok) return;
    std::lock_guard<std::mutex> g(d.mu);
    eager_of(dev).hip_pending.insert(s);
}
void eager_forget_stream(hipStream_t s) {  // the stream is being destroyed (its handle may be reused)
    for (int dev : eager_devices()) {
        Direct& d = direct_of(dev);
        std::lock_guard<std::mutex> g(d.mu);
        eager_of(dev).hip_pending.erase(s);
    }
}
void eager_request_sys_acquire(hipStream_t s) {  // device memory was written behind the queues' backs (a table upload by hipMemcpy)
    const int dev = device_of(s);
    Direct& d = direct_of(dev);
    if (!d.ok) return;
    std::lock_guard<std::mutex> g(d.mu);
    eager_of(dev).sys_acquire = ~0u;
}
long eager_stat(int which) {
    const int dev = current_device();
    Direct& d = direct_of(dev);
    std::lock_guard<std::mutex> g(d.mu);  // (the counters are written under the same lock)
    Eager& e = eager_of(dev);
    switch (which) {
        case 0: return e.n_launch;
        case 1: return e.n_free;
        case 2: return e.n_same;
        case 3: return e.n_cross;
        case 5: return e.kargs_device ? 1 : 0;
        case 7: return e.n_arg_hits;
        case 6: return e.gpu_only_signals ? 1 : 0;
        default: return e.n_fallback;
    }
}
bool eager_available(hipStream_t s) { return direct_of(device_of(s)).ok; }
// for paths that must not create the direct queues as a side effect (freeing memory, destroying plans)
int eager_fence_if_active() {
    {
        std::lock_guard<std::mutex> g(g_direct_mu);
        if (g_direct.empty()) return SMR_OK;
    }
    return eager_fence_all();
}
}  // namespace smr

using namespace smr;

// ---- the sequence object ----------------------------------------------------------------------------------------------------------
struct SeqItem {
    smr_plan* plan;
    bool has_bases;
    void* bases[SMR_MAXM];
};
struct SeqPacket {
    hsa_kernel_dispatch_packet_t pk;  // header / completion signal filled in at replay
    bool barrier;
    bool acquire;  // the launch reads bytes some launch of the sequence writes (read-after-write, accumulation, in-place update)
    bool self_released;  // every store of the launch is written through and acknowledged before the kernel ends: no release fence needed
};
struct smr_seq {
    std::vector<SeqItem> items;
    bool built = false;
    bool aql = false;           // every launch is a precompiled kernel without scratch: replayed as AQL packets
    std::string why_not_aql;
    std::vector<SeqPacket> packets[SEQ_MAXQ];  // one replay's packets, per hardware queue
    int nq = 0;                 // queues in use (= dependency components of the recorded list, at most max_queues)
    int max_queues = 4;         // measured: beyond 4 queues of its own a process is time-multiplexed by the hardware scheduler (6 queues: 6.0 -> 11.1 us per step)
    int slices = -1;            // a component that is ONE independent-workgroup launch is cut into this many block ranges, one queue each;
                                // -1 (default): automatic, only the heaviest chain is cut (seq_build, step 3b); 1: never
    std::map<int, int> comp_slices;  // "slices:<c>": block ranges of component c alone (asymmetric: only the long chain is cut)
    int nsliced = 0;            // components that were
    int ncomp = 0;
    void* d_kernargs = nullptr;
    std::vector<std::shared_ptr<void>> keep;  // runtime-compiled programs the packets name
    int64_t runs = 0;
    int n_any = 0, n_barrier = 0, n_acquire = 0, n_self_released = 0;
    // Fences of the packets INSIDE a replay (the first packet on a queue always acquires, the last always releases, at system scope).
    //   acquire: -1 = by need (default): agent scope on a launch that reads bytes which some launch of the sequence WRITES -- a
    //            read-after-write, a reduction accumulating into its destination, an in-place map -- and none on a launch whose
    //            inputs nobody in the sequence writes: an acquire invalidates every XCD's L2, i.e. it throws away the read-only
    //            inputs the next replay (and whatever runs concurrently on the other queues) would have hit there, and it is only
    //            needed to observe another workgroup's writes.  0 none / 1 agent / 2 system force one scope (experiments).
    //   release: 1 agent (default; the next writer of the same bytes may sit on another XCD: write-after-write needs the
    //            write-back), 0 none (experiment: NOT safe in general), 2 system.
    int acq_mid = -1, rel_mid = 1;
    // Scope of a replay's FIRST acquire on every queue: -1 (default) = by the memory the sequence reads -- agent scope when every
    // operand it reads lives in device-local memory of this GPU (hipMalloc): what an acquire has to do is drop THIS device's stale
    // cache lines, and for local memory the agent-scope invalidate drops all of them, whoever wrote the new data (a DMA copy, the
    // host through the BAR, a kernel of a peer device write to memory, not into these L2s; tests/test_gpu_round5.py replays after a
    // DMA upload); system scope as soon as an operand is host memory (pinned, zero-copy) or managed, or its type cannot be told.
    // Measured: the system-scope acquire costs a replay 6-7 us (profiles/r05_seq_fixed_cost.txt).  0 / 1 / 2 force a scope.
    // The LAST release stays at system scope (default 2): measured no cheaper at agent scope.
    int rel_self = 0;  // release scope of self-released launches' packets inside a replay: 0 none (default), 1 = like the others (experiment)
    int acq_first = -1, rel_last = 2;
    int acq_first_auto = 2;  // what -1 resolved to at build time
    bool all_ordered = false;   // experiment: every packet carries the barrier bit
    bool inflight = false;      // a replay was submitted and nobody waited for it yet
    int device = -1;            // the device the sequence was built for (its plans' tables and the kernarg blocks live there)
    int async_mode = -1;        // smr_seq_run returns after the doorbells: -1 automatic (see smr_seq_run), 0 never (wait inside), 1 always
    bool held = false;          // the last replay put a holding kernel on the caller's stream
};

namespace {
// dependency components of a list of executions given their byte ranges (union-find over "one writes what the other reads or writes");
// comp[i] = component of execution i, numbered in order of first appearance; returns their number
int components_of(const std::vector<Spans>& rd, const std::vector<Spans>& wr, std::vector<int>& comp) {
    const size_t ni = rd.size();
    std::vector<int> parent(ni);
    for (size_t i = 0; i < ni; ++i) parent[i] = (int)i;
    auto find = [&](int x) {
        while (parent[x] != x) x = parent[x] = parent[parent[x]];
        return x;
    };
    for (size_t i = 0; i < ni; ++i)
        for (size_t j = i + 1; j < ni; ++j)
            if (overlaps(wr[i], wr[j]) || overlaps(wr[i], rd[j]) || overlaps(rd[i], wr[j])) parent[find((int)j)] = find((int)i);
    std::vector<int> roots;
    comp.assign(ni, 0);
    for (size_t i = 0; i < ni; ++i) {
        const int r = find((int)i);
        size_t c = 0;
        for (; c < roots.size(); ++c)
            if (roots[c] == r) break;
        if (c == roots.size()) roots.push_back(r);
        comp[i] = (int)c;
    }
    return (int)roots.size();
}

int seq_build(smr_seq* q) {
    for (auto& v : q->packets) v.clear();
    q->keep.clear();
    q->aql = false;
    q->n_any = q->n_barrier = q->n_acquire = q->n_self_released = 0;
    q->nq = 0;
    q->nsliced = 0;
    q->device = current_device();  // the plans' tables are uploaded (prepare pass below) on the calling thread's device
    Direct& d = direct_of(q->device);
    if (!d.ok) {
        std::lock_guard<std::mutex> g(d.why_mu);
        q->why_not_aql = d.why;
    }
    // 1. record every launch of every item (tables uploaded / scratch allocated by a prepare pass first)
    struct Rec {
        std::vector<RecLaunch> launches;
        Spans rd, wr;
        size_t bytes = 0;
        int comp = 0;
    };
    std::vector<Rec> recs(q->items.size());
    for (size_t i = 0; i < q->items.size(); ++i) {
        SeqItem& it = q->items[i];
        int rc = seq_execute_plan(it.plan, it.has_bases ? it.bases : nullptr, nullptr, true);
        if (rc) return rc;
        seq_footprint(it.plan, it.has_bases ? it.bases : nullptr, recs[i].rd, recs[i].wr);
        for (const auto& x : recs[i].rd) recs[i].bytes += x.second - x.first;
        for (const auto& x : recs[i].wr) recs[i].bytes += x.second - x.first;
    }
    // Self-released launches (write-through stores, no release fence: smr_device.h) pay for the dropped fence with slower stores.
    // That trade wins while everything the sequence touches stays in the caches (the bench step: 5.3 -> 4.6 us) and loses when the
    // stores go to HBM -- 40 launches rotating over 640 MiB of operands: the 4-way sum's 32-byte runs 4.95 -> 5.92 us per launch
    // (profiles/r05_bench_n1.json vs r04).  So: only when the union of all byte ranges of the sequence is at most
    // "self_release_max_total" bytes (default 128 MiB, half the Infinity Cache).
    bool allow_self;
    {
        Spans all;
        for (const Rec& r : recs) {
            all.insert(all.end(), r.rd.begin(), r.rd.end());
            all.insert(all.end(), r.wr.begin(), r.wr.end());
        }
        std::sort(all.begin(), all.end());
        uintptr_t total = 0, hi = 0;
        for (const auto& x : all) {
            const uintptr_t lo = std::max(x.first, hi);
            if (x.second > lo) total += x.second - lo;
            hi = std::max(hi, x.second);
        }
        allow_self = (i64)total <= options().self_release_max_total;
    }
    for (size_t i = 0; i < q->items.size(); ++i) {
        SeqItem& it = q->items[i];
        set_recorder(&recs[i].launches, allow_self);
        int rc = seq_execute_plan(it.plan, it.has_bases ? it.bases : nullptr, nullptr, false);
        set_recorder(nullptr);
        if (rc) return rc;
        if (recs[i].launches.empty()) return set_error(SMR_EINVAL, "smr_seq: a plan recorded no launch");
    }
    bool aql = d.ok;
    // 2. resolve kernels (the device's kernel map and queues are shared with the eager path: d.mu)
    std::vector<std::vector<KernelRef>> refs(recs.size());
    std::unique_lock<std::mutex> dlock(d.mu);
    for (size_t i = 0; aql && i < recs.size(); ++i)
        for (const RecLaunch& l : recs[i].launches) {
            KernelRef k;
            if (!l.hostfn && l.kname.empty()) {
                aql = false;
                q->why_not_aql = "a launch without a kernel identity takes part";
                break;
            }
            if (l.keep) q->keep.push_back(l.keep);
            if ((l.hostfn ? resolve_kernel(d, l.hostfn, k) : resolve_name(d, l.kname, k)) != SMR_OK) {
                aql = false;
                q->why_not_aql = smr_last_error();
                break;
            }
            if (k.private_size != 0) {
                aql = false;
                q->why_not_aql = "a kernel needs scratch memory: " + k.name;
                break;
            }
            KernargLayout lay;
            if (!kernarg_layout_of(k, l.args.size(), lay)) {
                aql = false;
                q->why_not_aql = "the recorded arguments of " + k.name + " do not match the kernarg layout in its code object's metadata";
                break;
            }
            refs[i].push_back(k);
        }
    dlock.unlock();
    q->aql = aql;
    q->built = true;
    if (!aql) return SMR_OK;
    // 3. dependency components of the recorded list.  Two executions conflict when one writes bytes the other reads or writes
    //    (an execution conflicts with its own next replay through its destination).  Executions of one component stay on ONE
    //    hardware queue, in recorded order -- every ordering the in-order result needs is then an ordering inside a queue, no
    //    cross-queue signal exists, and replay r+1 follows replay r on every queue by construction.  Different components share
    //    nothing that is written: they go to different queues (longest-processing-time first over the bytes they touch) and run
    //    concurrently -- the spawn / wait of src/mapreduce.jl:203-223 at the granularity of whole launches.
    const size_t ni = recs.size();
    std::vector<int> csize, cfirst;
    std::vector<size_t> cbytes;
    {
        std::vector<Spans> rds(ni), wrs(ni);
        for (size_t i = 0; i < ni; ++i) {
            rds[i] = recs[i].rd;
            wrs[i] = recs[i].wr;
        }
        std::vector<int> comp;
        const int nc = components_of(rds, wrs, comp);
        csize.assign(nc, 0);
        cfirst.assign(nc, -1);
        cbytes.assign(nc, 0);
        for (size_t i = 0; i < ni; ++i) {
            recs[i].comp = comp[i];
            cbytes[comp[i]] += recs[i].bytes;
            ++csize[comp[i]];
            if (cfirst[comp[i]] < 0) cfirst[comp[i]] = (int)i;
        }
    }
    const int ncomp = (int)csize.size();
    q->ncomp = ncomp;
    const int maxq = std::max(1, std::min(q->max_queues, SEQ_MAXQ));
    // 3b. slices.  A component that consists of ONE execution with ONE launch whose workgroups are independent (the launcher says so:
    //     RecLaunch::slice_kind) is cut into `slices` contiguous block ranges, each on a queue of its own: slice k of replay r+1 follows
    //     slice k of replay r in its queue, the slices of one replay write disjoint parts of the destination (a workgroup owns its
    //     tiles) and nothing else belongs to the component -- still no cross-queue ordering to express.  This is the device form of
    //     _mapreduce_threaded! (src/mapreduce.jl:195-227: the box is bisected and the halves run as concurrent tasks): while one
    //     slice drains and releases, the next replay's other slice is already running.
    std::vector<int> cslices(ncomp, 1);
    {
        // (a component of SEVERAL executions can be cut when they are all the same execution recorded repeatedly -- same plan, same
        // base pointers: an unrolled replay, slice k of one follows slice k of the previous one like the replays of a single one)
        auto sliceable = [&](int c, int ns) {
            const Rec& r = recs[cfirst[c]];
            if (!(ns > 1 && r.launches.size() == 1 && r.launches[0].slice_kind != 0 && r.launches[0].grid >= (unsigned)(64 * ns))) return false;
            const SeqItem& first = q->items[cfirst[c]];
            for (size_t i = 0; i < ni; ++i)
                if (recs[i].comp == c && (int)i != cfirst[c]) {
                    const SeqItem& it = q->items[i];
                    if (it.plan != first.plan || it.has_bases != first.has_bases || (it.has_bases && std::memcmp(it.bases, first.bases, sizeof it.bases) != 0) ||
                        recs[i].launches.size() != 1 || recs[i].launches[0].grid != r.launches[0].grid)
                        return false;
                }
            return true;
        };
        std::vector<int> want(ncomp, 1);
        for (int c = 0; c < ncomp; ++c) {
            auto it = q->comp_slices.find(c);
            const int ns = it != q->comp_slices.end() ? it->second : std::max(1, q->slices);
            if (sliceable(c, ns)) want[c] = ns;
        }
        // automatic ("slices" = -1, the default): with fewer components than 3 queues, the HEAVIEST single-launch component -- by the
        // bytes its operands span, every view counted: the 4-way sum reads its buffer through four views -- is cut in two when it
        // outweighs the lightest chain by half or more.  Measured on the bench step (profiles/r05_fence_ab.txt): perm | sum/2 | sum/2
        // 5.44 us per step against 5.96 on two queues; cutting the light chain instead, or every chain, or the heavy one in three:
        // 5.77-5.95 (every additional packet is one more release, i.e. one more write-back of all eight L2s).
        if (q->slices < 0 && q->comp_slices.empty() && ncomp >= 2 && ncomp + 1 <= std::min(maxq, 3)) {
            int heavy = -1;
            size_t lightest = (size_t)-1;
            for (int c = 0; c < ncomp; ++c) {
                lightest = std::min(lightest, cbytes[c]);
                if (sliceable(c, 2) && (heavy < 0 || cbytes[c] > cbytes[heavy])) heavy = c;
            }
            if (heavy >= 0 && cbytes[heavy] * 2 >= lightest * 3) want[heavy] = 2;
            // ... and when every launch of the sequence is self-released (write-through stores, no release fence on its packet) a
            // further packet costs no write-back: every chain that can be cut is cut in two, up to four queues (measured 4.6 us per
            // step on four queues against 5.1 on three and 5.3 on two, profiles/r05_store_mode_ab.txt)
            bool all_self = true;
            for (size_t i = 0; i < ni; ++i)
                for (const RecLaunch& l : recs[i].launches) all_self = all_self && l.self_released;
            if (all_self) {
                int total2 = 0;
                std::vector<int> w2(ncomp, 1);
                for (int c = 0; c < ncomp; ++c) {
                    w2[c] = sliceable(c, 2) ? 2 : 1;
                    total2 += w2[c];
                }
                if (total2 <= std::min(maxq, 4)) want = w2;
            }
        }
        int total = 0;
        for (int c = 0; c < ncomp; ++c) total += want[c];
        if (total <= maxq)  // every component keeps at least one queue of its own; otherwise nothing is cut
            for (int c = 0; c < ncomp; ++c)
                if (want[c] > 1) {
                    cslices[c] = want[c];
                    ++q->nsliced;
                }
    }
    // queues: sliced components own cslices[c] queues each; the others share what is left, longest-processing-time first
    std::vector<int> cqueue(ncomp, 0);
    int nextq = 0;
    for (int c = 0; c < ncomp; ++c)
        if (cslices[c] > 1) {
            cqueue[c] = nextq;
            nextq += cslices[c];
        }
    {
        std::vector<int> order;
        for (int c = 0; c < ncomp; ++c)
            if (cslices[c] == 1) order.push_back(c);
        if (!order.empty()) {
            const int nshared = std::max(1, std::min<int>(maxq - nextq, (int)order.size()));
            std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cbytes[a] > cbytes[b]; });
            std::vector<size_t> load(nshared, 0);
            for (int c : order) {
                int best = 0;
                for (int k = 1; k < nshared; ++k)
                    if (load[k] < load[best]) best = k;
                cqueue[c] = nextq + best;
                load[best] += cbytes[c];
            }
            nextq += nshared;
        }
    }
    const int nq = nextq;
    dlock.lock();
    for (int k = 0; k < nq; ++k)
        if (int rc = direct_queue(d, k)) return rc;
    dlock.unlock();
    q->nq = nq;
    // 4. kernarg blocks (explicit arguments + the code-object-v5 hidden block), one resident copy in device memory; a sliced launch
    //    has one block per slice (its block-offset field / list pointer patched)
    size_t total = 0;
    std::vector<std::vector<size_t>> offs(recs.size());
    auto blocksize = [&](size_t i, size_t j) {
        const size_t need = std::max<size_t>(refs[i][j].kernarg_size, recs[i].launches[j].args.size());
        return (need + 255) & ~(size_t)255;
    };
    for (size_t i = 0; i < recs.size(); ++i)
        for (size_t j = 0; j < recs[i].launches.size(); ++j) {
            offs[i].push_back(total);
            total += blocksize(i, j) * (size_t)cslices[recs[i].comp];
        }
    // slice s of a launch of `grid` workgroups: [lo, hi), cut at multiples of 8 (workgroup b runs on XCD b mod 8: the planners' tile
    // orders rely on it, and a slice that starts at a multiple of 8 keeps every workgroup on the XCD it had in the whole launch)
    auto slice_range = [&](unsigned grid, int ns, int s2, unsigned& lo, unsigned& hi) {
        const unsigned per = ((grid + ns - 1) / ns + 7u) & ~7u;
        lo = std::min<unsigned>(grid, per * (unsigned)s2);
        hi = std::min<unsigned>(grid, lo + per);
    };
    std::vector<unsigned char> host(total, 0);
    for (size_t i = 0; i < recs.size(); ++i)
        for (size_t j = 0; j < recs[i].launches.size(); ++j) {
            const RecLaunch& l = recs[i].launches[j];
            const int ns = cslices[recs[i].comp];
            for (int s2 = 0; s2 < ns; ++s2) {
                unsigned lo = 0, hi = l.grid;
                if (ns > 1) slice_range(l.grid, ns, s2, lo, hi);
                unsigned char* b = host.data() + offs[i][j] + (size_t)s2 * blocksize(i, j);
                std::memcpy(b, l.args.data(), l.args.size());
                if (ns > 1 && l.slice_kind == 1) {  // a 32-bit "first block" field inside the argument block
                    uint32_t v;
                    std::memcpy(&v, b + l.slice_off, 4);
                    v += lo;
                    std::memcpy(b + l.slice_off, &v, 4);
                } else if (ns > 1 && l.slice_kind == 2) {  // a pointer to a per-workgroup table: advanced by `lo` rows
                    uint64_t v;
                    std::memcpy(&v, b + l.slice_off, 8);
                    v += (uint64_t)lo * l.slice_row;
                    std::memcpy(b + l.slice_off, &v, 8);
                }
#if SMR_STAMP
                // stamp build: the last explicit argument is the launch's stamp region (16 bytes per wave, indexed by the slice-relative
                // blockIdx.x): every slice gets its own part of it
                if (ns > 1 && l.args.size() >= 8) {
                    uint64_t sp = 0;
                    std::memcpy(&sp, b + l.args.size() - 8, 8);
                    if (sp) {
                        sp += (uint64_t)lo * ((l.block + 63) / 64) * 16;
                        std::memcpy(b + l.args.size() - 8, &sp, 8);
                    }
                }
#endif
                // the hidden arguments (block counts = this slice's, group sizes, grid dims, dynamic LDS size) at the offsets the code
                // object's metadata names
                KernargLayout lay;
                (void)kernarg_layout_of(refs[i][j], l.args.size(), lay);
                if (blocksize(i, j) >= (size_t)lay.kernarg_size) kmeta_fill_hidden(lay, b, hi - lo, l.block, l.lds);
            }
        }
    if (q->d_kernargs) (void)hipFree(q->d_kernargs);
    q->d_kernargs = nullptr;
    hipError_t e = hipMalloc(&q->d_kernargs, total ? total : 256);
    if (e != hipSuccess) return hip_error(e, "hipMalloc(sequence kernargs)");
    e = hipMemcpy(q->d_kernargs, host.data(), total, hipMemcpyHostToDevice);
    if (e != hipSuccess) return hip_error(e, "hipMemcpy(sequence kernargs)");
    // 5. packets + ordering inside each queue: a launch that conflicts with none of the launches since the queue's last ordered one
    //    goes out without the barrier bit.  The decisions are those of the SECOND of two simulated replays (steady state: the first
    //    launch of a replay is judged against the tail of the previous replay on the same queue).
    // which executions read something the sequence writes (their packets acquire; everything else reads only data that is constant
    // for the whole replay and was made visible by the first packet's system-scope acquire)
    std::vector<char> raw(ni, 0);
    for (size_t i = 0; i < ni; ++i)
        for (size_t j = 0; j < ni && !raw[i]; ++j)
            if (overlaps(recs[j].wr, recs[i].rd)) raw[i] = 1;
    // where the memory the sequence READS lives (first acquire: see smr_seq::acq_first)
    q->acq_first_auto = HSA_FENCE_SCOPE_AGENT;
    for (size_t i = 0; i < ni && q->acq_first_auto == HSA_FENCE_SCOPE_AGENT; ++i)
        for (const auto& x : recs[i].rd) {
            hipPointerAttribute_t at;
            std::memset(&at, 0, sizeof at);
            const bool local = x.second > x.first && hipPointerGetAttributes(&at, (const void*)x.first) == hipSuccess && at.type == hipMemoryTypeDevice &&
                               !at.isManaged && at.device == q->device;
            if (!local) {
                (void)hipGetLastError();
                q->acq_first_auto = HSA_FENCE_SCOPE_SYSTEM;
                break;
            }
        }
    for (int k = 0; k < nq; ++k) {
        Spans wrd, wwr;
        for (int pass = 0; pass < 2; ++pass)
            for (size_t i = 0; i < ni; ++i) {
                const int c = recs[i].comp;
                if (k < cqueue[c] || k >= cqueue[c] + cslices[c]) continue;
                const int ns = cslices[c], s2 = k - cqueue[c];
                bool free_ = !(wrd.empty() && wwr.empty()) && !q->all_ordered;
                if (free_) free_ = !overlaps(wrd, recs[i].wr) && !overlaps(wwr, recs[i].wr) && !overlaps(wwr, recs[i].rd);
                if (!free_) {
                    wrd.clear();
                    wwr.clear();
                }
                wrd.insert(wrd.end(), recs[i].rd.begin(), recs[i].rd.end());
                wwr.insert(wwr.end(), recs[i].wr.begin(), recs[i].wr.end());
                if (pass == 0) continue;
                for (size_t j = 0; j < recs[i].launches.size(); ++j) {
                    const RecLaunch& l = recs[i].launches[j];
                    unsigned lo = 0, hi = l.grid;
                    if (ns > 1) slice_range(l.grid, ns, s2, lo, hi);
                    if (hi <= lo) continue;  // an empty slice (tiny grid)
                    SeqPacket sp;
                    std::memset(&sp, 0, sizeof sp);
                    sp.barrier = j > 0 || !free_;  // later launches of one execution (folding passes) depend on the first
                    sp.acquire = j > 0 || raw[i] != 0;  // ... and read its partials
                    sp.self_released = l.self_released;
                    if (sp.self_released) ++q->n_self_released;
                    (sp.barrier ? q->n_barrier : q->n_any)++;
                    if (sp.acquire) ++q->n_acquire;
                    sp.pk.setup = 1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
                    sp.pk.workgroup_size_x = (uint16_t)l.block;
                    sp.pk.workgroup_size_y = 1;
                    sp.pk.workgroup_size_z = 1;
                    sp.pk.grid_size_x = (hi - lo) * l.block;
                    sp.pk.grid_size_y = 1;
                    sp.pk.grid_size_z = 1;
                    sp.pk.private_segment_size = 0;
                    sp.pk.group_segment_size = refs[i][j].group_static + l.lds;
                    sp.pk.kernel_object = refs[i][j].object;
                    sp.pk.kernarg_address = (char*)q->d_kernargs + offs[i][j] + (size_t)s2 * blocksize(i, j);
                    q->packets[k].push_back(sp);
                }
            }
    }
    return SMR_OK;
}

// writes `reps` replays into the rings; the last packet on every queue signals that queue's d.done
int seq_submit(smr_seq* q, Direct& d, int reps) {
    Hsa& h = hsa();
    const int acq_mid = q->acq_mid, rel_mid = q->rel_mid;
    uint64_t written[SEQ_MAXQ] = {}, totalp[SEQ_MAXQ] = {};
    int open_queues = 0;
    for (int k = 0; k < q->nq; ++k) {
        totalp[k] = (uint64_t)q->packets[k].size() * (uint64_t)reps;
        d.armed[k] = totalp[k] > 0;
        if (totalp[k]) {
            h.signal_store_relaxed(d.done[k], 1);
            ++open_queues;
        }
    }
    d.t_submit = now_s();
    // the limit is on the ABSENCE OF PROGRESS, as in wait_signal: a replay longer than the rings (16384 packets per queue) that runs
    // for minutes keeps draining and keeps being fed -- the clock restarts with every packet written (it used to run
    // from t_submit, so a legitimate replay of more than SMR_DIRECT_TIMEOUT_MS was declared dead while its rings were full)
    double t_progress = d.t_submit;
    unsigned idle_rounds = 0;
    while (open_queues > 0) {
        // every ring full (the replay is longer than the rings: the packet processor has to catch up): spin briefly, then yield; bounded
        if (++idle_rounds > 64) {
            if (idle_rounds < 1024) __builtin_ia32_pause();
            else sched_yield();
            if ((idle_rounds & 1023) == 0 && (d.failed.load() || now_s() - t_progress > direct_timeout_s())) {
                direct_fail(d, "the hardware queues stopped consuming packets during a replay");
                return SMR_EHIP;
            }
        }
        for (int k = 0; k < q->nq; ++k) {
            if (written[k] >= totalp[k]) continue;
            hsa_queue_t* hq = d.q[k];
            const uint32_t mask = hq->size - 1;
            const size_t np = q->packets[k].size();
            // as many packets as the ring has room for, at most 256 per visit so that every queue is fed early
            const uint64_t widx = h.add_write_index(hq, 0);
            const uint64_t room = hq->size - (widx - h.load_read_index(hq));
            if (room == 0) continue;  // the packet processor is behind: visit the other queues, come back
            const uint64_t nthis = std::min<uint64_t>({room, totalp[k] - written[k], (uint64_t)256});
            const uint64_t base = h.add_write_index(hq, nthis);
            for (uint64_t t = 0; t < nthis; ++t) {
                const uint64_t g = written[k] + t;
                const SeqPacket& sp = q->packets[k][g % np];
                hsa_kernel_dispatch_packet_t* slot = (hsa_kernel_dispatch_packet_t*)hq->base_address + ((base + t) & mask);
                const bool first = g == 0, last = g + 1 == totalp[k];
                hsa_kernel_dispatch_packet_t pk = sp.pk;
                pk.completion_signal = last ? d.done[k] : hsa_signal_t{0};
                // body first, header last (release): the packet processor owns the slot once the header is valid
                std::memcpy((char*)slot + 4, (const char*)&pk + 4, sizeof pk - 4);
                const int acq = acq_mid >= 0 ? acq_mid : (sp.acquire ? HSA_FENCE_SCOPE_AGENT : HSA_FENCE_SCOPE_NONE);
                // release: the last packet on a queue at system scope; inside the replay agent scope -- none for a self-released
                // launch (its stores were written through and acknowledged before it ended: nothing is dirty in an L2)
                const int rel = (sp.self_released && q->rel_self == 0) ? HSA_FENCE_SCOPE_NONE : rel_mid;
                const uint16_t hdr = header_of(sp.barrier || first, first ? (q->acq_first >= 0 ? q->acq_first : q->acq_first_auto) : acq, last ? q->rel_last : rel);
                __atomic_store_n((uint32_t*)slot, (uint32_t)hdr | ((uint32_t)pk.setup << 16), __ATOMIC_RELEASE);
            }
            h.signal_store_screlease(hq->doorbell_signal, (hsa_signal_value_t)(base + nthis - 1));
            idle_rounds = 0;
            t_progress = now_s();
            written[k] += nthis;
            if (written[k] >= totalp[k]) --open_queues;
        }
    }
    return SMR_OK;
}

// the holding kernel's view of the completion signals (device-visible host memory, one block per device, made on first use)
struct HoldBlock {
    const volatile long long** sigs = nullptr;  // [SEQ_MAXQ]
    unsigned* gave_up = nullptr;
};
HoldBlock* hold_block(Direct& d) {
    static std::mutex mu;
    static std::map<int, HoldBlock> all;
    std::lock_guard<std::mutex> g(mu);
    HoldBlock& hb = all[d.dev];
    if (!hb.sigs) {
        void* p = nullptr;
        if (hipHostMalloc(&p, 256, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        std::memset(p, 0, 256);
        hb.sigs = (const volatile long long**)p;
        hb.gave_up = (unsigned*)((char*)p + 128);
    }
    return &hb;
}
}  // namespace

extern "C" {

int smr_seq_create(smr_seq** out) {
    if (!out) return set_error(SMR_EINVAL, "null out");
    *out = new (std::nothrow) smr_seq();
    return *out ? SMR_OK : set_error(SMR_ENOMEM, "out of host memory");
}

int smr_seq_add(smr_seq* q, smr_plan* plan, void* const* bases) {
    if (!q || !plan) return set_error(SMR_EINVAL, "null argument");
    if (q->inflight) return set_error(SMR_EINVAL, "smr_seq_add: a replay is in flight (smr_seq_wait first)");
    SeqItem it;
    std::memset(&it, 0, sizeof it);
    it.plan = plan;
    it.has_bases = bases != nullptr;
    if (bases)
        for (int k = 0; k < seq_nops(plan); ++k) it.bases[k] = bases[k];
    q->items.push_back(it);
    q->built = false;
    return SMR_OK;
}

int smr_seq_run(smr_seq* q, int reps, void* stream) {
    if (!q || reps < 1) return set_error(SMR_EINVAL, "smr_seq_run: null sequence or reps < 1");
    if (q->items.empty()) return set_error(SMR_EINVAL, "smr_seq_run: empty sequence");
    hipStream_t s = (hipStream_t)stream;
    if (!q->built) {
        int rc = seq_build(q);
        if (rc) return rc;
    }
    if (q->aql && device_of(s) != q->device) return set_error(SMR_EINVAL, "smr_seq_run: the stream belongs to another device than the one the sequence was built on");
    if (q->aql && !direct_of(q->device).ok) {  // the direct path failed after the sequence was built: from now on through HIP
        q->aql = false;
        q->why_not_aql = direct_of(q->device).why;
    }
    if (!q->aql) {  // HIP path: the same launches, in order on the caller's stream
        for (int r = 0; r < reps; ++r)
            for (SeqItem& it : q->items) {
                int rc = seq_execute_plan(it.plan, it.has_bases ? it.bases : nullptr, s, false);
                if (rc) return rc;
            }
        ++q->runs;
        return SMR_OK;
    }
    Direct& d = direct_of(q->device);
    if (int rc = eager_fence_all()) return rc;  // launches a library-owned stream submitted directly share these queues: they come first
    std::lock_guard<std::mutex> g(d.mu);
    // the previous replay on these queues must have completed before the completion signals are re-armed
    if (int rc = wait_all(d)) return rc;
    // whatever the caller queued on `stream` before comes first (a library-owned stream holds HIP work only when the library put it
    // there: a copy, a fallback launch -- tracked per stream; nothing pending = nothing to ask HIP about)
    hipError_t e = hipSuccess;
    const bool owned_idle = seq_stream_is_owned(s) && eager_of(q->device).hip_pending.count(s) == 0;
    if (!owned_idle) {
        e = hipStreamQuery(s);
        if (e == hipErrorNotReady) e = hipStreamSynchronize(s);
        if (e != hipSuccess) return hip_error(e, "smr_seq_run: draining the caller's stream");
        eager_of(q->device).hip_pending.erase(s);
    }
    int rc = seq_submit(q, d, reps);
    if (rc) return rc;
    q->inflight = true;
    q->held = false;
    ++q->runs;
    // ... and whatever is queued on `stream` afterwards comes last.  The call returns as soon as the doorbells are rung (the reference's
    // @spawn returns at once and `wait` is separate, src/mapreduce.jl:214-223); what keeps later work on `stream` behind the replay:
    //   * a library-owned stream (smr_stream_create): nothing -- all its work goes through this library, which waits for the replay
    //     before anything it submits or does on the stream (eager_submit, the fences in smr_api.cpp);
    //   * a HIP stream: hipStreamWaitValue64 on the completion signals where the device supports it ($SMR_SEQ_STREAM_WAIT=1),
    //     otherwise a one-wave holding kernel on the stream that polls the signals (k_seq_hold; bounded).
    // "async" = 0 restores the blocking call; smr_seq_wait is the host-side wait in every mode.
    const bool owned = seq_stream_is_owned(s);
    if (q->async_mode == 0) {
        rc = wait_all(d);
        q->inflight = false;
        return rc;
    }
    if (owned) return SMR_OK;
    if (d.wait_value_ok) {
        e = hipSuccess;
        for (int k = 0; k < q->nq && e == hipSuccess; ++k)
            if (d.armed[k]) e = hipStreamWaitValue64(s, (void*)d.done_ptr[k], 0, hipStreamWaitValueEq, 0xFFFFFFFFFFFFFFFFull);
        if (e == hipSuccess) return SMR_OK;
        (void)hipGetLastError();
        d.wait_value_ok = false;
    }
    HoldBlock* hb = hold_block(d);
    int n = 0;
    if (hb)
        for (int k = 0; k < q->nq; ++k)
            if (d.armed[k] && d.done_ptr[k]) hb->sigs[n++] = (const volatile long long*)d.done_ptr[k];
    bool all_have_ptr = hb != nullptr;
    for (int k = 0; k < q->nq; ++k)
        if (d.armed[k] && !d.done_ptr[k]) all_have_ptr = false;
    if (all_have_ptr && n > 0) {
        (void)hipGetLastError();
        // 100 MHz device clock: the kernel's own limit in ticks -- a backstop far above the host's no-progress limit (a host that
        // declares the path failed zeroes the signals, which releases the kernel: direct_fail)
        const unsigned long long limit = (unsigned long long)(std::max(direct_timeout_s() * 20.0, 600.0) * 1e8);
        hipLaunchKernelGGL(k_seq_hold, dim3(1), dim3(64), 0, s, (const volatile long long* const*)hb->sigs, n, limit, hb->gave_up);
        if (hipGetLastError() == hipSuccess) {
            q->held = true;
            return SMR_OK;
        }
    }
    rc = wait_all(d);  // no way to hold the stream back: block here
    q->inflight = false;
    return rc;
}

int smr_seq_wait(smr_seq* q) {
    if (!q) return set_error(SMR_EINVAL, "null sequence");
    if (!q->aql || !q->inflight) return SMR_OK;
    Direct& d = direct_of(q->device);
    int rc;
    {
        std::lock_guard<std::mutex> g(d.mu);
        rc = wait_all(d);
    }
    q->inflight = false;
    if (rc == SMR_OK && q->held) {
        HoldBlock* hb = hold_block(d);
        if (hb && __atomic_load_n(hb->gave_up, __ATOMIC_RELAXED)) {
            __atomic_store_n(hb->gave_up, 0u, __ATOMIC_RELAXED);
            rc = set_error(SMR_EHIP, "smr_seq: the holding kernel on the caller's stream gave up before the replay completed");
        }
    }
    return rc;
}

int smr_seq_info(smr_seq* q, char* buf, size_t buflen) {
    if (!q || !buf || !buflen) return set_error(SMR_EINVAL, "null argument");
    if (!q->built) {
        int rc = seq_build(q);
        if (rc) return rc;
    }
    if (q->aql) {
        size_t np = 0;
        for (const auto& v : q->packets) np += v.size();
        std::snprintf(buf, buflen, "backend=aql items=%zu packets=%zu components=%d sliced=%d queues=%d ordered=%d unordered=%d acquire=%s(%d of %zu packets) first_acquire=%s release=%d self_released=%d kernarg_layout=%s agent=%s stream_wait=%s last_replay_us=%.3f",
                      q->items.size(), np, q->ncomp, q->nsliced, q->nq, q->n_barrier, q->n_any, q->acq_mid < 0 ? "by-need" : (q->acq_mid == 0 ? "none" : (q->acq_mid == 1 ? "agent" : "system")),
                      q->n_acquire, np, (q->acq_first >= 0 ? q->acq_first : q->acq_first_auto) == 2 ? "system" : ((q->acq_first >= 0 ? q->acq_first : q->acq_first_auto) == 1 ? "agent" : "none"),
                      q->rel_mid, q->n_self_released, layout_source(direct_of(q->device)), direct_of(q->device).agent_by_pci ? "pci-address" : "only-gpu",
                      q->async_mode == 0 ? "host(blocking)" : (direct_of(q->device).wait_value_ok ? "hipStreamWaitValue64" : "holding-kernel|owned-stream"),
                      direct_of(q->device).last_us);
    }
    else
        std::snprintf(buf, buflen, "backend=hip items=%zu (%s)", q->items.size(), q->why_not_aql.c_str());
    return SMR_OK;
}

// Host-only view of the dependency analysis (no device needed: footprints are arithmetic on the plans' strides and base pointers):
// comp[i] = dependency component of recorded execution i.  Returns the number of components, or a negative status.
int smr_seq_components(smr_seq* q, int32_t* comp, size_t cap) {
    if (!q) return set_error(SMR_EINVAL, "null sequence");
    std::vector<Spans> rd(q->items.size()), wr(q->items.size());
    for (size_t i = 0; i < q->items.size(); ++i) seq_footprint(q->items[i].plan, q->items[i].has_bases ? q->items[i].bases : nullptr, rd[i], wr[i]);
    std::vector<int> c;
    const int n = components_of(rd, wr, c);
    if (comp)
        for (size_t i = 0; i < c.size() && i < cap; ++i) comp[i] = c[i];
    return n;
}

// Host-only view of the fence analysis (no device needed): acquire[i] = 1 when recorded execution i reads bytes some execution of the
// sequence writes (its packets acquire inside a replay; all others do not), *footprint = bytes of the union of every range the
// sequence touches, *cache_resident = 1 when that is within "self_release_max_total" (launches may then be self-released).
int smr_seq_fences(smr_seq* q, int32_t* acquire, size_t cap, int64_t* footprint, int32_t* cache_resident) {
    if (!q) return set_error(SMR_EINVAL, "null sequence");
    const size_t ni = q->items.size();
    std::vector<Spans> rd(ni), wr(ni);
    for (size_t i = 0; i < ni; ++i) seq_footprint(q->items[i].plan, q->items[i].has_bases ? q->items[i].bases : nullptr, rd[i], wr[i]);
    for (size_t i = 0; i < ni && i < cap; ++i) {
        int raw = 0;
        for (size_t j = 0; j < ni && !raw; ++j)
            if (overlaps(wr[j], rd[i])) raw = 1;
        if (acquire) acquire[i] = raw;
    }
    Spans all;
    for (size_t i = 0; i < ni; ++i) {
        all.insert(all.end(), rd[i].begin(), rd[i].end());
        all.insert(all.end(), wr[i].begin(), wr[i].end());
    }
    std::sort(all.begin(), all.end());
    uintptr_t total = 0, hi = 0;
    for (const auto& x : all) {
        const uintptr_t lo = std::max(x.first, hi);
        if (x.second > lo) total += x.second - lo;
        hi = std::max(hi, x.second);
    }
    if (footprint) *footprint = (int64_t)total;
    if (cache_resident) *cache_resident = (i64)total <= options().self_release_max_total ? 1 : 0;
    return (int)ni;
}

int smr_seq_set(smr_seq* q, const char* name, int64_t value) {
    if (!q || !name) return set_error(SMR_EINVAL, "null argument");
    if (std::strcmp(name, "fence_scope") == 0 && value >= 0 && value <= 2) {  // both fences of every inner packet at one scope
        q->acq_mid = q->rel_mid = (int)value;
        return SMR_OK;
    }
    if (std::strcmp(name, "acquire") == 0 && value >= -1 && value <= 2) {
        q->acq_mid = (int)value;
        return SMR_OK;
    }
    if (std::strcmp(name, "release") == 0 && value >= 0 && value <= 2) {
        q->rel_mid = (int)value;
        return SMR_OK;
    }
    if (std::strcmp(name, "release_self") == 0 && value >= 0 && value <= 1) {
        q->rel_self = (int)value;
        return SMR_OK;
    }
    if (std::strcmp(name, "first_acquire") == 0 && value >= -1 && value <= 2) {
        q->acq_first = (int)value;
        return SMR_OK;
    }
    if (std::strcmp(name, "last_release") == 0 && value >= 0 && value <= 2) {
        q->rel_last = (int)value;
        return SMR_OK;
    }
    if (std::strcmp(name, "async") == 0 && value >= -1 && value <= 1) {
        q->async_mode = (int)value;
        return SMR_OK;
    }
    if (q->inflight) return set_error(SMR_EINVAL, "smr_seq_set: a replay is in flight (smr_seq_wait first)");
    if (std::strcmp(name, "order") == 0) {  // experiments: 0 = every packet ordered, 1 = dependency-aware (default)
        q->all_ordered = value == 0;
        q->built = false;
        return SMR_OK;
    }
    if (std::strcmp(name, "queues") == 0 && value >= 1 && value <= SEQ_MAXQ) {  // hardware queues a replay may spread over
        q->max_queues = (int)value;
        q->built = false;
        return SMR_OK;
    }
    if (std::strcmp(name, "slices") == 0 && (value == -1 || (value >= 1 && value <= SEQ_MAXQ))) {  // block ranges per single-launch component (-1: automatic)
        q->slices = (int)value;
        q->comp_slices.clear();
        q->built = false;
        return SMR_OK;
    }
    if (std::strncmp(name, "slices:", 7) == 0 && value >= 1 && value <= SEQ_MAXQ) {  // ... of component <c> alone (numbered by first appearance)
        char* end = nullptr;
        const long c = std::strtol(name + 7, &end, 10);
        if (end == name + 7 || *end || c < 0 || c > 4096) return set_error(SMR_EINVAL, "smr_seq_set: slices:<component>");
        q->comp_slices[(int)c] = (int)value;
        q->built = false;
        return SMR_OK;
    }
    return set_error(SMR_EINVAL, "smr_seq_set: unknown name");
}

int smr_seq_destroy(smr_seq* q) {
    if (!q) return SMR_OK;
    (void)smr_seq_wait(q);
    if (q->d_kernargs) (void)hipFree(q->d_kernargs);
    delete q;
    return SMR_OK;
}

}  // extern "C"
