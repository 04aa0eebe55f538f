// smr_comm.cpp -- multi-GPU execution of the funnel from plain C (one process per GPU, RCCL over
// xGMI).  The C twin of strided.jl_amd/distributed.py for hosts that do not go through
// torch.distributed (the Julia shim of INTEGRATION.md).
//
// Decomposition = the reference's task bisection (_mapreduce_threaded!, src/mapreduce.jl:195-227)
// applied across ranks by smr_shard: maps split the slowest destination dim (disjoint slabs, no
// collective); reductions split a kept dim when one is long enough (no collective), otherwise a
// reduced dim, and the per-rank partial destinations are combined by ONE ncclAllReduce -- the
// distributed form of the reference's per-task partial slots + fold (:153-170).  `initop` and the
// existing destination content enter exactly once (rank 0); the other ranks start from the
// neutral element.
//
// The collective step (round 5): when the kept destination elements are one dense run in the destination's own element type
// (config 4: ONE Float32; sum(A; dims=3) into a contiguous matrix) the all-reduce runs IN PLACE on the destination -- a step is then
// the local kernel(s) + one ncclAllReduce, nothing else.  Strided destinations and 16-bit integers are gathered into a dense staging
// buffer and scattered back (two more launches).  Option "allreduce_f64" = 1 stages Float32 / ComplexF32 SUMS through Float64
// (every rank's partial is converted exactly, the ranks' sum is exact in Float64 up to 2^29 terms' worth of headroom, ONE rounding at
// the end) at the price of those two launches; the default keeps Float32 -- measured on config 4, the error of the whole sharded sum
// against the Float64 truth is 2-5e-8 either way (the local tree reduction dominates; <= 7 further roundings across 8 ranks:
// DESIGN section 6).
//
// RCCL is loaded lazily with dlopen (no link-time dependency; a single-GPU user never needs it).
#include <dlfcn.h>
#include <link.h>
#include <rccl/rccl.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <utility>

#include "smr_internal.h"

namespace smr {
namespace {

struct Rccl {
    void* lib = nullptr;
    bool tried = false;
    decltype(&ncclGetUniqueId) get_unique_id = nullptr;
    decltype(&ncclCommInitRank) comm_init_rank = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclAllReduce) all_reduce = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
    decltype(&ncclCommCount) comm_count = nullptr;
    decltype(&ncclCommUserRank) comm_user_rank = nullptr;
    std::string path;  // what was loaded (smr_comm_library)
};

struct CommState {
    std::mutex mu;
    Rccl r;
    ncclComm_t comm = nullptr;
    int nranks = 1, rank = 0;
    void* staging = nullptr;  // dense buffer for strided destinations (grow-only)
    size_t staging_bytes = 0;
};
CommState& st() {
    static CommState s;
    return s;
}

int find_loaded_rccl(struct dl_phdr_info* info, size_t, void* data) {
    if (info->dlpi_name && std::strstr(info->dlpi_name, "librccl.so")) {
        *(std::string*)data = info->dlpi_name;
        return 1;
    }
    return 0;
}

// Which RCCL: (1) $SMR_RCCL_LIB when set (a path; the multi-process tests point it at a shared-memory stand-in, an operator
// at a specific build); (2) a librccl the process has ALREADY loaded -- a host that brought its own (torch ships
// torch/lib/librccl.so) must not end up with two RCCL copies and two sets of bootstrap state in one process; (3) the
// system library.
int load_rccl() {
    Rccl& r = st().r;
    if (r.tried) return r.lib ? SMR_OK : set_error(SMR_EUNSUPPORTED, "librccl.so is not available");
    r.tried = true;
    const char* forced = std::getenv("SMR_RCCL_LIB");
    if (forced && *forced) {
        r.lib = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
        if (!r.lib) return set_error(SMR_EUNSUPPORTED, std::string("$SMR_RCCL_LIB could not be loaded: ") + dlerror());
        r.path = forced;
    } else {
        std::string loaded;
        dl_iterate_phdr(find_loaded_rccl, &loaded);
        if (!loaded.empty()) {
            r.lib = dlopen(loaded.c_str(), RTLD_NOW | RTLD_NOLOAD | RTLD_LOCAL);
            if (r.lib) r.path = loaded;
        }
        if (!r.lib)
            for (const char* n : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
                r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
                if (r.lib) {
                    r.path = n;
                    break;
                }
            }
    }
    if (!r.lib) return set_error(SMR_EUNSUPPORTED, "librccl.so is not available");
    r.get_unique_id = (decltype(r.get_unique_id))dlsym(r.lib, "ncclGetUniqueId");
    r.comm_init_rank = (decltype(r.comm_init_rank))dlsym(r.lib, "ncclCommInitRank");
    r.comm_destroy = (decltype(r.comm_destroy))dlsym(r.lib, "ncclCommDestroy");
    r.all_reduce = (decltype(r.all_reduce))dlsym(r.lib, "ncclAllReduce");
    r.error_string = (decltype(r.error_string))dlsym(r.lib, "ncclGetErrorString");
    r.comm_count = (decltype(r.comm_count))dlsym(r.lib, "ncclCommCount");
    r.comm_user_rank = (decltype(r.comm_user_rank))dlsym(r.lib, "ncclCommUserRank");
    if (!r.get_unique_id || !r.comm_init_rank || !r.comm_destroy || !r.all_reduce) {
        dlclose(r.lib);
        r.lib = nullptr;
        return set_error(SMR_EUNSUPPORTED, "librccl.so lacks a required symbol");
    }
    return SMR_OK;
}

int nccl_error(ncclResult_t e, const char* what) {
    Rccl& r = st().r;
    return set_error(SMR_EHIP, std::string(what) + ": " + (r.error_string ? r.error_string(e) : "RCCL error"));
}

// element type / count as RCCL sees a destination of `n` elements of `dtype`; *stage = the element type of the dense staging
// buffer (RCCL has no 16-bit integers: Int16 / UInt16 destinations are gathered into 32-bit staging elements -- exact -- and
// scattered back with the truncation of a narrowing store, which is the wrapped sum / product and the exact min / max)
bool nccl_type(int dtype, int redop, ncclDataType_t* t, size_t* mult, int* stage) {
    *mult = 1;
    *stage = dtype;
    switch (dtype) {
        case SMR_I16: *t = ncclInt32; *stage = SMR_I32; return true;
        case SMR_U16: *t = ncclUint32; *stage = SMR_U32; return true;
        case SMR_F32: *t = ncclFloat32; return true;
        case SMR_F64: *t = ncclFloat64; return true;
        case SMR_C32: *t = ncclFloat32; *mult = 2; return redop == SMR_RED_ADD;  // sums are component-wise
        case SMR_C64: *t = ncclFloat64; *mult = 2; return redop == SMR_RED_ADD;
        case SMR_I8: *t = ncclInt8; return true;
        case SMR_U8: case SMR_BOOL: *t = ncclUint8; return true;
        case SMR_I32: *t = ncclInt32; return true;
        case SMR_U32: *t = ncclUint32; return true;
        case SMR_I64: *t = ncclInt64; return true;
        case SMR_U64: *t = ncclUint64; return true;
    }
    return false;
}

// the distinct destination elements of a reduction as an operand of a map: stride-0 dims -> extent 1
void kept_box(const smr_problem* p, int64_t* dims, int64_t* count) {
    *count = 1;
    for (int i = 0; i < p->N; ++i) {
        dims[i] = p->ops[0].strides[i] == 0 ? 1 : p->dims[i];
        *count *= dims[i];
    }
}

int fill_neutral(const smr_problem* p) {
    if (p->redop <= SMR_RED_NONE || p->redop > SMR_RED_OR) return set_error(SMR_EINVAL, "smr_init_reduction: the problem has no reduction op");
    smr_problem f;
    std::memset(&f, 0, sizeof f);
    int64_t cnt;
    f.N = p->N;
    f.M = 1;
    kept_box(p, f.dims, &cnt);
    f.ops[0] = p->ops[0];
    static const uint8_t code[2] = {SMR_OP_CONST, 0};
    double c[2] = {0, 0};
    if (p->redop == SMR_RED_MUL || p->redop == SMR_RED_AND) c[0] = 1;
    if (p->redop == SMR_RED_MIN) c[0] = __builtin_huge_val();
    if (p->redop == SMR_RED_MAX) c[0] = -__builtin_huge_val();
    const int dt = p->ops[0].dtype == SMR_BOOL ? SMR_U8 : p->ops[0].dtype;
    if (dt >= SMR_I8 && (p->redop == SMR_RED_MIN || p->redop == SMR_RED_MAX)) {
        // integer destination: typemax / typemin of ITS type (the integer class saturates +-Inf to the Int64 limits;
        // -1 stored into a UInt64 is all ones)
        static const double imax[8] = {127., 32767., 2147483647., __builtin_huge_val(), 255., 65535., 4294967295., -1.};
        static const double imin[8] = {-128., -32768., -2147483648., -__builtin_huge_val(), 0., 0., 0., 0.};
        c[0] = p->redop == SMR_RED_MIN ? imax[dt - SMR_I8] : imin[dt - SMR_I8];
    }
    f.fprog = code;
    f.fprog_len = 1;
    f.fconsts = c;
    f.nconsts = 1;
    f.redop = SMR_RED_NONE;
    f.stream = p->stream;
    // dims of extent 1 keep the destination's (zero) stride harmlessly; a pure map needs non-zero ones
    for (int i = 0; i < f.N; ++i)
        if (f.ops[0].strides[i] == 0) f.ops[0].strides[i] = 1;
    return smr_mapreduce(&f);
}

// Are the kept destination elements one dense run of `count` elements (positive strides, any dim order)?
bool kept_is_dense(const smr_problem* p) {
    int64_t ext[SMR_MAXN], str[SMR_MAXN];
    int n = 0;
    for (int i = 0; i < p->N; ++i)
        if (p->ops[0].strides[i] != 0 && p->dims[i] > 1) {
            if (p->ops[0].strides[i] < 0) return false;
            ext[n] = p->dims[i];
            str[n] = p->ops[0].strides[i];
            ++n;
        }
    for (int i = 1; i < n; ++i)  // insertion sort by stride
        for (int j = i; j > 0 && str[j] < str[j - 1]; --j) {
            std::swap(str[j], str[j - 1]);
            std::swap(ext[j], ext[j - 1]);
        }
    int64_t want = 1;
    for (int i = 0; i < n; ++i) {
        if (str[i] != want) return false;
        want *= ext[i];
    }
    return true;
}

std::atomic<long> g_allreduces{0}, g_allreduces_inplace{0};

// dense <-> strided copy of the kept destination elements (dir 0: gather into staging, 1: scatter back)
int copy_kept(const smr_problem* p, void* dense, int stage_dtype, int dir) {
    smr_problem c;
    std::memset(&c, 0, sizeof c);
    int64_t cnt;
    c.N = p->N;
    c.M = 2;
    kept_box(p, c.dims, &cnt);
    smr_operand view = p->ops[0];
    view.conj = 0;  // raw element moves: the stored representation is what gets reduced ...
    smr_operand flat;
    std::memset(&flat, 0, sizeof flat);
    flat.base = dense;
    flat.dtype = stage_dtype;
    int64_t s = 1;
    for (int i = 0; i < c.N; ++i) {
        flat.strides[i] = s;
        s *= c.dims[i];
        if (view.strides[i] == 0) view.strides[i] = 1;  // extent-1 dims
    }
    c.ops[0] = dir == 0 ? flat : view;
    c.ops[1] = dir == 0 ? view : flat;
    c.redop = SMR_RED_NONE;
    c.stream = p->stream;
    return smr_mapreduce(&c);
}

}  // namespace
}  // namespace smr

using namespace smr;

int smr_comm_unique_id(void* out, size_t len) {
    if (!out || len < NCCL_UNIQUE_ID_BYTES) return set_error(SMR_EINVAL, "unique id buffer must hold 128 bytes");
    std::lock_guard<std::mutex> g(st().mu);
    int rc = load_rccl();
    if (rc) return rc;
    ncclUniqueId id;
    ncclResult_t e = st().r.get_unique_id(&id);
    if (e != ncclSuccess) return nccl_error(e, "ncclGetUniqueId");
    std::memcpy(out, id.internal, NCCL_UNIQUE_ID_BYTES);
    return SMR_OK;
}

int smr_comm_init(int nranks, int rank, const void* unique_id, size_t len) {
    if (nranks < 1 || rank < 0 || rank >= nranks) return set_error(SMR_EINVAL, "bad rank / nranks");
    std::lock_guard<std::mutex> g(st().mu);
    CommState& s = st();
    if (s.comm) return set_error(SMR_EINVAL, "communicator already initialised (smr_comm_destroy first)");
    if (nranks == 1 && !unique_id) {  // single rank: nothing to set up
        s.nranks = 1;
        s.rank = 0;
        return SMR_OK;
    }
    if (!unique_id || len < NCCL_UNIQUE_ID_BYTES) return set_error(SMR_EINVAL, "unique id (128 bytes, from rank 0's smr_comm_unique_id) required");
    int rc = load_rccl();
    if (rc) return rc;
    ncclUniqueId id;
    std::memcpy(id.internal, unique_id, NCCL_UNIQUE_ID_BYTES);
    ncclResult_t e = s.r.comm_init_rank(&s.comm, nranks, id, rank);
    if (e != ncclSuccess) {
        s.comm = nullptr;
        return nccl_error(e, "ncclCommInitRank");
    }
    s.nranks = nranks;
    s.rank = rank;
    return SMR_OK;
}

// What the COMMUNICATOR says (ncclCommUserRank / ncclCommCount), not what smr_comm_init was told: a launcher that disagrees with
// its RCCL bootstrap shows up here.  Without a communicator (single rank) the answer is (0, 1).
int smr_comm_rank(int* rank, int* nranks) {
    std::lock_guard<std::mutex> g(st().mu);
    CommState& s = st();
    int r = s.rank, n = s.nranks;
    if (s.comm) {
        if (!s.r.comm_count || !s.r.comm_user_rank) return set_error(SMR_EUNSUPPORTED, "this RCCL exports no ncclCommCount / ncclCommUserRank");
        ncclResult_t e = s.r.comm_count(s.comm, &n);
        if (e == ncclSuccess) e = s.r.comm_user_rank(s.comm, &r);
        if (e != ncclSuccess) return nccl_error(e, "ncclCommCount / ncclCommUserRank");
    }
    if (rank) *rank = r;
    if (nranks) *nranks = n;
    return SMR_OK;
}

int smr_comm_library(char* buf, size_t buflen) {
    if (!buf || !buflen) return set_error(SMR_EINVAL, "null argument");
    std::lock_guard<std::mutex> g(st().mu);
    int rc = load_rccl();
    if (rc) return rc;
    std::snprintf(buf, buflen, "%s", st().r.path.c_str());
    return SMR_OK;
}

int smr_comm_destroy(void) {
    std::lock_guard<std::mutex> g(st().mu);
    CommState& s = st();
    if (s.comm && s.r.comm_destroy) (void)s.r.comm_destroy(s.comm);
    s.comm = nullptr;
    s.nranks = 1;
    s.rank = 0;
    if (s.staging) (void)hipFree(s.staging);
    s.staging = nullptr;
    s.staging_bytes = 0;
    return SMR_OK;
}

int smr_init_reduction(const smr_problem* p) {
    if (!p) return set_error(SMR_EINVAL, "null problem");
    return fill_neutral(p);
}

int smr_mapreduce_sharded(const smr_problem* p) { return smr_mapreduce_sharded_ex(p, 0u); }

int smr_mapreduce_sharded_ex(const smr_problem* p, uint32_t local_ops) {
    if (!p) return set_error(SMR_EINVAL, "null problem");
    CommState& s = st();
    int nranks, rank;
    {
        std::lock_guard<std::mutex> g(s.mu);
        nranks = s.nranks;
        rank = s.rank;
    }
    bool have_comm;
    {
        std::lock_guard<std::mutex> g(s.mu);
        have_comm = s.comm != nullptr;
    }
    if (nranks == 1 && !have_comm) return smr_mapreduce(p);
    smr_problem sub;
    int need = 0;
    int rc = smr_shard_ex(p, nranks, rank, local_ops, &sub, &need, nullptr, nullptr, nullptr);
    if (rc) return rc;
    // a one-rank communicator (created with a unique id) still runs the collective step of a complete
    // reduction: the whole path -- gather, ncclAllReduce, scatter -- is exercised on a single GPU
    if (nranks == 1 && p->redop != SMR_RED_NONE) {
        need = 1;
        for (int i = 0; i < p->N; ++i)
            if (p->ops[0].strides[i] != 0 && p->dims[i] > 1) need = 0;
    }
    if (!need) return smr_mapreduce(&sub);
    ncclDataType_t t;
    size_t mult;
    int stage = 0;
    if (!nccl_type(p->ops[0].dtype, p->redop, &t, &mult, &stage))
        return set_error(SMR_EUNSUPPORTED, "this destination type / reduction has no RCCL all-reduce");
    if (rank != 0) {
        rc = fill_neutral(p);
        if (rc) return rc;
    }
    rc = smr_mapreduce(&sub);
    if (rc) return rc;
    int64_t kd[SMR_MAXN], count;
    kept_box(p, kd, &count);
    static const ncclRedOp_t ops[] = {ncclSum, ncclSum, ncclProd, ncclMin, ncclMax, ncclMin /* & on 0/1 */, ncclMax /* | on 0/1 */};
    // Float32 / ComplexF32 sums through Float64 staging on request (exact conversion, exact-enough sum, one rounding)
    const bool via_f64 = options().allreduce_f64 && p->redop == SMR_RED_ADD && (p->ops[0].dtype == SMR_F32 || p->ops[0].dtype == SMR_C32);
    if (via_f64) {
        stage = p->ops[0].dtype == SMR_F32 ? SMR_F64 : SMR_C64;
        t = ncclFloat64;
    }
    std::lock_guard<std::mutex> g(s.mu);
    if (!s.comm) return set_error(SMR_EINVAL, "smr_comm_init has not been called");
    if (!via_f64 && stage == p->ops[0].dtype && kept_is_dense(p)) {
        // in place: the destination's kept elements are one dense run -- local kernel(s) + ONE all-reduce, no gather / scatter
        rc = fence_for_foreign_work((hipStream_t)p->stream);  // the local kernels may have been submitted directly (library-owned stream)
        if (rc) return rc;
        char* dst = (char*)p->ops[0].base + p->ops[0].offset * (int64_t)dtype_size(p->ops[0].dtype);
        ncclResult_t e2 = s.r.all_reduce(dst, dst, (size_t)count * mult, t, ops[p->redop], s.comm, (hipStream_t)p->stream);
        if (e2 != ncclSuccess) return nccl_error(e2, "ncclAllReduce (in place)");
        ++g_allreduces;
        ++g_allreduces_inplace;
        return SMR_OK;
    }
    const size_t bytes = (size_t)count * (size_t)dtype_size(stage);
    if (bytes > s.staging_bytes) {
        if (s.staging) {
            (void)hipStreamSynchronize((hipStream_t)p->stream);
            (void)hipFree(s.staging);
            s.staging = nullptr;
        }
        hipError_t e = hipMalloc(&s.staging, bytes);
        if (e != hipSuccess) return hip_error(e, "hipMalloc(all-reduce staging)");
        s.staging_bytes = bytes;
    }
    rc = copy_kept(p, s.staging, stage, 0);
    if (rc) return rc;
    rc = fence_for_foreign_work((hipStream_t)p->stream);  // the gather above may have been submitted directly (library-owned stream)
    if (rc) return rc;
    ncclResult_t e = s.r.all_reduce(s.staging, s.staging, (size_t)count * mult, t, ops[p->redop], s.comm, (hipStream_t)p->stream);
    if (e != ncclSuccess) return nccl_error(e, "ncclAllReduce");
    ++g_allreduces;
    return copy_kept(p, s.staging, stage, 1);
}

namespace smr {
long comm_stat(int which) { return which == 0 ? g_allreduces.load() : g_allreduces_inplace.load(); }
}  // namespace smr
