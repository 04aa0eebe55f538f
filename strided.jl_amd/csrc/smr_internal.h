// smr_internal.h -- host-side structures shared by the planner, the C ABI and the kernel
// launchers of libstrided_hip.so.  Not part of the public ABI (that is include/strided_hip.h).
#pragma once

// The top part (types, ProgD, enums) is also compiled by hiprtc (SMR_JIT, see smr_jit.cpp): device
// code only there, everything host-side sits behind #ifndef SMR_JIT.
#ifndef SMR_JIT
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/strided_hip.h"
#else
#include "strided_hip.h"
#endif

#include <cstdint>

namespace smr {

typedef int64_t i64;
constexpr int MAXN = SMR_MAXN;
constexpr int MAXM = SMR_MAXM;
constexpr int MAXIN = SMR_MAXM - 1;
constexpr int STACK = 8;  // device evaluator stack depth (register-rotated)
constexpr int MAXG = 4;   // FAM_ORBIT: largest permutation group handled (slots of LDS per workgroup)
constexpr int FLATB_MAXP = 512;  // FAM_FLAT, batched form: largest contiguous block (elements)

#ifndef SMR_JIT
int set_error(int code, const std::string& msg);  // returns code
int hip_error(hipError_t e, const char* what);
#endif

inline int dtype_size(int dt) {
    switch (dt) {
        case SMR_F32: return 4;
        case SMR_F64: return 8;
        case SMR_C32: return 8;
        case SMR_C64: return 16;
        case SMR_I8: case SMR_U8: case SMR_BOOL: return 1;
        case SMR_I16: case SMR_U16: return 2;
        case SMR_I32: case SMR_U32: return 4;
        case SMR_I64: case SMR_U64: return 8;
    }
    return 0;
}

// f-program as passed to kernels by value (lands in SGPRs / scalar cache: it is
// wave-uniform, so the interpreter's control flow never diverges).
struct ProgD {
    int32_t len;
    int32_t nconst;
    uint8_t code[2 * SMR_MAXPROG];
    double consts[2 * SMR_MAXCONST];
};

// Constants of a runtime-compiled f-program: passed as the SECOND kernel argument, so that the generated
// source (and the compiled-code cache key) depends on the program's structure only -- a loop whose captured
// scalar changes every iteration compiles once.
struct JitConsts {
    double c[2 * SMR_MAXCONST];
};

// Recognised shapes of f that have a natively compiled functor (smr_device.h).
enum FKind : int {
    FK_PROG = 0,   // bytecode interpreter
    FK_IDENT = 1,  // a1                      copy!/permutedims!  (src/mapreduce.jl:2-14)
    FK_ADD2 = 2,   // a1 + a2
    FK_ADD3 = 3,   // (a1 + a2) + a3
    FK_ADD4 = 4,   // ((a1 + a2) + a3) + a4   README 4-way permuted sum
    FK_SCALE = 5,  // a1 * c  (== c * a1 bitwise for IEEE)          mul!/rmul!/lmul!
    FK_SYM = 6,    // (a1 + a2) / c           README symmetrise
    FK_AXPY = 7,   // c * a1 + a2                                    axpy!
    FK_AXPBY = 8,  // c * a1 + d * a2                                axpby!
    FK_ABS2 = 9,   // abs2(a1)                                       mapreduce(abs2, +, A)
    FK_MUL2 = 10,  // a1 * a2                                        __mul! (src/linalg.jl:130)
    FK_EXPR5 = 11, // a1*exp(c*a1) + sin(a1*a1), c real             README compute-bound map
    FK_COUNT
};

enum Family : int {
    FAM_AUTO = 0,
    FAM_GENERIC = 1,  // one thread per element, index decomposition (any strides)
    FAM_STREAM = 2,   // every operand unit-stride (or broadcast) along the fast dim
    FAM_TILED = 3,    // LDS-staged: operands with different unit-stride axes
    FAM_REDUCE_ALL = 4,
    FAM_REDUCE_PART = 5,
    FAM_ORBIT = 6,    // LDS-staged: every input is a differently permuted view of ONE buffer
    FAM_FLAT = 7      // unary transposing map whose short leading dims (extents not powers of two) are addressed as one flat run
};

#ifndef SMR_JIT
// Canonical problem: size-1 dims dropped, dims sorted (kept dims by destination stride,
// then reduced dims), destination strides made positive, jointly contiguous dims fused,
// identical inputs deduplicated.  GPU analogue of _mapreduce_fuse! + _mapreduce_order!
// (reference src/mapreduce.jl:98-139).
struct Canon {
    int N = 0, M = 0;
    int NK = 0;  // kept (non-reduced) dims come first: 0..NK-1; reduced dims NK..N-1
    i64 dims[MAXN];
    i64 strides[MAXM][MAXN];
    i64 offsets[MAXM];  // elements
    void* base[MAXM];
    int orig[MAXM];     // index of this operand in the caller's smr_problem::ops
    int dtype[MAXM], conj[MAXM], esize[MAXM];
    int ct = SMR_F32;  // compute class
    bool mixed = false;  // some operand dtype != ct
    bool bitcopy = false;  // pure move of opaque elements (integer dtypes)
    int redop = 0, initop = 0;
    double initarg[2] = {0, 0};
    ProgD prog;
    int int_wraps = 0;  // SMR_OP_WRAP_* instructions the integer class added to prog (0: the caller's program as given)
    int fkind = FK_PROG;
    double fc[4] = {0, 0, 0, 0};  // constants of the recognised functor: c=(fc0,fc1) d=(fc2,fc3)
    i64 total = 1;                // number of box elements
    i64 nout = 1;                 // number of destination elements
    i64 algbytes = 0;             // distinct operand footprints, once each
};

// Tile description for FAM_TILED.  Tile extents are powers of two so that the element
// enumeration inside a tile is pure bit slicing.
struct TilePlan {
    int nt = 0;          // number of tiled dims
    int tdim[MAXN];      // canonical dim index of tiled dim j (tdim[0] == 0)
    int tlog[MAXN];      // log2 tile extent of tiled dim j
    int tilelog = 0;     // sum tlog = log2(elements per tile)
    int nstaged = 0;     // inputs staged through LDS
    int staged[MAXM];    // per input k (1..M-1): LDS slot or -1 (read straight from global)
    int order[MAXM][MAXN];  // per input: enumeration order of tiled dims (fastest first)
    int threads = 256;
    i64 ntiles[MAXN];    // tiles per canonical dim
    i64 grid = 1;
    size_t lds_bytes = 0;
    // locality-aware execution order (empty = natural): workgroup b runs tile ord[b], 0xffffffff = idle
    std::vector<uint32_t> ord;
    int ord_groups = 0;
    bool no_persist = false;  // block-ordered lists run in the one-shot form
    int gorder[MAXN] = {0, 1, 2, 3, 4, 5, 6, 7};  // order of the grid dims (canonical dims, fastest first; dims with one tile are skipped): plan_tiles
};

// Description for FAM_ORBIT (smr_k_orbit.hip).  Every input k is a view of one buffer whose strides are
// the identity view's strides permuted by pi_k; G = <pi_k> (|G| <= MAXG).  A workgroup owns the G-orbit of
// one tile (tile extents are equal along every cycle of G, so G permutes tiles): slot a holds the buffer
// on tile g_a . t, the outputs of that tile read input k from slot (pi_k o g_a).
struct OrbitPlan {
    int ng = 0;               // |G|
    int k0 = 1;               // identity view: the input whose strides equal the destination's
    int gdim[MAXG][MAXN];     // g_a as a permutation of canonical dims: (g_a . t)[gdim[a][d]] = t[d]
    int pdim[MAXM][MAXN];     // pi_k (k = 1..M-1) the same way
    int slot[MAXG][MAXM];     // slot[a][k] = index of pi_k o g_a
    int lg[MAXN];             // log2 tile extent per canonical dim (invariant under G)
    int tilelog = 0;
    int vec = 1;              // elements per 16-byte access when alignment allows
    i64 ntiles[MAXN];
    i64 ntiles_total = 1;
    int norbits = 0;
    std::vector<uint32_t> list;  // root tile of every orbit, in execution order (grouped by super-cells)
    // The work list the kernel walks (round 6): `nslots` tile ids per workgroup (0xffffffff in the first = idle), dealt in one contiguous
    // run per XCD, and per workgroup the 2-bit fields (slot * 8 + input) * 2 of `wmap`: the LDS slot an output of that slot reads that
    // input from.  A full orbit fills the slots with g_a . t (the map is `slot` for every workgroup); orbits with a stabiliser (tiles on
    // a diagonal: 36 of the 1044 at 32^4) have fewer distinct tiles and SHARE a workgroup (option orbit_pack), so that no tile is loaded
    // or stored twice and -- at 32^4 -- the launch is 1024 workgroups, four on every CU, instead of 1048.
    int nslots = 0;
    std::vector<uint32_t> wtile;
    std::vector<uint64_t> wmap;
    // PAIR form (round 6, 4^4 cubes of 8-byte elements): a workgroup of 256 lanes owns TWO such slot sets (8 tiles: [w * 8 + b * 4 + g]);
    // where it can, set b = 1 is the orbit whose slot-0 tile is the unit-axis neighbour of set 0's, so that lane pairs {b = 0, b = 1}
    // move one 64-byte run (smr_k_orbit.hip: orbit_pair_body; tools/orbit16_probe.hip).  pmap[w * 2 + b]: the set's slot map.
    bool pair_ok = false;
    std::vector<uint32_t> ptile;
    std::vector<uint64_t> pmap;
    size_t lds_bytes = 0;
};

// Description for FAM_FLAT (smr_k_flat.hip): one side ("flat": dir 0 = destination, 1 = the input) is contiguous over a
// group of leading dims taken whole (extents multiply to R <= 64) and a power-of-two tile of the next contiguous dim p;
// the other side ("line") is unit-stride along dim q.
struct FlatPlan {
    int dir = 0, R = 1, tplog = 0, tqlog = 5;
    int p = -1, q = -1;
    int kt = 1;  // the input with the other layout; every other input has the destination's strides
    bool lshare = false;  // the line side is unit-stride along the SAME lead and continues along q (a transposition of R-element groups)
    bool fuse = false;  // the flat side's run continues along q itself (planar <-> interleaved): the R x TQ tile is one run
    bool ingroup[MAXN] = {false, false, false, false, false, false, false, false};
    int32_t roff[128];  // line-side element offset of the leading index r (round 6: leads of up to 128 elements)
};

// Two-sided form of the FLAT family: BOTH sides' memory runs are short groups of leading dims (permutedims of (5,300,300,7)
// into (7,300,300,5), of (17,33,65,31) reversed).  Side 0 = destination, 1 = input: run = leading dims taken whole (product R)
// x a tile TP of the next contiguous dim p; the two runs share no dim.
struct Flat2Plan {
    bool on = false;
    int kt = 1;  // the input with the other layout (side 1); every other input has the destination's strides
    bool shared = false;  // the input's run would continue along p[0] as well: phase 1 walks the destination run tile-index-major
    int R[2] = {1, 1}, TP[2] = {1, 1}, p[2] = {-1, -1};
    bool ingroup[2][MAXN] = {{false, false, false, false, false, false, false, false}, {false, false, false, false, false, false, false, false}};
    int32_t roff[2][64];  // roff[s][r]: element offset on the OTHER side of leading index r of side s's run
};

// FLAT, batched form (round 4): a unary map whose first g dims form ONE contiguous block of P elements on both sides (in different
// orders: a small matrix / tensor transposed) and whose next dim continues both sides right behind it -- (9,11,N) -> (11,9,N),
// batched transposes of small matrices.  A workgroup moves K consecutive batch entries: K * P contiguous elements on each side.
struct FlatBPlan {
    bool on = false;
    int g = 0, P = 0, K = 1;
    uint16_t srcoff[FLATB_MAXP];  // input offset (inside the block) of destination position r
};

struct Plan {
    Canon c;
    int family = FAM_GENERIC;
    FlatBPlan flatb;
    TilePlan tile;
    OrbitPlan orbit;
    FlatPlan flat;
    Flat2Plan flat2;
    // STREAM
    int vec = 1;        // elements per vector access
    // reductions
    bool vec_ua = false;      // STREAM: `vec` elements per access at ELEMENT alignment, rows end in a partial vector (round 5)
    mutable bool eager_seen = false;  // the plan's device tables (uploaded by hipMemcpy on its first execution) have been made visible to the direct queues
    void* scratch = nullptr;  // partials (owned), followed by RED_COUNTERS arrival counters (zero between launches)
    size_t scratch_bytes = 0;
    size_t counter_off = 0;   // byte offset of the counters inside `scratch` (set when it is allocated)
    int red_blocks = 0;
    int part_tr = 1;    // REDUCE_PART: lanes cooperating on one output
    int part_split = 1; // REDUCE_PART: chunks of the reduced range (two-pass when > 1)
    // REDUCE_PART, vectorised forms (see smr_k_reduce.hip): 1 = ROW (inputs unit-stride along the
    // first reduced dim), 2 = COL (inputs unit-stride along kept dim 0), 0 = general
    int part_kind = 0;
    int part_g0log = 0, part_g1log = 0;  // ROW: lanes of a group along the inner reduced dim / the outer index
                                         // COL: rows of a workgroup along the inner reduced dim / the outer index
    int part_txlog = 0;                  // COL: lanes along kept dim 0
    int part_col_tx = 0, part_col_v = 1; // COL, exact lane map (round 6): lanes along kept dim 0 (0 = the power of two above), valid for this vector width
    int part_col_y0 = 1, part_col_y1 = 1; //   rows of a workgroup along the inner reduced dim x along the outer index (TY = 256 / tx = y0 * y1)
    int part_xsplit = 1, part_qsplit = 1;  // split of the inner / outer reduced range over workgroups
    // TILED: per-lane index tables in device memory, one per kernel variant (built on first use)
    mutable void* lanetab[4] = {nullptr, nullptr, nullptr, nullptr};
    // first execution builds the tables below; concurrent executions of one (cached) plan from
    // several host threads serialise on this
    mutable std::shared_ptr<std::mutex> build_mu = std::make_shared<std::mutex>();
    mutable void* ordtab = nullptr;  // TILED: tile-order table in device memory (large grids)
    mutable std::vector<unsigned char> tiled_args[4];  // fully built kernel arguments per variant
    // eager direct dispatch (smr_seq.cpp): argument blocks of this plan's launches that are resident in device memory, keyed by their
    // bytes -- a repeated execution (same base pointers) reuses the block: no write through the BAR, no read-back round trip
    struct ArgBlock {
        int launch = 0;
        std::vector<unsigned char> bytes;  // explicit arguments (the key)
        void* dev = nullptr;
        int dev_index = 0;                 // the device whose arena holds the block (epochs are per device)
        unsigned long long epoch = 0;      // arena generation the block belongs to
    };
    mutable std::vector<ArgBlock> eager_args;
    std::string desc;
};

// Options (smr_set_option)
constexpr int RED_SHARDS = 16;       // shard partials per output of the two-level in-launch fold
constexpr int RED_COUNTERS = 16384;  // arrival counters per reduction plan (one per output group of a split reduction)
struct Options {
    i64 force_family = 0;
    i64 tile_log2 = 0;       // 0 = planner default
    i64 tile_order = 1;      // orbit-major tile order for inputs that are permuted views of one buffer
    i64 tile_block = -1;     // distinct arrays with >= 3 unit axes: tiles in compact blocks of this many per dim (0 = natural
                             // order, -1 = blocks of 4 for grids of >= 1024 tiles)
    i64 tile_block_min_axes = 3;  // experiment: 2 = block order for plain transposes as well
    i64 tile_block_xcd = -1; // block-ordered list in one contiguous run per XCD: 0 never, 1 always, -1 = while the operands fit the Infinity Cache
    i64 jit = 1;             // compile unrecognised f-programs with hiprtc (0 = always interpret)
    i64 reduce_col_txlog = 5;   // COL form: log2 of the lanes along kept dim 0 (cap)
    i64 reduce_part_wgs = 1024; // partial reductions with fewer workgroups than this are split until about this many run (4096 until
                                // round 3: 512-1024 is as fast or faster on every shape of tools/reduce_sweep.py, with 4x fewer partials)
    i64 reduce_row_floor = -1;  // ROW form: least log2 lanes per output (-1 = planner's rule)
    i64 flat_wide = 1;          // one-sided FLAT form for whole rows of 65..128 elements that are not a multiple of the 128-byte line: 1 = matrices of 32 MiB and more, 2 = wherever it applies
    i64 flat2 = 1;              // two-sided FLAT form (both sides' runs are short leading dims): on / off
    i64 flat2_pair = 1;         // two-sided FLAT form: move pairs of elements where a row's parity allows (4- / 8-byte element types)
    i64 flat2_bytes = 384;      // ... target bytes of a run
    i64 flat2_lead_bytes = 512; // ... both sides' unit-stride dims must be shorter than this
    i64 reduce_row_dense = 1;   // ROW form: lanes along the outputs when the inner reduced dim is at most 64 bytes and kept dim 0 is dense behind it
    i64 reduce_col_exact = 1;   // COL form: lanes along kept dim 0 sized to the row (25 lanes x 10 rows for 100 Float32) when that fills more of the workgroup than a power of two
    i64 reduce_col_narrow = 1;  // COL form: narrow the row segments when that yields reduce_part_wgs workgroups without a split
    i64 reduce_part_kind = -1;  // -1 = planner's choice; 0/1/2 force general / ROW / COL when applicable
    i64 reduce_single = 4;     // split reductions of at most this many chunks fold their partials inside the SAME launch (the workgroup
                               // arriving last at its group's counter does it); 0 = always a second launch.  Measured: 2-4 chunks
                               // -1.0..-1.7 us, 8 and more +0.3..+100 us (one counter serialises its arrivals at ~12 ns each)
    i64 reduce_blocks = 2048;  // cap on the workgroups (= partials) of a complete reduction
    i64 tiled_persist = 1;      // persistent + software-pipelined tiled kernel for grids larger than the machine
    i64 tiled_persist_wpc = 0;  // workgroups per CU of that form (0 = derived from threads / LDS)
    i64 tiled_persist_min = 32; // ... used when the work list holds at least this many rounds (measured: 64^4 / 4000^2
                                // problems with ~16 rounds are 2-12 % faster in the classic form, 128^4 / 8192^2 ones 6-18 % slower)
    i64 stream_u = 0;           // experiment: vectors per lane of the STREAM family (runtime-compiled functors only)
    i64 reduce_tree = 0;        // split reductions of up to this many chunks (beyond reduce_single) fold inside the launch through TWO levels of arrival
                                // counters (shards of ~sqrt(chunks)); 0: a second launch folds, as in rounds 1-3
    i64 eager_direct = 1;       // launches on a library-owned stream (smr_stream_create) are submitted by the library itself (AQL packets on its HSA queues,
                                // queue chosen by the data dependencies); 0: through HIP, in stream order
    i64 flat2_long = 80;        // two-sided FLAT form for LONG unit-stride dims (arrays of >= 8 MiB) whose 32 x 32 tiles would be under this many per cent full (0: off)
    i64 flatb = 1;              // FLAT family: batched form for contiguous small blocks (batched transposes of small matrices)
    i64 stream_pack_rows = 1;   // STREAM: rows of 129 .. 128*U vectors share a workgroup (U / ceil(n0v / 256) rows per lane) instead of one row segment per workgroup
    i64 tiled_vec = 1;       // 16-byte global accesses in the tiled family when alignment allows
    i64 tiled_edge_first = 1; // partly filled last tiles start first: one ragged grid dim runs slowest and backwards, several only backwards (2 / 3: one of the two forms always)
    i64 tiled_force_edge = 0; // experiment: run the bounds-checking variant of TILED even when every tile is whole
    i64 tiled_uavec = 1;     // ... and at element alignment (odd extents / row strides), partial vectors of ragged tiles element by element
    i64 orbit = 1;           // FAM_ORBIT for inputs that are permuted views of one buffer (0 = classic tiled kernel)
    i64 stream_ua = 1;          // STREAM: element-aligned 16-byte vectors + a partial vector per row for rows that are not whole aligned vectors
    i64 tiled_xpose = 1;        // HBM-sized transposing copies (one staged input, 128 x 32 tiles, whole tiles, 8- / 16-byte elements) run the lean kernel k_xpose_big
    i64 tiled_gorder = -1;      // TILED grid-dim order: 0 canonical, 1 the staged input's split unit axis second, -1 = that for HBM-sized 128 x 32 transposes
    i64 overlap_window_hip = 0; // 1: launches of an overlap window that go through HIP carry hipExtAnyOrderLaunch when independent (ignored by HIP on gfx9: default off)
    i64 allreduce_f64 = 0;      // smr_mapreduce_sharded: Float32 / ComplexF32 sums cross the ranks as Float64 (staging + two launches); default: in the destination's type
    i64 eager_self_release = 1; // launches of library-owned streams use write-through stores and their packets drop the release fence while the recently written destinations fit the caches (profiles/r05_eager_self_release.txt: bench step issued eagerly 6.19 -> 5.35 us, dependent chain 3.24 -> 2.93, independent launches 2.17 -> 1.62 us)
    i64 seq_self_release = 1;   // launches recorded for a sequence use write-through stores where the family can, and their packets drop the release fence
    i64 self_release_max_total = (i64)128 << 20; // ... and when everything the sequence touches is at most this big (half the Infinity Cache: write-through to HBM loses)
    i64 self_release_max_bytes = (i64)64 << 20;  // ... when the destination is at most this big (beyond, a launch lasts far longer than its fences)
    i64 orbit_lg = -1;       // tuning: force the log2 edge of the orbit tiles (-1 = planner's choice)
    i64 orbit_min = 150;     // pick the largest tile edge that still yields this many orbits (measured: tools/orbit_sweep.py)
    i64 orbit_pipe = -1;     // persistent pipelined ORBIT form: 0 never, 1 whenever there are more orbits than CUs, -1 = when LDS leaves one workgroup per CU
    i64 orbit_group = 2;     // super-cell edge (tiles per tiled dim) of the ORBIT work list: the orbits of one super-cell run
                             // next to each other on one XCD
    i64 stamp_base = 0, stamp_cap = 0, stamp_used = 0;  // SMR_STAMP builds: device buffer of 8-byte words for wave stamps
    i64 flat = 1;            // FAM_FLAT for transposing unary maps with short non-power-of-two leading dims (0 = TILED as in round 2)
    i64 orbit_deal = 0;      // experiment: 1 = super-cell c runs on XCD c mod 8 (instead of one contiguous run of the list per XCD)
    i64 orbit_skew = 0;      // experiment: diagonal enumeration of the ORBIT super-cells (step per super-cell along the other dims)
    i64 orbit_minrun = 16;   // shortest contiguous run (bytes) an ORBIT tile edge may have (round 3: 16 -- Float32 4^4 cubes at 32^4:
                             // 5.60 -> 4.61 us, 24^4 3.41 -> 3.01 us; larger sizes keep the 8^4 cubes)
    i64 orbit_wgs = 0;       // persistent ORBIT form: cap on the number of workgroups (0 = as many as the machine holds at once)
    i64 orbit_lds_min = 0;   // experiment: request at least this much LDS per ORBIT workgroup (limits residency)
    i64 orbit_pair = 1;      // 4^4 cubes of 8-byte elements: two orbits per workgroup, unit-axis neighbours in the lane pairs (64-byte runs in slot 0)
    i64 orbit_pack = 1;      // orbits with fewer distinct tiles than |G| share a workgroup (0: one workgroup per orbit, tiles repeated)
    i64 orbit_few = 40;      // fewer orbits than this even with the smallest admissible edge: classic tiled kernel
    i64 nt_store = -1;       // non-temporal stores: 0 never, 1 always, -1 = STREAM outputs of >= nt_stream_min bytes (default 0: all) and
                             // TILED tiles that write whole 128-byte lines (profiles/r02_nt_store_ab.txt: configs[4]
                             // 91.8 -> 80.2 us, 32^4 permutedims! 3.36 -> 2.76 us); ORBIT's 32-byte runs get slower (4.8 -> 6.3 us)
    i64 nt_stream_min = 0;
    i64 nt_load = -1;        // non-temporal loads in REDUCE_ALL: 0 never, 1 always, -1 = inputs below 2 GiB
    i64 max_lds_bytes = 65536;
    i64 tile_lg[MAXN] = {-1, -1, -1, -1, -1, -1, -1, -1};  // per canonical dim log2 tile extent override
};
Options& options();

// SMR_STAMP builds: the region (2 words per wave) of the next launch, or nullptr when no buffer is set / it is full
unsigned long long* stamp_next(size_t waves);

int canonicalise(const smr_problem* p, Canon& c);
int make_plan(const smr_problem* p, Plan& plan);
void describe(Plan& plan);

// ---- runtime compilation of f-programs without a natively compiled functor (smr_jit.cpp) ------------
// The kernel family's own source file is compiled by hiprtc with the f-program turned into a
// C++ functor (straight-line code over the same mathx<T> primitives the interpreter calls, so
// results are bit-identical) and ONE extern "C" kernel instantiating the family's body template
// for exactly the variant the launcher picked.  Modules are cached per (source, device).
constexpr int SMR_JIT_UNAVAILABLE = 1000;  // internal status: fall back to the interpreter
struct JitLaunch {
    const char* family;  // "generic" | "stream" | "tiled" | "reduce": source file smr_k_<family>.hip
    const char* tname;   // compute type as spelled in device code
    std::string entry;   // body of the extern "C" kernel: a call of the family's body template
    const char* argtype; // type of the single kernel argument `a`
    unsigned grid = 1, block = 256;
    size_t lds = 0;
    const void* args = nullptr;
    size_t argsize = 0;
};
int jit_launch(const Canon& c, const JitLaunch& l, hipStream_t s);
// Generates + compiles without loading or launching (no device needed): plan-time check and tests.
int jit_compile_only(const Canon& c, const JitLaunch& l, size_t* code_size);
std::string jit_functor_source(const Canon& c, const char* tname);
// dry run (thread-local): jit_launch() compiles only; launchers skip device allocations
bool jit_dry_run();
// prepare mode (thread-local): everything a first execution would build is built (tables uploaded, kernel
// compiled and loaded, scratch allocated) but nothing is launched; jit_no_launch() = dry run or prepare
void jit_set_prepare(bool on);
bool jit_no_launch();
void jit_set_dry_run(bool on);
size_t jit_dry_code_size();
struct JitStats {
    long compiles = 0, hits = 0, failures = 0;
    double compile_ms = 0;
};
JitStats jit_stats();

// AQL ordering of the next kernel launch of the calling thread (0 or hipExtAnyOrderLaunch), consumed by the first launch that asks
// (SMR_LAUNCH in smr_dispatch.h, jit_launch): set by the overlap window in smr_api.cpp
unsigned take_launch_flags();
// Before work that does NOT go through the library's launchers is queued on `s` (a collective, a copy): inside an overlap window / on a
// library-owned stream everything the library launched so far is ordered before it (and, on an owned stream, that work before the
// library's next direct launch).  No-op on ordinary streams.
int fence_for_foreign_work(hipStream_t s);

// Recording (smr_seq.cpp): while a sequence records, SMR_LAUNCH / jit_launch append what they WOULD launch instead of launching it
struct RecLaunch {
    const void* hostfn = nullptr;  // host stub of a precompiled kernel; nullptr = runtime-compiled: found by `kname`
    std::string kname;             // runtime-compiled: the (per-program unique) entry point of the loaded code object
    std::shared_ptr<void> keep;    // runtime-compiled: owner of the loaded module (it must outlive the packets that name its code)
    unsigned grid = 0, block = 0, lds = 0;
    // The launcher's statement that the workgroups of this launch are independent and that a contiguous block range [lo, hi) can be
    // launched on its own (smr_seq.cpp cuts single-launch components into such ranges, one hardware queue each):
    //   1: the 32-bit field at byte `slice_off` of the argument block is added to blockIdx.x by the kernel (set it to lo);
    //   2: the pointer at byte `slice_off` addresses a table with one row of `slice_row` bytes per workgroup (advance it by lo rows).
    int slice_kind = 0;
    unsigned slice_off = 0, slice_row = 0;
    // The launcher's statement that every global store of this launch is an agent-scope write-through store and that every wave waits
    // for the acknowledgements before it ends (smr_device.h: store policy 2): nothing the launch wrote is left dirty in an L2, so
    // its dispatch packet needs no release fence inside a replay.
    bool self_released = false;
    std::vector<unsigned char> args;  // the explicit kernel arguments in kernarg-segment layout
};
std::vector<RecLaunch>* recorder();  // thread-local, nullptr when nothing records
// eager direct dispatch on library-owned streams (smr_seq.cpp)
struct Plan;
int eager_submit(const Plan& plan, std::vector<RecLaunch>& launches, const std::vector<std::pair<uintptr_t, uintptr_t>>& rd,
                 const std::vector<std::pair<uintptr_t, uintptr_t>>& wr, hipStream_t s);
long comm_stat(int which);                        // smr_comm.cpp: 0 = all-reduces issued, 1 = of which in place
void count_launch();                              // every kernel launch the library issues (through HIP or directly): option "launches"
int eager_fence_all();                            // SMR_OK, or SMR_EHIP: a device's direct path failed (reported once)
void eager_note_hip_work(hipStream_t s);          // the library queued HIP work on owned stream s: s's next direct launch drains it first
void eager_forget_stream(hipStream_t s);
void eager_request_sys_acquire(hipStream_t s);
long eager_stat(int which);
bool eager_available(hipStream_t s);              // (of the stream's device)
int eager_fence_if_active();
bool eager_recent_writes_fit(uintptr_t dest_lo, uintptr_t dest_hi);  // smr_seq.cpp: the eager path's "are the recent destinations cache-resident" estimate
void mark_sliceable(int kind, unsigned off, unsigned row);  // applies to the NEXT recorded launch of the calling thread (no-op when nothing records)
void mark_self_released();                                  // likewise: RecLaunch::self_released
void take_slice_mark(RecLaunch& r);
void set_recorder(std::vector<RecLaunch>* r, bool allow_self_release = false, bool eager = false);
// Is the launch being recorded for a SEQUENCE (smr_seq) whose packets may drop their release fence, and is this execution in the
// regime where the fence matters (what it writes fits the caches: option "self_release_max_bytes", default 64 MiB)?  Launchers that
// can issue write-through stores then do (store policy 2) and call mark_self_released().
bool want_self_release(const Plan& plan);

// launchers (one per kernel TU)
int launch_generic_map(const Plan& plan, void* const* bases, hipStream_t s);
int launch_stream_map(const Plan& plan, void* const* bases, hipStream_t s);
int launch_tiled_map(const Plan& plan, void* const* bases, hipStream_t s);
int launch_reduce_all(const Plan& plan, void* const* bases, hipStream_t s);
int launch_reduce_part(const Plan& plan, void* const* bases, hipStream_t s);
int launch_orbit_map(const Plan& plan, void* const* bases, hipStream_t s);
int launch_flat_map(const Plan& plan, void* const* bases, hipStream_t s);

#endif  // !SMR_JIT

}  // namespace smr
