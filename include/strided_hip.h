/*
 * strided_hip.h -- C ABI of libstrided_hip.so: the MI355X (gfx950) implementation of
 * Strided.jl's fused N-ary strided map / map-reduce engine.
 *
 * Drop-in boundary.  The reference has no FFI; its single internal
 * funnel is
 *     _mapreduce_fuse!(f, op, initop, dims::Dims, arrays::Tuple{Vararg{StridedView}})
 *                                                   (reference src/mapreduce.jl:98-117)
 * reached from map! (src/mapreduce.jl:50), broadcast copyto! (src/broadcast.jl:35) and
 * _mapreducedim! (src/mapreduce.jl:93).  A Julia method specialised on a device-backed
 * parent array serialises its arguments into an `smr_problem` and `ccall`s `smr_mapreduce`
 * (binding shown in INTEGRATION.md / julia/StridedHIP.jl).  Plain C: pointers, sizes and
 * PODs only, no exceptions, no C++ or torch types.
 *
 * Conventions
 *  - Column-major, strides and offsets in ELEMENTS, 0-based:
 *        element (i_1..i_N), 0 <= i_k < dims[k], of operand j lives at
 *        ((T*)ops[j].base)[ ops[j].offset + sum_k i_k * ops[j].strides[k] ]
 *    (the reference's 1-based ParentIndex = offset + 1 + sum (i_k-1)*stride_k,
 *     src/mapreduce.jl:268).  Strides may be 0 (broadcast / reduced dim,
 *    src/broadcast.jl:50-65) or negative (reversed ranges).
 *  - ops[0] is the destination (arrays[1] of the reference), ops[1..M-1] are the inputs.
 *  - redop == SMR_RED_NONE : pure map, destination overwritten with f(inputs...)
 *    (kernel body src/mapreduce.jl:310-312).
 *    redop != NONE : dest[I] = op(dest[I], f(inputs...)) accumulated over every index that
 *    maps to the same destination element (dims with destination stride 0),
 *    src/mapreduce.jl:313-316.  The existing destination content takes part, after
 *    `initop` (if any) was applied to it exactly once (src/mapreduce.jl:351-382,403-409).
 *  - `conj` on an input: the loaded value is conjugated; on the destination: loads and
 *    stores conjugate (StridedView.op in {identity, conj}, ParentIndex get/set,
 *    src/mapreduce.jl:276-278, src/linalg.jl:50).
 *  - No dimension may be 0 (callers return early: src/mapreduce.jl:48,88-91).
 *  - All entry points return 0 (SMR_OK) or a negative smr_status; a human-readable
 *    message for the calling thread's last failure is available from smr_last_error().
 *    The reference throws DimensionMismatch etc. on the Julia side before the funnel
 *    (src/mapreduce.jl:43-46, src/broadcast.jl:61); the engine itself never throws.
 *  - Device pointers must belong to the current HIP device of the calling thread.
 *    Calls are asynchronous on `stream` (NULL = the null stream); the caller keeps the
 *    buffers alive until the stream reaches the kernels.
 */
#ifndef STRIDED_HIP_H
#define STRIDED_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SMR_ABI_VERSION 1
#define SMR_MAXN 8      /* max rank of the iteration box (reference tests go to N = 6)   */
#define SMR_MAXM 8      /* max operands including the destination                        */
#define SMR_MAXPROG 96  /* max instructions in an f-program                              */
#define SMR_MAXCONST 16 /* max (complex) constants captured by an f-program              */

typedef enum {
    SMR_OK = 0,
    SMR_EINVAL = -1,       /* malformed problem (rank, dims, program, null pointers)      */
    SMR_EUNSUPPORTED = -2, /* valid in the reference but outside the device whitelist ->
                              the host shim falls back to the CPU method                  */
    SMR_EHIP = -3,         /* a HIP runtime call failed                                   */
    SMR_ENOMEM = -4,
    SMR_ENODEVICE = -5     /* no usable gfx950 device                                     */
} smr_status;

/* Element types.  Arithmetic is done in one of the four float classes every reference
 * test-set iterates over (test/othertests.jl:2), or -- round 3 -- in the INTEGER class: when every
 * operand (destination included) has an integer eltype and f uses only integer-closed operations
 * (+ - * neg abs abs2 min max comparisons select, integer-valued constants; reductions + * min max & |),
 * the call computes in wrapping 64-bit two's-complement arithmetic like Julia's Int64 (sum of an Int64
 * view above 2^53, overflow wrap-around) and truncates on store to a narrower destination type (= the
 * wrapped result of Julia's arithmetic in that type for + - * chains); where Julia would observe a narrow
 * intermediate result at its own width (an order / equality test on it, a conversion to a wider type, a wider
 * destination) the library re-wraps the 64-bit value to that type (SMR_OP_WRAP_*, inserted by the planner),
 * so Int32 a .* b into an Int64 array is the wrapped 32-bit product.  UInt64 operands take part in ring
 * operations only (no order: min / max / < / abs are refused with SMR_EUNSUPPORTED, as is any 64-bit integer
 * input that meets floating-point arithmetic).  Pure moves (copy!/permutedims!) of any integer width are
 * bit copies; integers mixed with floats compute in Float64.                                             */
typedef enum {
    SMR_F32 = 0,
    SMR_F64 = 1,
    SMR_C32 = 2, /* ComplexF32: interleaved (re, im) float  */
    SMR_C64 = 3, /* ComplexF64: interleaved (re, im) double */
    SMR_I8 = 4,
    SMR_I16 = 5,
    SMR_I32 = 6,
    SMR_I64 = 7,
    SMR_U8 = 8,
    SMR_U16 = 9,
    SMR_U32 = 10,
    SMR_U64 = 11,
    SMR_BOOL = 12, /* Julia's Bool / NumPy's bool_: one byte holding 0 or 1.  Moves and computes like a UInt8; what differs is Julia's
                      typing of f (Bool yields to every integer type: true + Int8(1) is an Int8, -true and true + true are Ints) */
    SMR_DTYPE_COUNT = 13
} smr_dtype;

/* Reduction operator `op` (neutral elements follow _init_reduction!,
 * src/mapreduce.jl:182-191; the caller pre-fills the destination).                      */
typedef enum {
    SMR_RED_NONE = 0,
    SMR_RED_ADD = 1, /* +, Base.add_sum */
    SMR_RED_MUL = 2, /* *, Base.mul_prod */
    SMR_RED_MIN = 3,
    SMR_RED_MAX = 4,
    SMR_RED_AND = 5, /* &: both operands non-zero -> 1, else 0 (neutral element true, :188)  */
    SMR_RED_OR = 6   /* |  (neutral element false, :189)                                     */
} smr_redop;

/* `initop`, applied once to each destination element before accumulation.  These are
 * exactly the five forms the reference exercises (test/othertests.jl:76-102,
 * src/linalg.jl:146-160).                                                               */
typedef enum {
    SMR_INIT_NONE = 0,     /* nothing            */
    SMR_INIT_IDENTITY = 1, /* identity           */
    SMR_INIT_ZERO = 2,     /* zero / x -> 0      */
    SMR_INIT_SCALE = 3,    /* x -> x * beta      (beta = initarg[0] + i*initarg[1])      */
    SMR_INIT_CONST = 4,    /* x -> beta                                                   */
    SMR_INIT_CONJ = 5      /* conj                                                        */
} smr_initop;

/* f-program: the fused N-ary elementwise function f (the CaptureArgs functor tree of
 * src/broadcast.jl:67-98, or map!'s closure) serialised as postfix code, two bytes per
 * instruction: {opcode, immediate}.  The program must leave exactly one value.          */
typedef enum {
    SMR_OP_ARG = 0,   /* push input #imm (1-based: ops[imm]), conj flag applied           */
    SMR_OP_CONST = 1, /* push fconsts[2*imm] + i*fconsts[2*imm+1]                         */
    /* unary: pop 1, push 1 */
    SMR_OP_NEG = 8,
    SMR_OP_ABS = 9,
    SMR_OP_ABS2 = 10,
    SMR_OP_CONJ = 11,
    SMR_OP_REAL = 12,
    SMR_OP_IMAG = 13,
    SMR_OP_SQRT = 14,
    SMR_OP_EXP = 15,
    SMR_OP_LOG = 16,
    SMR_OP_SIN = 17,
    SMR_OP_COS = 18,
    SMR_OP_TANH = 19,
    SMR_OP_INV = 20,
    SMR_OP_ROUND32 = 21, /* round to Float32 (each part of a complex value): the host inserts it after
                            every operation Julia would have carried out in Float32 / ComplexF32 while
                            the call as a whole computes in Float64 (per-operation typing)         */
    SMR_OP_WIDEN = 22,   /* identity; its presence makes the call compute in the 64-bit class although no
                            operand is 64-bit: a Float64 / ComplexF64 scalar meets Float32 arrays
                            (`B32 .= A32 .* 0.1` multiplies in Float64 in Julia and rounds once on store) */
    /* 23..28 are the library's own: a user program holding one is malformed (SMR_EINVAL).  The integer class inserts them where
       Julia would have observed a narrow intermediate result at its own width (Int32 a .* b into an Int64 destination, UInt8
       min(a - b, c), ...): the 64-bit value is reduced to its low 8 / 16 / 32 bits, sign- or zero-extended (csrc/smr_plan.cpp) */
    SMR_OP_WRAP_I8 = 23,
    SMR_OP_WRAP_I16 = 24,
    SMR_OP_WRAP_I32 = 25,
    SMR_OP_WRAP_U8 = 26,
    SMR_OP_WRAP_U16 = 27,
    SMR_OP_WRAP_U32 = 28,
    /* binary: pop b, pop a, push a (op) b */
    SMR_OP_ADD = 32,
    SMR_OP_SUB = 33,
    SMR_OP_MUL = 34,
    SMR_OP_DIV = 35,
    SMR_OP_MIN = 36,
    SMR_OP_MAX = 37,
    SMR_OP_LT = 38, /* comparisons act on real parts and push 1.0 / 0.0                   */
    SMR_OP_LE = 39,
    SMR_OP_GT = 40,
    SMR_OP_GE = 41,
    SMR_OP_EQ = 42,
    SMR_OP_NE = 43,
    /* ternary: pop c, pop b, pop a, push (real(a) != 0 ? b : c)                           */
    SMR_OP_SELECT = 64
} smr_opcode;

/* One StridedView operand: (parent, size, strides, offset, op) of the reference's
 * StridedView (5-argument constructor call src/broadcast.jl:64) minus the shared size.  */
typedef struct smr_operand {
    void* base;                /* device pointer to parent[0]                             */
    int64_t offset;            /* element offset of the view's first element              */
    int64_t strides[SMR_MAXN]; /* element strides; 0 = broadcast/reduced; may be < 0      */
    int32_t dtype;             /* smr_dtype                                               */
    int32_t conj;              /* 0 = identity, 1 = conj                                  */
} smr_operand;

/* The fully lowered argument list of _mapreduce_fuse! (src/mapreduce.jl:98-99). */
typedef struct smr_problem {
    int32_t N;              /* rank of the iteration box, 1..SMR_MAXN                     */
    int32_t M;              /* operands incl. destination, 2..SMR_MAXM (1 allowed when
                               the program has no ARG, e.g. fill)                         */
    int64_t dims[SMR_MAXN]; /* common size of all operands, every entry >= 1             */
    smr_operand ops[SMR_MAXM];
    const uint8_t* fprog; /* 2*fprog_len bytes; NULL/0 = identity on ops[1]             */
    int32_t fprog_len;
    int32_t nconsts;
    const double* fconsts; /* 2*nconsts doubles (re, im)                                 */
    int32_t redop;         /* smr_redop                                                  */
    int32_t initop;        /* smr_initop, only with redop != NONE                        */
    double initarg[2];     /* beta for SCALE / CONST                                     */
    void* stream;          /* hipStream_t, NULL = null stream                            */
} smr_problem;

typedef struct smr_plan smr_plan; /* opaque: canonicalised problem + chosen kernel */

/* ---- library / device ---------------------------------------------------------------- */
int smr_abi_version(void);
/* Bind the calling thread to `device` (>= 0) and create the per-device state.  Replaces
 * the reference's thread-count configuration (src/Strided.jl:18-35,50-52).             */
int smr_init(int device);
int smr_shutdown(void);
int smr_device_count(void);
const char* smr_last_error(void);
/* Device memory for hosts without their own allocator (the Julia shim's HipBuffer).     */
int smr_malloc(size_t bytes, void** out);
int smr_free(void* p);
int smr_memcpy_h2d(void* dst, const void* src, size_t bytes, void* stream);
int smr_memcpy_d2h(void* dst, const void* src, size_t bytes, void* stream);
int smr_stream_sync(void* stream);

/* ---- overlap windows: independent launches of one stream may run concurrently -----------------
 * The reference runs the independent halves of a problem as concurrent tasks and waits only where it
 * must (src/mapreduce.jl:203-223).  The GPU analogue: between smr_overlap_begin(stream) and
 * smr_overlap_end(stream) the library tracks the byte ranges its launches on `stream` read and write;
 * a launch that conflicts with none of the launches still in flight (no read-after-write, write-after-
 * write, write-after-read on any operand or on a plan's own partials) is dispatched without the AQL
 * barrier bit, so its waves start while its predecessors are still running or draining (a kernel
 * boundary costs 1.6-1.9 us on MI355X: more than a third of a 16 MiB launch); every other launch is
 * stream-ordered as usual.  Results are those of in-order execution.  The FIRST launch after begin is
 * always ordered (after whatever the caller queued before); end issues one ordered empty kernel when
 * needed, so that work the caller queues afterwards -- kernels, copies, events, synchronisation -- is
 * ordered after everything in the window.  Inside an open window the caller must not put work of its
 * own on `stream` (or must call smr_overlap_fence(stream) first).  Windows nest; they survive stream
 * capture into a hipGraph.
 * ON gfx942 / gfx950 THE WINDOW IS A NO-OP FOR LAUNCHES THAT GO THROUGH HIP: HIP accepts hipExtAnyOrderLaunch and ignores it on gfx9
 * (measured with device stamps, profiles/r04_overlap.txt), so the library does not even run the analysis there unless option
 * "overlap_window_hip" = 1 asks for it.  Independent launches DO overlap where the library dispatches by itself: on the streams of
 * smr_stream_create (next paragraph) and in recorded sequences (smr_seq_*).
 * smr_stream_create returns a stream of the library's own -- the stream of a host (the Julia shim) that routes ALL its device work
 * through this library.  On MI355X its launches do not go through HIP at all: the library submits every launch itself as an AQL
 * packet on one of four HSA queues it owns, choosing the queue by the data -- a launch that conflicts with nothing in flight goes to
 * the least loaded queue and runs concurrently with its predecessors, one that conflicts with launches on one queue follows them
 * there, one that conflicts with several queues waits for them through a barrier-AND packet; completion signals retire the ranges
 * (csrc/smr_seq.cpp: eager direct dispatch; ~1 us of host time per launch instead of HIP's 3.6-4 us).  Results are those of in-order
 * execution.  The library fences by itself (waits for its queues) before every copy / synchronisation / sequence replay it performs on
 * such a stream (smr_memcpy_*, smr_stream_sync, smr_mapreduce_scalar, smr_seq_run, smr_free, smr_plan_destroy) and drains the HIP
 * work it queued there before the next direct launch.  Option "eager_direct" = 0 sends the launches through HIP instead (the window
 * of such a stream is then permanently open).                                                                                      */
int smr_overlap_begin(void* stream);
int smr_overlap_end(void* stream);
int smr_overlap_fence(void* stream);
int smr_stream_create(void** out);
int smr_stream_destroy(void* stream);

/* ---- recorded sequences: the library's own replay of a list of plan executions ---------------------
 * smr_seq_add records executions of plans (with optional rebinding of base pointers, as smr_plan_execute);
 * smr_seq_run(seq, reps, stream) performs the recorded list `reps` times with the results of in-order
 * execution on `stream`.  On MI355X the replay does not go through HIP's launch path: every launch
 * becomes a pre-built AQL dispatch packet (kernel object of the code object HIP loaded, kernarg block
 * resident in device memory) on HSA queues the library owns.  The recorded executions are split into
 * dependency components -- two executions belong together when one writes bytes the other reads or
 * writes (operand footprints and the plans' own partials) -- and every component keeps its recorded
 * order on ONE queue, while different components run on different queues (up to 4), concurrently: the
 * device form of src/mapreduce.jl:203-223 (spawn what is independent, wait where it must).  Inside one
 * queue MI355X runs consecutive dispatches one after the other whatever the barrier bit says, and a
 * kernel boundary costs 1.6-1.9 us there (profiles/r04_overlap.txt); two queues overlap one chain's
 * boundary with the other chain's kernel.  A replay costs ~0.1 us of host time per launch instead of
 * HIP's 3.6-4 us.  Work queued on `stream` before the call completes first (the call waits for it on
 * the host when the stream is busy).  smr_seq_run is ASYNCHRONOUS (round 5): it returns when the doorbells are rung -- like the
 * reference's @spawn, whose `wait` is a separate step (src/mapreduce.jl:214-223) -- and work queued on `stream` afterwards still
 * comes after the replay:
 *   - a library-owned stream (smr_stream_create): everything on it goes through this library, which waits for the replay before
 *     whatever it submits or does there next;
 *   - a HIP stream: hipStreamWaitValue64 on the completion signals where the device offers it ($SMR_SEQ_STREAM_WAIT=1), otherwise a
 *     one-wave holding kernel on the stream that polls the signals (bounded by $SMR_DIRECT_TIMEOUT_MS; smr_seq_wait reports a
 *     hold that gave up).
 * smr_seq_wait blocks the host until the last replay has completed (active wait on the signals: microseconds sooner than
 * hipStreamSynchronize).  smr_seq_set "async" = 0 makes smr_seq_run itself wait.  Runtime-compiled kernels take part like
 * precompiled ones (their entry points carry a per-program name; the sequence co-owns the loaded program).  A sequence holding a
 * kernel that needs scratch memory is replayed through HIP, in order (smr_seq_info tells which, and reports "last_replay_us":
 * first doorbell -> completion observed; "kernarg_layout": how the hidden-argument offsets of the packets are known -- "metadata"
 * = read from the code objects' NT_AMDGPU_METADATA notes, "+verified" = a self-test packet saw the blockDim / gridDim / LDS it was
 * given; csrc/smr_kmeta.cpp).
 * Failure: an HSA queue error, or a completion signal that does not arrive within $SMR_DIRECT_TIMEOUT_MS (default 30 s), marks the
 * device's direct path as failed -- the call that notices returns SMR_EHIP, nothing waits on those queues again, and later
 * executions and replays go through HIP.
 * One replay per device is in flight at a time.  The recorded base pointers / plans must stay alive
 * while the sequence exists.                                                                         */
typedef struct smr_seq smr_seq;
int smr_seq_create(smr_seq** out);
int smr_seq_add(smr_seq* seq, smr_plan* plan, void* const* bases);
int smr_seq_run(smr_seq* seq, int reps, void* stream);
int smr_seq_wait(smr_seq* seq);
int smr_seq_info(smr_seq* seq, char* buf, size_t buflen);
/* The dependency analysis alone (host arithmetic on strides and base pointers, no device needed): comp[i] = dependency component
 * of recorded execution i (numbered in order of first appearance); returns the number of components or a negative status.      */
int smr_seq_components(smr_seq* seq, int32_t* comp, size_t cap);
/* The fence analysis alone (host arithmetic too): acquire[i] = 1 when execution i reads bytes some execution of the sequence writes
 * (only those packets acquire inside a replay), *footprint = bytes of the union of every range the sequence touches,
 * *cache_resident = 1 when that is within option "self_release_max_total" (launches may then be self-released).  Returns the number
 * of recorded executions or a negative status.                                                                                       */
int smr_seq_fences(smr_seq* seq, int32_t* acquire, size_t cap, int64_t* footprint, int32_t* cache_resident);
/* "queues" (1..8, default 4: hardware queues a replay may spread over; 1 = everything in recorded order on one queue; more than 4
 * are time-multiplexed by the hardware scheduler); "slices" (-1 | 1..8: a component that consists of ONE launch of independent
 * workgroups is cut into that many contiguous block ranges, one queue each -- the device form of _mapreduce_threaded!'s bisection;
 * -1, the default: only the heaviest such component is cut, in two, when a third queue is free); "slices:<c>" (block ranges of
 * component c alone); "async" (-1 default / 0 blocking smr_seq_run / 1).
 * Fences of the packets inside a replay: "acquire" (-1 default: agent scope only on packets that read bytes the sequence writes --
 * an acquire invalidates the L2s, i.e. the read-only inputs of everything in flight; 0 none, 1 agent, 2 system on every packet),
 * "release" (1 agent, default; 2 system; 0 none: an EXPERIMENT, write-after-write across XCDs needs the write-back).
 * Experiments: "fence_scope" (both at once), "first_acquire" / "last_release" (scope of a replay's first / last packet, default 2),
 * "order" (0: every packet carries the barrier bit, 1: only those that conflict with an earlier one in flight) */
int smr_seq_set(smr_seq* seq, const char* name, int64_t value);
int smr_seq_destroy(smr_seq* seq);
/* Diagnostics (no device needed): the kernarg layout csrc/smr_kmeta.cpp reads from an AMDGPU code object image.  out[0..19] =
 * kernarg_size, explicit_end, nargs_explicit, hidden block_count x/y/z, group_size x/y/z, remainder x/y/z, global_offset x/y/z,
 * grid_dims, dynamic_lds_size offsets (-1 = not declared), needs_runtime, private_size, group_static.  `symbol` = "<mangled>.kd", or
 * NULL with index >= 0 to enumerate (the symbol is copied to name_out).  Returns the number of kernels, negative on a parse error. */
int smr_debug_kernarg_layout(const void* elf, size_t bytes, const char* symbol, int index, int32_t* out, char* name_out, size_t name_cap);
/* Test hook, host-only: the f-program the kernels of this problem would run -- the caller's program after canonicalisation, i.e.
   with the SMR_OP_WRAP_* instructions of the integer class in place.  code receives min(2 * length, cap) bytes (opcode, immediate
   pairs; SMR_OP_ARG immediates number the canonical operands: orig[j], SMR_MAXM entries, is the caller's index of operand j, -1 past
   the end -- unused and repeated operands are dropped); *nwraps the number of added instructions, *compute_class the SMR_* dtype
   computed in (-1: a bit copy).  Output pointers may be null.  Returns the length in instructions, or the (negative) error code. */
int smr_debug_canon_prog(const smr_problem* problem, uint8_t* code, int cap, int* nwraps, int* compute_class, int32_t* orig);


/* ---- the hot path --------------------------------------------------------------------- */
/* One-shot replacement of _mapreduce_fuse! (src/mapreduce.jl:98): canonicalise, pick the
 * kernel family, launch on problem->stream.  Plans are cached per problem signature.    */
int smr_mapreduce(const smr_problem* problem);
/* Complete reduction returning its value to the host: runs `problem` (whose destination is
 * ONE device element: every destination stride 0 or dim 1), waits for problem->stream and
 * copies the destination element (its own dtype) to host_result.  The synchronous tail of
 * `_mapreduce` (src/mapreduce.jl:70-71: `return out[ParentIndex(1)]`).                    */
int smr_mapreduce_scalar(const smr_problem* problem, void* host_result);

/* Planned form (the planner is the analogue of _mapreduce_order!/_mapreduce_block!/
 * _computeblocks, src/mapreduce.jl:119-180,452-500, evaluated once and reused).          */
int smr_plan_create(const smr_problem* problem, smr_plan** out);
/* `bases` (optional, M entries) rebinds the operand base pointers; NULL keeps the ones the
 * plan was created with.  `stream` overrides problem->stream.                            */
int smr_plan_execute(smr_plan* plan, void* const* bases, void* stream);
int smr_plan_destroy(smr_plan* plan);
/* Builds everything the first execution would build -- index tables uploaded, the kernel for a
 * runtime-compiled f compiled and loaded, reduction scratch allocated -- without launching.
 * Call it before capturing smr_plan_execute into a hipGraph (uploads are synchronous copies).  */
int smr_plan_prepare(smr_plan* plan);
/* Writes a one-line description ("family=tiled tile=32x32 grid=1024 ...") into buf.      */
int smr_plan_describe(const smr_plan* plan, char* buf, size_t buflen);
/* Algorithmic bytes of one execution: every distinct operand footprint counted once
 * (SURVEY.md section 8d).                                                                 */
int64_t smr_plan_algorithmic_bytes(const smr_plan* plan);
/* Introspection of the TILED family's execution order (no reference counterpart: the
 * reference walks its blocks in loop order, src/mapreduce.jl:385-401).  Returns the number
 * of workgroups n of the launch, or 0 when tiles run in natural order; writes
 * min(n, cap) entries: out[b] = linear tile id executed by workgroup b, 0xffffffff = idle
 * padding.  Workgroup b runs on XCD b mod 8.
 * ORBIT family: a workgroup holds one tile per LDS slot (as many slots as the views' permutation group has elements, four for a
 * group of order three); the list has `slots` entries per workgroup (n = slots * grid, `grid` as smr_plan_describe prints it):
 * out[b * slots + g] = tile in slot g of workgroup b, the first of an idle workgroup is 0xffffffff.  Every tile of the box
 * appears exactly once, except that a group of order three repeats slot 0 in slot 3 and an incomplete last workgroup of
 * shared (diagonal) orbits repeats its slot 0.                                             */
int64_t smr_plan_tile_order(const smr_plan* plan, uint32_t* out, size_t cap);
/* ORBIT family, PAIR form (4^4 cubes of 8-byte elements, a group of order four; `pair_grid` in smr_plan_describe): the work list
 * the PAIR kernel walks -- EIGHT tiles per workgroup, out[w * 8 + b * 4 + g] = tile in slot g of the workgroup's slot set b
 * (the first of an idle workgroup is 0xffffffff).  Every tile of the box appears exactly once (an odd set out repeats set 0);
 * where the planner found it, set 1 is the orbit of set 0's unit-axis neighbour (tile + 1).  Returns the number of entries,
 * 0 when the plan has no PAIR form.  No reference counterpart.                                                            */
int64_t smr_plan_orbit_pairs(const smr_plan* plan, uint32_t* out, size_t cap);
/* Introspection of the two-sided FLAT form (no reference counterpart): the two memory runs a tile is the product of, for the
 * planner tests (tests/test_abi.py walks every tile with these numbers on the CPU and checks that each element of the box is
 * moved exactly once, from the right place to the right place).  Returns the number of values the description has -- 0 when the
 * plan is of another kind -- and writes min(that, cap) of them:
 *   [0] kt (operand index of the transposed input)   [1] shared   [2] N (canonical rank)
 *   [3..4] R[0], R[1]   [5..6] TP[0], TP[1]   [7..8] p[0], p[1] (-1: none)
 *   then N dims, N destination strides, N strides of operand kt, N flags ingroup[0], N flags ingroup[1],
 *   R[0] offsets roff[0][r], R[1] offsets roff[1][r]      (side 0 = destination, 1 = operand kt; canonical dim order) */
int64_t smr_plan_flat_runs(const smr_plan* plan, int64_t* out, size_t cap);
/* The same for the ONE-sided FLAT forms (plain, fused, shared-lead); 0 for any other plan.  Values:
 *   [0] dir (0: the destination is the flat side, 1: operand kt is)   [1] R   [2] tplog   [3] tqlog   [4] p   [5] q
 *   [6] lshare   [7] fuse   [8] kt   [9] N   then N dims, N destination strides, N strides of operand kt, N flags ingroup,
 *   R offsets roff[r] (line-side offset of leading index r of the flat run)                                                */
int64_t smr_plan_flat_side(const smr_plan* plan, int64_t* out, size_t cap);
/* The same for the BATCHED form (round 4: contiguous blocks on both sides, csrc/smr_k_flat.hip: flatb_body); 0 for any other plan:
 *   [0] g (dims of a block)   [1] P (elements of a block)   [2] K (blocks per workgroup)   [3] N
 *   then N dims, N destination strides, N input strides, P offsets srcoff[r] (input offset inside the block of destination position r) */
int64_t smr_plan_flat_batched(const smr_plan* plan, int64_t* out, size_t cap);

/* Runtime compilation.  An `f` without a natively compiled functor is specialised the way
 * Julia specialises the reference's @generated kernel per closure (src/mapreduce.jl:229-425):
 * the f-program is turned into a C++ functor and the plan's kernel is compiled for it with
 * hiprtc on first execution (cached per process; option "jit" = 0 keeps the bytecode
 * interpreter, which is also the fallback when hiprtc is unavailable).
 * smr_plan_jit_compile: generate + compile now, without a device (warms the compiler cache;
 * *code_size = bytes of the code object, 0 when the plan's kernel needs no compilation).
 * smr_plan_jit_source: the generated functor as text.                                     */
int smr_plan_jit_compile(smr_plan* plan, size_t* code_size);
int smr_plan_jit_source(const smr_plan* plan, char* buf, size_t buflen);

/* Block-partition a problem over `nshards` devices/ranks exactly like the reference's
 * task bisection splits the iteration box (src/mapreduce.jl:203-222): sub-box `shard`
 * gets its dims and per-operand offsets in *out.  Pure host arithmetic.  For reductions
 * whose split dim is a reduced one, *needs_allreduce is set: the caller combines the
 * per-shard partial destinations with RCCL (ncclAllReduce, op = redop).                  */
int smr_shard(const smr_problem* problem, int nshards, int shard, smr_problem* out,
              int* needs_allreduce);
/* The same with block-partitioned operands: bit k of `local_ops` says that ops[k].base/offset
 * already address THIS shard's slab (box index `start` of the split dim is the operand's index 0
 * along it; strides unchanged), so that operand's offset is not shifted and nothing has to be
 * replicated across devices.  *split_dim / *start / *stop (optional) report the slab
 * [start, stop) of box dim split_dim this shard owns (-1 when nshards == 1).               */
int smr_shard_ex(const smr_problem* problem, int nshards, int shard, uint32_t local_ops,
                 smr_problem* out, int* needs_allreduce, int* split_dim, int64_t* start,
                 int64_t* stop);
/* Writes the neutral element of problem->redop (0, 1, +inf, -inf, true, false) into every
 * distinct destination element: what _init_reduction! does for the per-task partial slots
 * (src/mapreduce.jl:182-191,157-163).  Asynchronous on problem->stream.                    */
int smr_init_reduction(const smr_problem* problem);

/* ---- multi-GPU from plain C: one process per GPU, RCCL over xGMI --------------------------
 * The C twin of what strided.jl_amd/distributed.py does through torch.distributed, for hosts
 * without it (the Julia shim).  Rank 0 calls smr_comm_unique_id and ships the 128 bytes to
 * the other ranks out of band (MPI, a socket, a file); every rank -- with its device already
 * selected (smr_init) -- calls smr_comm_init.  smr_mapreduce_sharded then executes ONE logical
 * problem cooperatively: every rank passes the same problem over its own device copies of
 * the operands; smr_shard picks this rank's sub-box; a split reduced dim is completed by one
 * ncclAllReduce of the destination elements (initop / old destination content enter once,
 * on rank 0: the distributed form of src/mapreduce.jl:153-170).  Map: on return each rank's
 * destination holds its own slab.  Reduce: every rank's destination holds the full result.
 * With nranks == 1 everything degenerates to smr_mapreduce and RCCL is never loaded.       */
int smr_comm_unique_id(void* out, size_t len);
int smr_comm_init(int nranks, int rank, const void* unique_id, size_t len);
/* (rank, nranks) as the COMMUNICATOR reports them (ncclCommUserRank / ncclCommCount); (0, 1) without one. */
int smr_comm_rank(int* rank, int* nranks);
/* Path of the RCCL library in use.  Choice: $SMR_RCCL_LIB when set; else a librccl the process has already
 * loaded (a host that ships its own, e.g. torch/lib/librccl.so: never two copies in one process); else the
 * system's librccl.so.1.                                                                                   */
int smr_comm_library(char* buf, size_t buflen);
int smr_comm_destroy(void);
int smr_mapreduce_sharded(const smr_problem* problem);
/* ... with block-partitioned operands (see smr_shard_ex).                                   */
int smr_mapreduce_sharded_ex(const smr_problem* problem, uint32_t local_ops);

/* Tuning knobs; smr_set_option returns SMR_EINVAL for unknown names.  Analogue of the
 * reference's compile-time constants MINTHREADLENGTH / BLOCKMEMORYSIZE
 * (src/mapreduce.jl:141,462).  Names: "force_family" (0 auto, 1 generic, 3 tiled),
 * "tile_log2" (0 auto, 10, 12), "tile_lg0".."tile_lg7" (per-dim log2 tile extent, -1 auto),
 * "tiled_vec", "max_lds_bytes", "tile_order" (orbit-major tile order on/off),
 * "tiled_persist" / "tiled_persist_wpc" / "tiled_persist_min" (persistent pipelined form),
 * "reduce_blocks", "reduce_part_kind" (-1 auto, 0 general, 1 row, 2 col), "reduce_col_txlog",
 * "reduce_part_wgs" (partial reductions with fewer workgroups are split until about this many run), "reduce_col_narrow",
 * "reduce_single" (split reductions of at most this many chunks fold their partials inside the same launch; 0 = always a
 * second launch; a plan that owns partials must not run concurrently with itself on two streams), "reduce_tree" (up to this many
 * chunks fold inside the launch through two levels of arrival counters; 0 = off),
 * "jit" (runtime compilation of f on/off), "orbit" (ORBIT family on/off),
 * "orbit_lg" / "orbit_min" / "orbit_few" (orbit tile edge and thresholds), "orbit_pipe" (persistent
 * pipelined orbits: -1 auto, 0, 1), "nt_store" (non-temporal stores: -1 auto, 0 never, 1 always),
 * "nt_stream_min", "nt_load" (non-temporal loads of complete reductions: -1 auto, 0, 1), "flat" (FLAT family for short
 * leading dims that are not powers of two, on/off), "flat2" (its two-sided form: 0 off, 1 planner's rule, 2 wherever it applies) /
 * "flat2_bytes" / "flat2_lead_bytes", "reduce_row_floor", "reduce_row_dense", "tile_block" (block tile order for distinct arrays with several unit
 * axes: -1 auto, 0 off, n), "tile_block_xcd", "orbit_group", "orbit_minrun", "orbit_wgs".  Experiment
 * switches: "stream_u", "stream_pack_rows", "flatb" (batched FLAT form on/off), "eager_direct", "orbit_lds_min", "orbit_skew", "tile_block_min_axes"; "stamp_base" / "stamp_cap" / "stamp_used" (debug build
 * with device-side wall-clock stamps, csrc/smr_device.h).  Read-only counters through
 * smr_get_option: "jit_compiles", "jit_hits", "jit_failures", "jit_compile_ms", "overlap_any" / "overlap_ordered" /
 * "overlap_fences" (launches dispatched without / with the barrier bit inside overlap windows, fences issued); "eager_launches",
 * "eager_free" / "eager_same" / "eager_cross" (direct launches; executions that conflicted with nothing / one queue / several),
 * "eager_fallback" (executions sent through HIP).
 * Round 5: "seq_self_release" (1: launches recorded for a sequence issue agent-scope write-through stores where the family can and
 * their packets carry no release fence), "eager_self_release" (1: the same on library-owned streams), "self_release_max_bytes"
 * (64 MiB: largest destination for which that is done), "self_release_max_total" (128 MiB: largest footprint of a whole sequence /
 * of the recently written destinations), "nt_store" = 2 (force write-through stores), "tiled_gorder" (-1: HBM-sized transposing
 * copies walk the tile index along the input's unit axis second; 0 canonical; 1 always), "tiled_xpose" (1: the lean kernel for
 * HBM-sized transposing copies), "stream_ua" (1: element-aligned 16-byte vectors + a partial vector per row in STREAM),
 * "allreduce_f64" (0; 1: Float32 / ComplexF32 sums cross the ranks as Float64), "overlap_window_hip" (0; see smr_overlap_begin),
 * "orbit_deal" (experiment).  Read-only: "launches" (kernel launches the library issued, through HIP or directly), "allreduces",
 * "allreduces_inplace".  Environment: $SMR_DIRECT_TIMEOUT_MS (30000: no-progress limit of waits on the direct queues),
 * $SMR_DIRECT_SELFTEST (0 skips, "fail" forces the failure path), $SMR_DIRECT_METADATA (0: code-object-v5 rule instead of the
 * metadata), $SMR_SEQ_DIRECT (0: sequences replay through HIP), $SMR_SEQ_STREAM_WAIT (1: hipStreamWaitValue64 where supported),
 * $SMR_RCCL_LIB (which collective library to dlopen).      */
int smr_set_option(const char* name, int64_t value);
int64_t smr_get_option(const char* name);

#ifdef __cplusplus
}
#endif
#endif /* STRIDED_HIP_H */
