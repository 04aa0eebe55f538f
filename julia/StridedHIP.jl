# StridedHIP.jl -- the reference-side binding a Strided.jl maintainer would add so that
# StridedViews whose parent lives in MI355X memory run on libstrided_hip.so.
#
# NOT EXECUTED IN THIS REPOSITORY: there is no Julia runtime in the build image or on the GPU
# box.  tests/test_julia_shim.py checks by parsing this file that its C structs, opcode / dtype /
# redop / initop tables equal include/strided_hip.h.  It is kept deliberately thin: every method only
# *serialises* what the reference's funnel already holds -- `_mapreduce_fuse!(f, op, initop, dims,
# arrays)`, src/mapreduce.jl:98 -- into the C struct of include/strided_hip.h and `ccall`s it.  All view
# algebra, `promoteshape`, argument checks, the `@strided` macro AND the BroadcastStyle rules stay the
# reference's own: a device view mixed with a plain host `Array` resolves to DefaultArrayStyle
# (src/broadcast.jl:11-18), i.e. Base's scalar-indexing broadcast, which `HipBuffer` refuses -- upload first.
module StridedHIP

using Strided, StridedViews
using Strided: CaptureArgs, Arg, StridedArrayStyle, capturestridedargs
using Base.Broadcast: Broadcasted, DefaultArrayStyle
import Strided: _mapreduce_fuse!

const lib = get(ENV, "STRIDED_HIP_LIB", "libstrided_hip.so")
const MAXN, MAXM = 8, 8

# Calls are ORDERED on the library's stream and return as soon as they are queued (round 4; rounds 1-3 ended every funnel call
# with a stream synchronisation, which costs more than a 3 us kernel).  That is indistinguishable from the reference's synchronous
# behaviour (tasks are `wait`ed, src/mapreduce.jl:214-223) for a Julia program: a HipBuffer cannot be indexed on the host, and every
# way of observing its content -- `download`, `copyto!(::Array, ::HipBuffer)`, the CPU fallback below -- synchronises first; `hipFree`
# in the finaliser drains the device too.  `StridedHIP.async!(false)` restores a synchronisation after every call (timing with
# `@time`, debugging); `StridedHIP.synchronize()` waits explicitly.
const ASYNC = Ref(true)
async!(on::Bool=true) = (ASYNC[] = on; nothing)
# The shim's stream belongs to the library (smr_stream_create): on MI355X the library submits its launches itself -- AQL packets on
# HSA queues it owns, the queue chosen by the operands' byte ranges, so that independent broadcasts overlap and a call costs ~1.8 us
# of host time instead of HIP's 3.8 -- and fences by itself before every copy / synchronisation below.
const STREAM = Ref{Ptr{Cvoid}}(C_NULL)
function stream()
    if STREAM[] == C_NULL
        r = Ref{Ptr{Cvoid}}(C_NULL)
        ccall((:smr_stream_create, lib), Cint, (Ptr{Ptr{Cvoid}},), r) == 0 && (STREAM[] = r[])   # on failure: the null stream, through HIP
    end
    return STREAM[]
end
synchronize() = check(ccall((:smr_stream_sync, lib), Cint, (Ptr{Cvoid},), stream()))

struct Unsupported <: Exception
    msg::String
end
function check(rc::Cint)
    rc == 0 && return nothing
    msg = unsafe_string(ccall((:smr_last_error, lib), Cstring, ()))
    rc == -2 && throw(Unsupported(msg))          # SMR_EUNSUPPORTED -> CPU fallback below
    rc == -1 && throw(ArgumentError(msg))        # SMR_EINVAL
    error("libstrided_hip: $msg (status $rc)")
end

# ---- device memory: a DenseArray whose storage is a HIP allocation (smr_malloc) ---------------
mutable struct HipBuffer{T,N} <: DenseArray{T,N}
    ptr::Ptr{T}
    dims::NTuple{N,Int}
    function HipBuffer{T}(::UndefInitializer, dims::NTuple{N,Int}) where {T,N}
        p = Ref{Ptr{Cvoid}}()
        check(ccall((:smr_malloc, lib), Cint, (Csize_t, Ptr{Ptr{Cvoid}}), prod(dims) * sizeof(T), p))
        b = new{T,N}(convert(Ptr{T}, p[]), dims)
        finalizer(x -> ccall((:smr_free, lib), Cint, (Ptr{Cvoid},), x.ptr), b)
        return b
    end
end
Base.size(b::HipBuffer) = b.dims
Base.strides(b::HipBuffer) = Base.size_to_strides(1, b.dims...)
Base.elsize(::Type{<:HipBuffer{T}}) where {T} = sizeof(T)
Base.pointer(b::HipBuffer) = b.ptr
Base.unsafe_convert(::Type{Ptr{T}}, b::HipBuffer{T}) where {T} = b.ptr
Base.similar(b::HipBuffer, ::Type{T}, dims::Dims) where {T} = HipBuffer{T}(undef, dims)
Base.getindex(::HipBuffer, I...) = error("scalar indexing of device memory: download(...) first")
Base.setindex!(::HipBuffer, v, I...) = error("scalar indexing of device memory: upload(...) instead")
function Base.copyto!(dst::HipBuffer{T,N}, src::Array{T,N}) where {T,N}   # host -> device
    check(ccall((:smr_memcpy_h2d, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Ptr{Cvoid}), dst.ptr, src, sizeof(src), stream()))
    check(ccall((:smr_stream_sync, lib), Cint, (Ptr{Cvoid},), stream()))
    return dst
end
function Base.copyto!(dst::Array{T,N}, src::HipBuffer{T,N}) where {T,N}   # device -> host
    check(ccall((:smr_memcpy_d2h, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Ptr{Cvoid}), dst, src.ptr, sizeof(dst), stream()))
    check(ccall((:smr_stream_sync, lib), Cint, (Ptr{Cvoid},), stream()))
    return dst
end
upload(a::Array{T,N}) where {T,N} = copyto!(HipBuffer{T}(undef, size(a)), a)
download(b::HipBuffer{T,N}) where {T,N} = copyto!(Array{T,N}(undef, size(b)), b)

const HipView = StridedView{<:Any,<:Any,<:HipBuffer}
# out-of-place `@strided A .+ B` on device views allocates its result on the device (the reference's method,
# src/broadcast.jl:20-22, makes a host Array); more specific than the reference's `<:StridedArrayStyle{N}`
function Base.similar(bc::Broadcasted{StridedArrayStyle{N}}, ::Type{T}) where {N,T}
    any(a -> a isa HipView, capturestridedargs(bc)) && return StridedView(HipBuffer{T}(undef, map(length, axes(bc))))
    return StridedView(similar(convert(Broadcasted{DefaultArrayStyle{N}}, bc), T))
end

# ---- C structs (include/strided_hip.h) -------------------------------------------------------------
struct SmrOperand
    base::Ptr{Cvoid}
    offset::Int64
    strides::NTuple{MAXN,Int64}
    dtype::Int32
    conj::Int32
end
struct SmrProblem
    N::Int32
    M::Int32
    dims::NTuple{MAXN,Int64}
    ops::NTuple{MAXM,SmrOperand}
    fprog::Ptr{UInt8}
    fprog_len::Int32
    nconsts::Int32
    fconsts::Ptr{Float64}
    redop::Int32
    initop::Int32
    initarg::NTuple{2,Float64}
    stream::Ptr{Cvoid}
end

const DTYPES = Dict(Float32 => 0, Float64 => 1, ComplexF32 => 2, ComplexF64 => 3, Int8 => 4, Int16 => 5, Int32 => 6, Int64 => 7,
                    UInt8 => 8, Bool => 12, UInt16 => 9, UInt32 => 10, UInt64 => 11)
dtypecode(T) = get(() -> throw(Unsupported("eltype $T")), DTYPES, T)

pad(t::NTuple{N,Int}, v) where {N} = ntuple(i -> i <= N ? Int64(t[i]) : Int64(v), MAXN)
operand(a::StridedView{T}) where {T} = SmrOperand(pointer(a.parent), a.offset, pad(a.strides, 0), dtypecode(T), a.op === conj ? 1 : 0)
const NULLOP = SmrOperand(C_NULL, 0, ntuple(_ -> Int64(0), MAXN), 0, 0)

# ---- f -> postfix f-program (walks the CaptureArgs tree of src/broadcast.jl:67-83) ---------------------
const UNARY = Dict((-) => 8, abs => 9, abs2 => 10, conj => 11, real => 12, imag => 13, sqrt => 14, exp => 15,
                   log => 16, sin => 17, cos => 18, tanh => 19, inv => 20)
const BINARY = Dict((+) => 32, (-) => 33, (*) => 34, (/) => 35, min => 36, max => 37, (<) => 38, (<=) => 39, (>) => 40,
                    (>=) => 41, (==) => 42, (!=) => 43)
const OP_ARG, OP_CONST, OP_ROUND32, OP_WIDEN, OP_SELECT = 0x00, 0x01, 0x15, 0x16, 0x40
const NARROW, WIDE = (Float32, ComplexF32), (Float64, ComplexF64)
mutable struct Prog
    code::Vector{UInt8}
    consts::Vector{Float64}
    nextarg::Int
    eltypes::Vector{DataType}   # of the inputs, in argument order
    wide::Bool                  # the library computes this call in a 64-bit class
end
# every emit! returns the Julia type of the value it pushed: Julia types each operation of a fused expression
# separately, the library computes one class per call -> ROUND32 after operations Julia carries out in 32 bits
function emit!(p::Prog, ::Arg)
    p.nextarg += 1
    push!(p.code, OP_ARG, UInt8(p.nextarg))
    return p.eltypes[p.nextarg]
end
struct ArgK                                    # argument k of a traced closure (explicit index; `Arg` counts occurrences)
    k::Int
end
emit!(p::Prog, a::ArgK) = (push!(p.code, OP_ARG, UInt8(a.k)); p.eltypes[a.k])
function emit!(p::Prog, x::Number)
    push!(p.consts, real(x), imag(x))
    push!(p.code, OP_CONST, UInt8(length(p.consts) ÷ 2 - 1))
    return typeof(x)
end
function round32!(p::Prog, T)
    p.wide && T in NARROW && push!(p.code, OP_ROUND32, 0x00)
    return T
end
function emit!(p::Prog, c::CaptureArgs)
    f, args = c.f, c.args
    if length(args) == 1 && haskey(UNARY, f)
        T = emit!(p, args[1])
        push!(p.code, UNARY[f], 0x00)
        return round32!(p, Base.promote_op(f, T))
    elseif haskey(BINARY, f) && (length(args) == 2 || f in (+, *, min, max))
        T = emit!(p, args[1])                  # Julia's +(a,b,c,d) folds left
        for a in args[2:end]
            T = Base.promote_op(f, T, emit!(p, a))
            push!(p.code, BINARY[f], 0x00)
            round32!(p, T)
        end
        return T
    elseif f === ifelse && length(args) == 3
        Ts = map(a -> emit!(p, a), args)
        push!(p.code, OP_SELECT, 0x00)
        return promote_type(Ts[2], Ts[3])
    end
    throw(Unsupported("function $f"))
end
emit!(p::Prog, x) = throw(Unsupported("captured $(typeof(x))"))
function fprogram(c::Union{CaptureArgs,ArgK,Number}, eltypes, desttype)
    p = Prog(UInt8[], Float64[], 0, collect(eltypes), false)
    T = emit!(p, c)                            # dry pass: does a 64-bit type occur anywhere?
    arrays64 = any(t -> t in WIDE || t <: Integer, (desttype, eltypes...))
    if arrays64 || T in WIDE
        p = Prog(UInt8[], Float64[], 0, collect(eltypes), true)
        emit!(p, c)
        arrays64 || push!(p.code, OP_WIDEN, 0x00)   # the 64-bit class comes from a scalar (`A32 .* 0.1`)
    end
    return p
end

# ---- plain closures: map!((x, y, z) -> sin(x) + y / exp(-abs(z)), ...) (test/othertests.jl:22-24), mapreduce(f, op, ...) ----------
# A closure is opaque to dispatch, so it is TRACED: called once on tracer values that record every operation of the UNARY /
# BINARY tables (and ifelse) as the same CaptureArgs tree a broadcast would have produced, with explicit argument indices.
# `Traced{T}` carries the Julia type T the value would have, so that Base.promote_op types the trace operation by operation
# exactly as it types a CaptureArgs tree.  Anything else the closure does with its arguments -- control flow on a comparison,
# an untabulated function, indexing -- throws while tracing and the call falls back to the CPU method (Unsupported).
struct Traced{T} <: Number
    node::Any                                  # ArgK | Number | CaptureArgs
end
node(x::Traced) = x.node
node(x::Number) = x
jltype(::Traced{T}) where {T} = T
jltype(x::Number) = typeof(x)
traced(f, args...) = Traced{Base.promote_op(f, map(jltype, args)...)}(CaptureArgs(f, map(node, args)))
for f in keys(UNARY)
    @eval (::typeof($f))(x::Traced) = traced($f, x)
end
for f in keys(BINARY)
    @eval (::typeof($f))(x::Traced, y::Traced) = traced($f, x, y)
    @eval (::typeof($f))(x::Traced, y::Number) = traced($f, x, y)
    @eval (::typeof($f))(x::Number, y::Traced) = traced($f, x, y)
end
Base.ifelse(c::Traced{Bool}, x::Number, y::Number) = Traced{promote_type(jltype(x), jltype(y))}(CaptureArgs(ifelse, (node(c), node(x), node(y))))
function fprogram(f, eltypes, desttype)
    tree = try
        node(f(ntuple(k -> Traced{eltypes[k]}(ArgK(k)), length(eltypes))...))
    catch e
        throw(Unsupported("closure $(typeof(f)) could not be traced: $(typeof(e))"))   # stays on the CPU
    end
    tree isa Union{CaptureArgs,ArgK,Number} || throw(Unsupported("closure $(typeof(f)) returns $(typeof(tree))"))
    return fprogram(tree, eltypes, desttype)
end
# map!'s plain functions: identity / conj / a few arities of + and *
fprogram(::typeof(identity), eltypes, desttype) = Prog(UInt8[OP_ARG, 1], Float64[], 1, DataType[], false)
fprogram(::typeof(conj), eltypes, desttype) = Prog(UInt8[OP_ARG, 1, 11, 0], Float64[], 1, DataType[], false)
function fprogram(f::Union{typeof(+),typeof(*)}, eltypes, desttype)
    p = Prog(UInt8[OP_ARG, 1], Float64[], 1, DataType[], false)
    for k in 2:length(eltypes)
        push!(p.code, OP_ARG, UInt8(k), BINARY[f], 0x00)
    end
    return p
end

const REDOPS = Dict(nothing => 0, (+) => 1, Base.add_sum => 1, (*) => 2, Base.mul_prod => 2, min => 3, max => 4, (&) => 5, (|) => 6)
redcode(op) = get(() -> throw(Unsupported("reduction $op")), REDOPS, op)
const INITOPS = Dict(nothing => 0, identity => 1, zero => 2, conj => 5)
struct Scale{T}; β::T; end; (s::Scale)(x) = x * s.β     # x -> x*β and x -> β (src/linalg.jl:150,158) are opaque
struct Const{T}; β::T; end; (s::Const)(x) = s.β         # closures in Julia: __mul! passes these structs instead
initcode(s::Scale) = (3, complex(s.β))
initcode(s::Const) = (4, complex(s.β))
initcode(f) = (get(() -> throw(Unsupported("initop $(typeof(f))")), INITOPS, f), 0.0 + 0im)

# ---- recorded sequences (round 5): `@strided` code replayed by the library itself -------------------------------------------------
# A host loop that repeats the same few `@strided` statements pays a kernel boundary per statement even on the library's stream.
#     seq = StridedHIP.record() do
#         @strided permutedims!(B, A, (4, 3, 2, 1))
#         @strided C .= A .+ permutedims(A, (2, 3, 4, 1)) .+ permutedims(A, (3, 4, 1, 2)) .+ permutedims(A, (4, 1, 2, 3))
#     end
#     StridedHIP.replay(seq, 1000); StridedHIP.wait(seq)
# records the funnel calls of the block as plans (nothing runs while recording) and replays the list `reps` times with the results of
# in-order execution: statements that share no written data run concurrently on separate hardware queues, fences only where the data
# needs them (csrc/smr_seq.cpp; the bench step: 8.5 us in order -> 4.6 us).  `replay` returns when the work is queued (like `@spawn`);
# `wait` -- and every observation of a HipBuffer -- waits.  The arrays and the sequence must stay alive until then.
mutable struct Sequence
    handle::Ptr{Cvoid}
    plans::Vector{Ptr{Cvoid}}
    keep::Vector{Any}                 # operands, f-programs: everything the plans point into
end
const RECORDING = Ref{Union{Nothing,Sequence}}(nothing)
function record(body)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:smr_seq_create, lib), Cint, (Ptr{Ptr{Cvoid}},), h))
    seq = Sequence(h[], Ptr{Cvoid}[], Any[])
    finalizer(seq) do s
        ccall((:smr_seq_destroy, lib), Cint, (Ptr{Cvoid},), s.handle)
        foreach(p -> ccall((:smr_plan_destroy, lib), Cint, (Ptr{Cvoid},), p), s.plans)
    end
    RECORDING[] === nothing || error("StridedHIP.record: already recording")
    RECORDING[] = seq
    try
        body()
    finally
        RECORDING[] = nothing
    end
    return seq
end
replay(seq::Sequence, reps::Integer=1) = check(ccall((:smr_seq_run, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}), seq.handle, reps, stream()))
wait(seq::Sequence) = check(ccall((:smr_seq_wait, lib), Cint, (Ptr{Cvoid},), seq.handle))

# ---- the drop-in: one more method at the reference's funnel -----------------------------------------------
function _mapreduce_fuse!(f, op, initop, dims::Dims, arrays::Tuple{HipView,Vararg{HipView}})
    M, N = length(arrays), length(dims)
    try
        (N <= MAXN && M <= MAXM) || throw(Unsupported("rank/operand count"))
        prog = fprogram(f, map(eltype, Base.tail(arrays)), eltype(arrays[1]))
        ic, β = initcode(initop)
        ops = ntuple(k -> k <= M ? operand(arrays[k]) : NULLOP, MAXM)
        GC.@preserve arrays prog begin
            p = Ref(SmrProblem(N, M, pad(dims, 1), ops, pointer(prog.code), length(prog.code) ÷ 2,
                               length(prog.consts) ÷ 2, pointer(prog.consts), redcode(op), ic,
                               (real(β), imag(β)), stream()))
            # one process per GPU: after `smr_comm_init` (see INTEGRATION.md) the same call shards the box over the
            # ranks and all-reduces a split reduced dim; with a single rank it is plain smr_mapreduce.  An `f`
            # without a precompiled functor is compiled for gfx950 on first use (library-side, cached).
            if RECORDING[] !== nothing        # inside StridedHIP.record: the call becomes a plan of the sequence, nothing runs now
                seq = RECORDING[]
                plan = Ref{Ptr{Cvoid}}(C_NULL)
                check(ccall((:smr_plan_create, lib), Cint, (Ptr{SmrProblem}, Ptr{Ptr{Cvoid}}), p, plan))
                push!(seq.plans, plan[]); push!(seq.keep, (arrays, prog))
                check(ccall((:smr_seq_add, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Ptr{Cvoid}}), seq.handle, plan[], C_NULL))
                return arrays[1]
            end
            check(ccall((:smr_mapreduce_sharded, lib), Cint, (Ptr{SmrProblem},), p))
            ASYNC[] || synchronize()          # async!(false): wait for the result before returning
        end
    catch e
        e isa Unsupported || rethrow()
        # outside the device whitelist: run the reference's own CPU method on host copies, copy the result back
        host = map(a -> StridedView(download(a.parent), a.size, a.strides, a.offset, a.op), arrays)
        invoke(_mapreduce_fuse!, Tuple{Any,Any,Any,Dims,Tuple{Vararg{StridedView}}}, f, op, initop, dims, host)
        copyto!(arrays[1].parent, host[1].parent)
    end
    return arrays[1]
end

end # module
