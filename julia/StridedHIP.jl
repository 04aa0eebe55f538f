# StridedHIP.jl -- the reference-side binding a Strided.jl maintainer would add so that
# StridedViews whose parent lives in MI355X memory run on libstrided_hip.so.
#
# NOT EXECUTED IN THIS REPOSITORY: there is no Julia runtime in the build image or on the GPU
# box.  It is kept deliberately thin: every method below only *serialises* what the reference's
# funnel already holds -- `_mapreduce_fuse!(f, op, initop, dims, arrays)`, src/mapreduce.jl:98 --
# into the C struct of include/strided_hip.h and `ccall`s it.  All view algebra, broadcasting
# style rules, `promoteshape`, argument checks and the `@strided` macro stay the reference's own.
module StridedHIP

using Strided, StridedViews
using Strided: CaptureArgs, Arg
import Strided: _mapreduce_fuse!

const lib = get(ENV, "STRIDED_HIP_LIB", "libstrided_hip.so")
const MAXN, MAXM = 8, 8

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
Base.similar(b::HipBuffer, ::Type{T}, dims::Dims) where {T} = HipBuffer{T}(undef, dims)
Base.unsafe_convert(::Type{Ptr{T}}, b::HipBuffer{T}) where {T} = b.ptr
function upload(a::Array{T,N}) where {T,N}
    b = HipBuffer{T}(undef, size(a))
    check(ccall((:smr_memcpy_h2d, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Ptr{Cvoid}), b.ptr, a, sizeof(a), C_NULL))
    return b
end
function download(b::HipBuffer{T,N}) where {T,N}
    a = Array{T,N}(undef, size(b))
    check(ccall((:smr_memcpy_d2h, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Ptr{Cvoid}), a, b.ptr, sizeof(a), C_NULL))
    check(ccall((:smr_stream_sync, lib), Cint, (Ptr{Cvoid},), C_NULL))
    return a
end

function check(rc::Cint)
    rc == 0 && return nothing
    msg = unsafe_string(ccall((:smr_last_error, lib), Cstring, ()))
    rc == -2 && throw(Unsupported(msg))          # SMR_EUNSUPPORTED -> CPU fallback below
    rc == -1 && throw(ArgumentError(msg))        # SMR_EINVAL
    error("libstrided_hip: $msg (status $rc)")
end
struct Unsupported <: Exception
    msg::String
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

dtypecode(::Type{Float32}) = 0; dtypecode(::Type{Float64}) = 1
dtypecode(::Type{ComplexF32}) = 2; dtypecode(::Type{ComplexF64}) = 3
dtypecode(::Type{Int8}) = 4; dtypecode(::Type{Int16}) = 5; dtypecode(::Type{Int32}) = 6; dtypecode(::Type{Int64}) = 7
dtypecode(::Type{UInt8}) = 8; dtypecode(::Type{Bool}) = 8
dtypecode(T) = throw(Unsupported("eltype $T"))

pad(t::NTuple{N,Int}, v) where {N} = ntuple(i -> i <= N ? Int64(t[i]) : Int64(v), MAXN)
function operand(a::StridedView{T,N}) where {T,N}
    return SmrOperand(pointer(a.parent), a.offset, pad(a.strides, 0), dtypecode(T), a.op === conj ? 1 : 0)
end
const NULLOP = SmrOperand(C_NULL, 0, ntuple(_ -> Int64(0), MAXN), 0, 0)

# ---- f -> postfix f-program (walks the CaptureArgs tree of src/broadcast.jl:67-83) ---------------------
const UNARY = Dict(:- => 8, abs => 9, abs2 => 10, conj => 11, real => 12, imag => 13, sqrt => 14, exp => 15,
                   log => 16, sin => 17, cos => 18, tanh => 19, inv => 20)
const BINARY = Dict(+ => 32, - => 33, * => 34, / => 35, min => 36, max => 37, < => 38, <= => 39, > => 40,
                    >= => 41, == => 42, != => 43)
mutable struct Prog
    code::Vector{UInt8}
    consts::Vector{Float64}
    nextarg::Int
end
emit!(p::Prog, ::Arg) = (p.nextarg += 1; push!(p.code, 0x00, UInt8(p.nextarg)))
function emit!(p::Prog, x::Number)
    push!(p.consts, real(x), imag(x))
    push!(p.code, 0x01, UInt8(length(p.consts) ÷ 2 - 1))
end
function emit!(p::Prog, c::CaptureArgs)
    f, args = c.f, c.args
    if length(args) == 1
        f === (-) ? (emit!(p, args[1]); push!(p.code, 8, 0)) :
        haskey(UNARY, f) ? (emit!(p, args[1]); push!(p.code, UNARY[f], 0)) : throw(Unsupported("function $f"))
    elseif haskey(BINARY, f) && (length(args) == 2 || f in (+, *, min, max))
        emit!(p, args[1])                      # Julia's +(a,b,c,d) folds left
        for a in args[2:end]
            emit!(p, a); push!(p.code, BINARY[f], 0)
        end
    elseif f === ifelse && length(args) == 3
        foreach(a -> emit!(p, a), args); push!(p.code, 64, 0)
    else
        throw(Unsupported("function $f"))
    end
end
emit!(p::Prog, x) = throw(Unsupported("captured $(typeof(x))"))
# map!'s plain functions: identity / conj / a few arities of + and *
fprogram(::typeof(identity), M) = Prog(UInt8[0, 1], Float64[], 1)
fprogram(::typeof(conj), M) = Prog(UInt8[0, 1, 11, 0], Float64[], 1)
fprogram(c::CaptureArgs, M) = (p = Prog(UInt8[], Float64[], 0); emit!(p, c); p)
function fprogram(f::Union{typeof(+),typeof(*)}, M)
    p = Prog(UInt8[0, 1], Float64[], 1)
    for k in 2:(M - 1)
        push!(p.code, 0, UInt8(k), BINARY[f], 0)
    end
    return p
end
fprogram(f, M) = throw(Unsupported("closure $(typeof(f))"))   # arbitrary closures stay on the CPU

redcode(::Nothing) = 0
redcode(::Union{typeof(+),typeof(Base.add_sum)}) = 1
redcode(::Union{typeof(*),typeof(Base.mul_prod)}) = 2
redcode(::typeof(min)) = 3; redcode(::typeof(max)) = 4
redcode(::typeof(&)) = 5; redcode(::typeof(|)) = 6           # neutral elements true / false (src/mapreduce.jl:188-189)
redcode(op) = throw(Unsupported("reduction $op"))
initcode(::Nothing) = (0, 0.0 + 0im); initcode(::typeof(identity)) = (1, 0.0 + 0im)
initcode(::typeof(zero)) = (2, 0.0 + 0im); initcode(::typeof(conj)) = (5, 0.0 + 0im)
initcode(f) = throw(Unsupported("initop $(typeof(f))"))      # x->x*β / x->β: see `Scale`, `Const`
struct Scale{T}; β::T; end; (s::Scale)(x) = x * s.β; initcode(s::Scale) = (3, complex(s.β))
struct Const{T}; β::T; end; (s::Const)(x) = s.β; initcode(s::Const) = (4, complex(s.β))

const HipView = StridedView{<:Any,<:Any,<:HipBuffer}

# ---- the drop-in: one more method at the reference's funnel -----------------------------------------------
function _mapreduce_fuse!(f, op, initop, dims::Dims, arrays::Tuple{HipView,Vararg{HipView}})
    M, N = length(arrays), length(dims)
    try
        (N <= MAXN && M <= MAXM) || throw(Unsupported("rank/operand count"))
        prog = fprogram(f, M)
        ic, β = initcode(initop)
        ops = ntuple(k -> k <= M ? operand(arrays[k]) : NULLOP, MAXM)
        GC.@preserve arrays prog begin
            p = Ref(SmrProblem(N, M, pad(dims, 1), ops, pointer(prog.code), length(prog.code) ÷ 2,
                               length(prog.consts) ÷ 2, pointer(prog.consts), redcode(op), ic,
                               (real(β), imag(β)), C_NULL))
            # one process per GPU: after `smr_comm_init` (see INTEGRATION.md) the same call shards the box over the
            # ranks and all-reduces a split reduced dim; with a single rank it is plain smr_mapreduce.  An `f`
            # without a precompiled functor is compiled for gfx950 on first use (library-side, cached).
            check(ccall((:smr_mapreduce_sharded, lib), Cint, (Ptr{SmrProblem},), p))
            check(ccall((:smr_stream_sync, lib), Cint, (Ptr{Cvoid},), C_NULL))   # the reference is synchronous
        end
        return arrays[1]
    catch e
        e isa Unsupported || rethrow()
        # outside the device whitelist: run the reference's own CPU method on host copies
        host = map(a -> StridedView(download(a.parent), a.size, a.strides, a.offset, a.op), arrays)
        invoke(_mapreduce_fuse!, Tuple{Any,Any,Any,Dims,Tuple{Vararg{StridedView}}}, f, op, initop, dims, host)
        copyto!(arrays[1].parent, upload(host[1].parent))
        return arrays[1]
    end
end

end # module
