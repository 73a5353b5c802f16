# MPSKitROCm.jl -- the reference-side binding of libmpsk (include/mpsk.h).
#
# NOT executed in this repository's CI: the build image has no Julia toolchain (DESIGN.md section 1).
# Its executable twin is mpskit.jl_amd/_lib.py + backend.py (ctypes, same argument lists).
#
# The reference has no plugin interface; the seams are multiple-dispatch methods (SURVEY.md
# section 8b).  This module adds a device tensor type and overloads exactly those methods:
#   (h::MPO_∂∂AC)(x), (h::MPO_∂∂C)(x), (h::MPO_∂∂AC2)(x)      src/algorithms/derivatives.jl:26-31
#   transfer_left / transfer_right (SparseMPOSlice)             src/transfermatrix/transfer.jl:137-143
#   leftorth!(; alg = QRpos()), rightorth!(; alg = LQpos()), tsvd!   (TensorKit, call sites orthoview.jl:52,56; dmrg.jl:96)
#   VectorInterface: inner, add!!, scale!!, zerovector, norm    (what KrylovKit needs, quasiparticle_state.jl:357-411)
module MPSKitROCm

using Libdl, MPSKit, TensorKit, VectorInterface
import MPSKit: MPO_∂∂AC, MPO_∂∂C, MPO_∂∂AC2, transfer_left, transfer_right, SparseMPOSlice

const libmpsk = Ref{String}(get(ENV, "LIBMPSK", "libmpsk.so"))
const CTX = Ref{Ptr{Cvoid}}(C_NULL)

struct MpskError <: Exception
    code::Cint
    msg::String
end
function check(rc::Cint)
    rc == 0 && return
    msg = unsafe_string(ccall((:mpsk_last_error, libmpsk[]), Cstring, ()))
    throw(MpskError(rc, msg))
end

function __init__()
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:mpsk_ctx_create, libmpsk[]), Cint, (Cint, Ref{Ptr{Cvoid}}), 0, h))
    CTX[] = h[]
end

# ---- device tensor: fp64, column-major, TensorKit index order (include/mpsk.h "Conventions") ----
mutable struct ROCTensor{N}
    ptr::Ptr{Cvoid}
    dims::NTuple{N,Int}
    function ROCTensor(dims::NTuple{N,Int}) where {N}
        p = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:mpsk_malloc, libmpsk[]), Cint, (Ptr{Cvoid}, Csize_t, Ref{Ptr{Cvoid}}), CTX[], 8 * prod(dims), p))
        t = new{N}(p[], dims)
        finalizer(x -> ccall((:mpsk_free, libmpsk[]), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), CTX[], x.ptr), t)
        return t
    end
end
Base.length(t::ROCTensor) = prod(t.dims)

"upload a trivial-sector TensorMap (its data matrix is already column-major in index order)"
function ROCTensor(t::AbstractTensorMap)
    a = convert(Array, t)
    eltype(a) <: Real || all(iszero, imag.(a)) || throw(ArgumentError("MPSK_C128 is reserved; real data only"))
    d = ROCTensor(size(a))
    h = Array{Float64}(real.(a))
    check(ccall((:mpsk_memcpy_h2d, libmpsk[]), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Csize_t), CTX[], d.ptr, h, sizeof(h)))
    return d
end
function Base.Array(d::ROCTensor)
    h = Array{Float64}(undef, d.dims...)
    check(ccall((:mpsk_memcpy_d2h, libmpsk[]), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Cvoid}, Csize_t), CTX[], h, d.ptr, sizeof(h)))
    return h
end

# ---- VectorInterface protocol for KrylovKit (the Krylov loop stays in Julia) ----
VectorInterface.scalartype(::Type{<:ROCTensor}) = Float64
function VectorInterface.inner(x::ROCTensor, y::ROCTensor)
    out = Ref{Float64}(0)
    check(ccall((:mpsk_vdot, libmpsk[]), Cint, (Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ref{Float64}), CTX[], length(x), x.ptr, y.ptr, out))
    return out[]
end
function LinearAlgebra_norm(x::ROCTensor)
    out = Ref{Float64}(0)
    check(ccall((:mpsk_vnrm2, libmpsk[]), Cint, (Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ref{Float64}), CTX[], length(x), x.ptr, out))
    return out[]
end
VectorInterface.zerovector(x::ROCTensor) = (y = ROCTensor(x.dims); check(ccall((:mpsk_vzero, libmpsk[]), Cint, (Ptr{Cvoid}, Int64, Ptr{Cvoid}), CTX[], length(y), y.ptr)); y)
function VectorInterface.add!!(y::ROCTensor, x::ROCTensor, α::Number=1, β::Number=1)
    check(ccall((:mpsk_vaxpby, libmpsk[]), Cint, (Ptr{Cvoid}, Int64, Float64, Ptr{Cvoid}, Float64, Ptr{Cvoid}), CTX[], length(x), Float64(α), x.ptr, Float64(β), y.ptr))
    return y
end
function VectorInterface.scale!!(x::ROCTensor, α::Number)
    check(ccall((:mpsk_vscal, libmpsk[]), Cint, (Ptr{Cvoid}, Int64, Float64, Ptr{Cvoid}), CTX[], length(x), Float64(α), x.ptr))
    return x
end

# ---- MPO slice handle: SparseMPOSlice -> mpsk_mposlice (sparseslice.jl:13-27) ----
struct ROCSlice
    handle::Ptr{Cvoid}
    d::Int
    Wl::Int
    Wr::Int
end
function ROCSlice(H::SparseMPOSlice)
    odim = H.odim
    chil = Int32[dim(H.domspaces[i]) for i in 1:odim]
    chir = Int32[dim(H.imspaces[j]) for j in 1:odim]
    d = dim(H.pspace)
    kind = zeros(Int32, odim, odim); scal = zeros(Float64, odim, odim)
    blocks = fill(C_NULL, odim, odim); keep = Any[]
    for (i, j) in keys(H)
        if MPSKit.isscal(H, i, j)
            kind[i, j] = 1; scal[i, j] = real(H.Os[i, j])
        else
            a = Array{Float64}(real.(convert(Array, H[i, j])))   # [chi_i, d, d, chi_j], column-major
            push!(keep, a); kind[i, j] = 2; blocks[i, j] = pointer(a)
        end
    end
    h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve keep check(ccall((:mpsk_mposlice_create, libmpsk[]), Cint,
        (Ptr{Cvoid}, Cint, Cint, Ptr{Int32}, Ptr{Int32}, Cint, Ptr{Int32}, Ptr{Float64}, Ptr{Ptr{Cvoid}}, Ref{Ptr{Cvoid}}),
        CTX[], 0, odim, chil, chir, d, kind, scal, blocks, h))
    return ROCSlice(h[], d, sum(chil), sum(chir))
end

# ---- the hot matvec: (h::MPO_∂∂AC)(x)  derivatives.jl:29,77-104 ----
# leftenv / rightenv are ROCTensor{3} of slabs (W, D, D) built by the overloaded transfer_left/right.
function (h::MPO_∂∂AC{ROCSlice,<:ROCTensor,<:ROCTensor})(x::ROCTensor{3})
    Dl, d, Dr = x.dims
    y = ROCTensor((h.leftenv.dims[2], d, Dr))
    check(ccall((:mpsk_dAC, libmpsk[]), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
        CTX[], h.o.handle, h.leftenv.dims[2], Dl, Dr, h.leftenv.ptr, h.rightenv.ptr, x.ptr, y.ptr))
    return y
end
function (h::MPO_∂∂C{<:ROCTensor,<:ROCTensor})(c::ROCTensor{2})
    Dl, Dr = c.dims
    y = ROCTensor((h.leftenv.dims[2], Dr))
    check(ccall((:mpsk_dC, libmpsk[]), Cint,
        (Ptr{Cvoid}, Cint, Cint, Cint, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
        CTX[], h.leftenv.dims[1], h.leftenv.dims[2], Dl, Dr, h.leftenv.ptr, h.rightenv.ptr, c.ptr, y.ptr))
    return y
end
function (h::MPO_∂∂AC2{ROCSlice,<:ROCTensor,<:ROCTensor})(x::ROCTensor{4})
    Dl, d1, Dr, d2 = x.dims
    y = ROCTensor((h.leftenv.dims[2], d1, Dr, d2))
    check(ccall((:mpsk_dAC2, libmpsk[]), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
        CTX[], h.o1.handle, h.o2.handle, h.leftenv.dims[2], Dl, Dr, h.leftenv.ptr, h.rightenv.ptr, x.ptr, y.ptr))
    return y
end

# ---- environment updates: transfer.jl:166-259 ----
function transfer_left(v::ROCTensor{3}, H::ROCSlice, A::ROCTensor{3}, Ab::ROCTensor{3})
    out = ROCTensor((H.Wr, Ab.dims[3], A.dims[3]))
    check(ccall((:mpsk_transfer_left, libmpsk[]), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint, Cint, Cint, Cint, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
        CTX[], H.handle, H.Wl, H.d, A.dims[1], A.dims[3], Ab.dims[1], Ab.dims[3], v.ptr, A.ptr, Ab.ptr, out.ptr))
    return out
end
function transfer_right(v::ROCTensor{3}, H::ROCSlice, A::ROCTensor{3}, Ab::ROCTensor{3})
    out = ROCTensor((H.Wl, A.dims[1], Ab.dims[1]))
    check(ccall((:mpsk_transfer_right, libmpsk[]), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint, Cint, Cint, Cint, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
        CTX[], H.handle, H.Wr, H.d, A.dims[1], A.dims[3], Ab.dims[1], Ab.dims[3], A.ptr, Ab.ptr, v.ptr, out.ptr))
    return out
end

# ---- gauge steps (TensorKit leftorth!/rightorth!/tsvd! on the matricised tensor) ----
function qrpos(A::ROCTensor{2})
    m, n = A.dims
    Q, R = ROCTensor((m, n)), ROCTensor((n, n))
    check(ccall((:mpsk_qrpos, libmpsk[]), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Cint),
        CTX[], m, n, A.ptr, m, Q.ptr, m, R.ptr, n))
    return Q, R
end
function lqpos(A::ROCTensor{2})
    m, n = A.dims
    L, Q = ROCTensor((m, m)), ROCTensor((m, n))
    check(ccall((:mpsk_lqpos, libmpsk[]), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Cint),
        CTX[], m, n, A.ptr, m, L.ptr, m, Q.ptr, m))
    return L, Q
end
function tsvd(theta::ROCTensor{2}; truncdim::Int=0, truncerr::Float64=0.0)
    m, n = theta.dims; k = min(m, n)
    U, S, Vh = ROCTensor((m, k)), ROCTensor((k,)), ROCTensor((k, n))
    kept = Ref{Cint}(0); disc = Ref{Float64}(0)
    check(ccall((:mpsk_tsvd, libmpsk[]), Cint,
        (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint, Float64, Ref{Cint}, Ref{Float64}),
        CTX[], m, n, theta.ptr, m, U.ptr, m, S.ptr, Vh.ptr, k, truncdim, truncerr, kept, disc))
    return U, S, Vh, Int(kept[]), disc[]
end

end # module
