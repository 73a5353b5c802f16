# MPSKitROCm.jl -- the reference-side binding of libmpsk (include/mpsk.h).
#
# SKETCH -- never executed: the build image has no Julia toolchain (DESIGN.md section 1).  It documents the ccall
# signatures and where they hook into the reference; INTEGRATION.md ("What the shim text can and cannot claim") lists
# what a real integration needs on top (ROCTensor as an AbstractTensorMap or device twins of FiniteMPS / FinEnv).
# Its executable twin is mpskit.jl_amd/_lib.py + backend.py (ctypes, same argument lists).
#
# The reference has no plugin interface; the seams are multiple-dispatch methods (SURVEY.md
# section 8b).  This module adds a device tensor type and overloads exactly those methods:
#   (h::MPO_∂∂AC)(x), (h::MPO_∂∂C)(x), (h::MPO_∂∂AC2)(x)      src/algorithms/derivatives.jl:26-31
#   transfer_left / transfer_right (SparseMPOSlice)             src/transfermatrix/transfer.jl:137-143
#   leftorth!(; alg = QRpos()), rightorth!(; alg = LQpos()), tsvd!   (TensorKit, call sites orthoview.jl:52,56; dmrg.jl:96)
#   VectorInterface: inner, add!!, scale!!, zerovector, norm    (what KrylovKit needs, quasiparticle_state.jl:357-411)
module MPSKitROCm

using Libdl, LinearAlgebra, MPSKit, TensorKit, VectorInterface
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

# ---- device tensor: fp64 or interleaved complex128, column-major, TensorKit index order (include/mpsk.h "Conventions") ----
mutable struct ROCTensor{N}
    ptr::Ptr{Cvoid}
    dims::NTuple{N,Int}
    cplx::Bool            # MPSK_C128: (re, im) pairs, the bytes of an Array{ComplexF64}
    function ROCTensor(dims::NTuple{N,Int}; cplx::Bool=false) where {N}
        p = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:mpsk_malloc, libmpsk[]), Cint, (Ptr{Cvoid}, Csize_t, Ref{Ptr{Cvoid}}), CTX[], (cplx ? 16 : 8) * prod(dims), p))
        t = new{N}(p[], dims, cplx)
        finalizer(x -> ccall((:mpsk_free, libmpsk[]), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), CTX[], x.ptr), t)
        return t
    end
end
Base.length(t::ROCTensor) = prod(t.dims)
nreal(t::ROCTensor) = (t.cplx ? 2 : 1) * prod(t.dims)   # doubles behind the pointer: what the vector helpers run over

"upload a trivial-sector TensorMap (its data matrix is already column-major in index order)"
function ROCTensor(t::AbstractTensorMap)
    a = convert(Array, t)
    cx = !(eltype(a) <: Real)
    d = ROCTensor(size(a); cplx=cx)
    h = cx ? Array{ComplexF64}(a) : Array{Float64}(a)         # complex: uploaded as is (interleaved = MPSK_C128 layout)
    check(ccall((:mpsk_memcpy_h2d, libmpsk[]), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t), CTX[], d.ptr, h, sizeof(h)))
    return d
end
function Base.Array(d::ROCTensor)
    h = d.cplx ? Array{ComplexF64}(undef, d.dims...) : Array{Float64}(undef, d.dims...)
    check(ccall((:mpsk_memcpy_d2h, libmpsk[]), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t), CTX[], h, d.ptr, sizeof(h)))
    return h
end

# ---- VectorInterface protocol for KrylovKit (the Krylov loop stays in Julia) ----
# The vector helpers run over the nreal(x) doubles of a tensor.  For complex tensors `inner` below is Re<x, y> -- what the
# Hermitian Lanczos solvers of the hot path need (fixedpoint.jl:19-30, integrators.jl:20-25); a general complex inner
# product needs a second pass (Im<x, y> = <x, J y> with mpsk_vtimes_i) and is left to the integrator.
VectorInterface.scalartype(::Type{<:ROCTensor}) = Float64
function VectorInterface.inner(x::ROCTensor, y::ROCTensor)
    out = Ref{Float64}(0)
    check(ccall((:mpsk_vdot, libmpsk[]), Cint, (Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ref{Float64}), CTX[], nreal(x), x.ptr, y.ptr, out))
    return out[]
end
function LinearAlgebra.norm(x::ROCTensor)
    out = Ref{Float64}(0)
    check(ccall((:mpsk_vnrm2, libmpsk[]), Cint, (Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ref{Float64}), CTX[], nreal(x), x.ptr, out))
    return out[]
end
VectorInterface.zerovector(x::ROCTensor) = (y = ROCTensor(x.dims; cplx=x.cplx); check(ccall((:mpsk_vzero, libmpsk[]), Cint, (Ptr{Cvoid}, Int64, Ptr{Cvoid}), CTX[], nreal(y), y.ptr)); y)
VectorInterface.zerovector!!(x::ROCTensor) = (check(ccall((:mpsk_vzero, libmpsk[]), Cint, (Ptr{Cvoid}, Int64, Ptr{Cvoid}), CTX[], nreal(x), x.ptr)); x)
VectorInterface.scale(x::ROCTensor, α::Number) = VectorInterface.scale!!(copy(x), α)
VectorInterface.scale!(x::ROCTensor, α::Number) = VectorInterface.scale!!(x, α)
VectorInterface.add(y::ROCTensor, x::ROCTensor, α::Number=1, β::Number=1) = VectorInterface.add!!(copy(y), x, α, β)
VectorInterface.add!(y::ROCTensor, x::ROCTensor, α::Number=1, β::Number=1) = VectorInterface.add!!(y, x, α, β)
function Base.copy(x::ROCTensor)
    y = ROCTensor(x.dims; cplx=x.cplx)
    check(ccall((:mpsk_vcopy, libmpsk[]), Cint, (Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Cvoid}), CTX[], nreal(x), x.ptr, y.ptr))
    return y
end
function VectorInterface.add!!(y::ROCTensor, x::ROCTensor, α::Number=1, β::Number=1)
    check(ccall((:mpsk_vaxpby, libmpsk[]), Cint, (Ptr{Cvoid}, Int64, Float64, Ptr{Cvoid}, Float64, Ptr{Cvoid}), CTX[], nreal(x), Float64(α), x.ptr, Float64(β), y.ptr))
    return y
end
function VectorInterface.scale!!(x::ROCTensor, α::Number)
    check(ccall((:mpsk_vscal, libmpsk[]), Cint, (Ptr{Cvoid}, Int64, Float64, Ptr{Cvoid}), CTX[], nreal(x), Float64(α), x.ptr))
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
    cx = !(scalartype(H) <: Real)                       # MPSK_C128 slice: interleaved complex scalars / blocks
    kind = zeros(Int32, odim, odim); scal = cx ? zeros(ComplexF64, odim, odim) : zeros(Float64, odim, odim)
    blocks = fill(C_NULL, odim, odim); keep = Any[]
    for (i, j) in keys(H)
        if MPSKit.isscal(H, i, j)
            kind[i, j] = 1; scal[i, j] = cx ? H.Os[i, j] : real(H.Os[i, j])
        else
            a = cx ? Array{ComplexF64}(convert(Array, H[i, j])) : Array{Float64}(real.(convert(Array, H[i, j])))   # [chi_i, d, d, chi_j]
            push!(keep, a); kind[i, j] = 2; blocks[i, j] = pointer(a)
        end
    end
    h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve keep check(ccall((:mpsk_mposlice_create, libmpsk[]), Cint,
        (Ptr{Cvoid}, Cint, Cint, Ptr{Int32}, Ptr{Int32}, Cint, Ptr{Int32}, Ptr{Cvoid}, Ptr{Ptr{Cvoid}}, Ref{Ptr{Cvoid}}),
        CTX[], cx ? 1 : 0, odim, chil, chir, d, kind, scal, blocks, h))
    return ROCSlice(h[], d, sum(chil), sum(chir))
end

# ---- the prepared operator: MPO_∂∂AC built once per site visit (derivatives.jl:11-15,44-46), applied per Krylov step ----
mutable struct ROCddAC <: MPSKit.DerivativeOperator
    handle::Ptr{Cvoid}
    leftenv::ROCTensor{3}      # kept alive: mpsk_hac_create does not copy GL / GR
    rightenv::ROCTensor{3}
    function ROCddAC(o::ROCSlice, GL::ROCTensor{3}, GR::ROCTensor{3})
        h = Ref{Ptr{Cvoid}}(C_NULL)
        Dlo, Dl, Dr = GL.dims[2], GL.dims[3], GR.dims[3]
        check(ccall((:mpsk_hac_create, libmpsk[]), Cint,
            (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ref{Ptr{Cvoid}}),
            CTX[], o.handle, Dlo, Dl, Dr, GL.ptr, GR.ptr, h))
        t = new(h[], GL, GR)
        finalizer(x -> ccall((:mpsk_hac_destroy, libmpsk[]), Cint, (Ptr{Cvoid},), x.handle), t)
        return t
    end
end
function (h::ROCddAC)(x::ROCTensor{3})
    y = ROCTensor((h.leftenv.dims[2], x.dims[2], x.dims[3]); cplx=x.cplx)
    check(ccall((:mpsk_hac_apply, libmpsk[]), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Cvoid}), h.handle, x.ptr, 1, y.ptr))
    return y
end

# ---- the one-shot matvec: (h::MPO_∂∂AC)(x)  derivatives.jl:29,77-104 ----
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
# mpsk_qrpos / mpsk_lqpos / mpsk_tsplit follow the ctx dtype (include/mpsk.h): under MPSK_C128 their operands are the
# interleaved bytes of an Array{ComplexF64}, leading dimensions count complex elements
function with_dtype(f, cplx::Bool)
    check(ccall((:mpsk_ctx_set_dtype, libmpsk[]), Cint, (Ptr{Cvoid}, Cint), CTX[], cplx ? 1 : 0))
    try
        return f()
    finally
        cplx && check(ccall((:mpsk_ctx_set_dtype, libmpsk[]), Cint, (Ptr{Cvoid}, Cint), CTX[], 0))
    end
end
function qrpos(A::ROCTensor{2})
    m, n = A.dims
    Q, R = ROCTensor((m, n); cplx=A.cplx), ROCTensor((n, n); cplx=A.cplx)
    with_dtype(A.cplx) do
        check(ccall((:mpsk_qrpos, libmpsk[]), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Cint),
            CTX[], m, n, A.ptr, m, Q.ptr, m, R.ptr, n))
    end
    return Q, R
end
function lqpos(A::ROCTensor{2})
    m, n = A.dims
    L, Q = ROCTensor((m, m); cplx=A.cplx), ROCTensor((m, n); cplx=A.cplx)
    with_dtype(A.cplx) do
        check(ccall((:mpsk_lqpos, libmpsk[]), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Cint),
            CTX[], m, n, A.ptr, m, L.ptr, m, Q.ptr, m))
    end
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

# ---- MPO-less transfers + regularize!  (transfer.jl:18-45,66-75 ; transfermatrix.jl:70-76) ----
# v is a slab tensor (W, D, D); W = 1 for the bond tensors of the uniform gauge / the identity levels of the
# infinite environments.  The same two entry points take the excited tensor B of a quasiparticle state as the ket
# (transfer.jl:48-62,113-126 with a trivial utility leg) and mixed (AR, AL) ket / bra pairs (qpenv.jl:66-97).
function transfer_left(v::ROCTensor{3}, ::Nothing, A::ROCTensor{3}, Ab::ROCTensor{3})
    W = v.dims[1]
    out = ROCTensor((W, Ab.dims[3], A.dims[3]))
    check(ccall((:mpsk_transfer_left, libmpsk[]), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint, Cint, Cint, Cint, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
        CTX[], C_NULL, W, A.dims[2], A.dims[1], A.dims[3], Ab.dims[1], Ab.dims[3], v.ptr, A.ptr, Ab.ptr, out.ptr))
    return out
end
function transfer_right(v::ROCTensor{3}, ::Nothing, A::ROCTensor{3}, Ab::ROCTensor{3})
    W = v.dims[1]
    out = ROCTensor((W, A.dims[1], Ab.dims[1]))
    check(ccall((:mpsk_transfer_right, libmpsk[]), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint, Cint, Cint, Cint, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
        CTX[], C_NULL, W, A.dims[2], A.dims[1], A.dims[3], Ab.dims[1], Ab.dims[3], A.ptr, Ab.ptr, v.ptr, out.ptr))
    return out
end
"v[w] -= <lvec, v[w]> rvec on every slab (RegTransferMatrix, transfermatrix.jl:55,70-90)"
function regularize!(v::ROCTensor{3}, lvec::ROCTensor{2}, rvec::ROCTensor{2})
    check(ccall((:mpsk_regularize, libmpsk[]), Cint, (Ptr{Cvoid}, Cint, Cint, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
        CTX[], v.dims[1], v.dims[2], v.dims[3], v.ptr, lvec.ptr, rvec.ptr))
    return v
end

# ---- small gauge products  AC = AL*C, C*AR, theta = AC*AR  (orthoview.jl:99,103 ; dmrg.jl:92) ----
# (complex operands: trans = adjoint, alpha real; mpsk_gemm follows the ctx dtype)
function mul(A::ROCTensor{2}, B::ROCTensor{2}; transA::Bool=false, transB::Bool=false, alpha=1.0)
    M = transA ? A.dims[2] : A.dims[1]; K = transA ? A.dims[1] : A.dims[2]; N = transB ? B.dims[1] : B.dims[2]
    @assert A.cplx == B.cplx
    Cm = ROCTensor((M, N); cplx=A.cplx)
    with_dtype(A.cplx) do
        check(ccall((:mpsk_gemm, libmpsk[]), Cint,
            (Ptr{Cvoid}, Cint, Cint, Cint, Cint, Cint, Float64, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Int64, Float64, Ptr{Cvoid}, Int64),
            CTX[], transA, transB, M, N, K, Float64(alpha), A.ptr, A.dims[1], B.ptr, B.dims[1], 0.0, Cm.ptr, M))
    end
    return Cm
end

# ---- the two-site split  al, c, ar = tsvd!(ac2; trunc) ; normalize!(c)   (dmrg.jl:96-104, tdvp.jl:124-126) ----
# c comes back triangular instead of diagonal (al*c*ar is the same truncated theta, al / ar isometries): no rotation
# accumulation in the Jacobi sweeps.  S is a min(m, n) buffer (include/mpsk.h): the kept Schmidt values lead it; with the
# truncation-aware default (svd mode 3) only the first r = truncdim + max(64, truncdim / 2) entries are computed, NaN behind.
# Complex theta (truncdim scheme only): al / ar complex isometries, c lower triangular with a real positive diagonal.
function tsplit(theta::ROCTensor{2}; truncdim::Int=0, truncerr::Float64=0.0)
    m, n = theta.dims; k = min(m, n); kmax = truncdim > 0 ? min(k, truncdim) : k
    cx = theta.cplx
    AL, Cm, AR = ROCTensor((m, kmax); cplx=cx), ROCTensor((kmax, kmax); cplx=cx), ROCTensor((kmax, n); cplx=cx)
    S = ROCTensor((k,))                                           # singular values are real
    kept = Ref{Cint}(0); disc = Ref{Float64}(0)
    with_dtype(cx) do
        check(ccall((:mpsk_tsplit, libmpsk[]), Cint,
            (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Cint, Cint, Float64, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ref{Cint}, Ref{Float64}),
            CTX[], m, n, theta.ptr, m, truncdim, truncerr, AL.ptr, m, Cm.ptr, kmax, AR.ptr, kmax, S.ptr, kept, disc))
    end
    return AL, Cm, AR, S, Int(kept[]), disc[]
end

# ---- two QRpos factorizations in flight (old and new AC of a right-moving site update: toolbox.jl:20 + orthoview.jl:56) ----
function qrpos2(A1::ROCTensor{2}, A2::ROCTensor{2})
    m, n = A1.dims
    Q1, R1, Q2, R2 = ROCTensor((m, n)), ROCTensor((n, n)), ROCTensor((m, n)), ROCTensor((n, n))
    check(ccall((:mpsk_qrpos2, libmpsk[]), Cint,
        (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Cint),
        CTX[], m, n, A1.ptr, m, Q1.ptr, m, R1.ptr, n, A2.ptr, m, Q2.ptr, m, R2.ptr, n))
    return (Q1, R1), (Q2, R2)
end

# ---- fused Krylov orthogonalisation step (KrylovKit ModifiedGramSchmidt2 replaced by CGS2 + normalise, ONE sync) ----
function orth_step!(basis::Vector{<:ROCTensor}, y::ROCTensor)
    k = length(basis)
    ptrs = Ptr{Cvoid}[b.ptr for b in basis]
    h = zeros(Float64, k); beta = Ref{Float64}(0)
    check(ccall((:mpsk_vorth_step, libmpsk[]), Cint, (Ptr{Cvoid}, Int64, Cint, Ptr{Ptr{Cvoid}}, Ptr{Cvoid}, Ptr{Float64}, Ref{Float64}),
        CTX[], length(y), k, ptrs, y.ptr, h, beta))
    return h, beta[]
end

# ---- fixed-budget :SR solve without a host synchronisation (what mpskit.jl_amd/krylov.py does with values = false) ----
# KrylovKit's eigsolve decides convergence on the host after every step; for `Arnoldi(; krylovdim = m, maxiter = 1)` with a
# fixed number of steps nothing has to be decided, so the recurrence, the Ritz step of the projected matrix and the assembly of
# the Ritz vector can all stay on the device.  `slot` holds the (2k+1) scalars of every step, `buf` the Ritz coefficients.
function fixedpoint_fixed_budget(h::ROCddAC, x0::ROCTensor, m::Int)
    @assert 1 <= m <= 32
    stride = 2m + 1
    slot = ROCTensor((m * stride,)); buf = ROCTensor((40,))
    V = Vector{typeof(x0)}(undef, m + 1)
    V[1] = ROCTensor(x0.dims; cplx=x0.cplx)
    check(ccall((:mpsk_vnormalize_dev, libmpsk[]), Cint, (Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
        CTX[], nreal(x0), x0.ptr, V[1].ptr, C_NULL))
    for k in 1:m
        V[k + 1] = h(V[k])                                  # mpsk_hac_apply
        ptrs = Ptr{Cvoid}[v.ptr for v in V[1:k]]
        check(ccall((:mpsk_vorth_step_dev, libmpsk[]), Cint, (Ptr{Cvoid}, Int64, Cint, Ptr{Ptr{Cvoid}}, Ptr{Cvoid}, Ptr{Cvoid}),
            CTX[], nreal(x0), k, ptrs, V[k + 1].ptr, slot.ptr + 8 * (k - 1) * stride))
    end
    check(ccall((:mpsk_vritz_dev, libmpsk[]), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
        CTX[], m, stride, slot.ptr, buf.ptr, buf.ptr + 8 * 32))
    y = ROCTensor(x0.dims; cplx=x0.cplx)
    ptrs = Ptr{Cvoid}[v.ptr for v in V[1:m]]
    check(ccall((:mpsk_vlincomb_dev, libmpsk[]), Cint, (Ptr{Cvoid}, Int64, Cint, Ptr{Ptr{Cvoid}}, Ptr{Cvoid}, Ptr{Cvoid}),
        CTX[], nreal(x0), m, ptrs, buf.ptr, y.ptr))
    check(ccall((:mpsk_vnormalize_dev, libmpsk[]), Cint, (Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
        CTX[], nreal(x0), y.ptr, y.ptr, C_NULL))
    return y                                                # eigenvalue / residual estimate: buf[33:35] (device)
end

# ---- deferred completion of a gauge step (the left-moving DMRG visit: LQ of the new AC, galerkin evaluation, commit) ----
qr_defer() = check(ccall((:mpsk_ctx_qr_defer, libmpsk[]), Cint, (Ptr{Cvoid},), CTX[]))
function qr_commit()
    redone = Ref{Cint}(0)
    check(ccall((:mpsk_qr_commit, libmpsk[]), Cint, (Ptr{Cvoid}, Ref{Cint}), CTX[], redone))
    return redone[]
end

end # module
