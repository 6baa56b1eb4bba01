"""
    JchemoHIP

Julia host side of the MI355X-native PLS engine: the same call shapes as Jchemo.jl
(`plskern`, `plskern!`, `plsnipals`, `plsnipals!`, `transform`, `coef`, `predict`, `summary`), implemented
as thin `ccall`s into `libjchemo_hip.so` (C ABI: include/jchemo_hip.h).  AMDGPU.jl is used only as a handle
for device buffers (`ROCArray`); there is no CUDA path and no CPU fallback.

What a fit returns
  * Jchemo.jl loaded in the session (`using Jchemo` before or after `using JchemoHIP`) and HOST arrays in:
    the reference's own record `Jchemo.Plsr(T, P, R, W, C, TT, xmeans, xscales, ymeans, yscales, weights, niter)`
    (src/plskern.jl:1-14), so every consumer of the reference (`Jchemo.transform / coef / predict / summary`,
    `gridscorelv(...; fun = JchemoHIP.plskern)`, `locwlv`, `plsrda`, ... SURVEY §3.5) accepts it unchanged.
  * otherwise (Jchemo not loaded, or device-resident `ROCArray` inputs whose scores stay on the GPU): the
    fallback record `JchemoHIP.Plsr` below — same field names, same shapes, `T` / `weights` of the input's
    array type.  `attach!(Jchemo)` (called automatically when Jchemo is found) adds
    `Jchemo.transform / coef / predict` methods for it.
The accessors of this module (`JchemoHIP.transform`, `coef`, `predict`, `explvarx`) take EITHER record (they only
read the fields) and run on the GPU.

NOT EXECUTED IN THIS REPOSITORY'S CI: the build image has no Julia toolchain (see DESIGN.md).  Every
behaviour below is exercised through the identical C entry points by the ctypes mirror in
`jchemo.jl_amd/jchemo_hip/` (tests/test_gpu_parity.py); tests/test_julia_wrapper.py checks every `ccall` of this
file against include/jchemo_hip.h (literal signatures, argument counts and types).
"""
module JchemoHIP

using LinearAlgebra

export Plsr, Lwplsr, plskern, plskern!, plsnipals, plsnipals!, plssimp, plssimp!, plsrosa, plsrosa!, plswold, plswold!,
       lwplsr, transform, coef, predict, explvarx, JchCtx, attach!

const LIB = get(ENV, "JCHEMO_HIP_LIB", joinpath(@__DIR__, "..", "lib", "libjchemo_hip.so"))

# ---- status / context ---------------------------------------------------------------------------
mutable struct JchCtx
    h::Ptr{Cvoid}
    function JchCtx(device::Integer = 0; stream::Ptr{Cvoid} = C_NULL)
        r = Ref{Ptr{Cvoid}}(C_NULL)
        st = ccall((:jch_ctx_create, LIB), Int32, (Ref{Ptr{Cvoid}}, Int32, Ptr{Cvoid}, UInt32), r, device, stream, 0)
        st == 0 || error("jch_ctx_create: ", unsafe_string(ccall((:jch_last_error, LIB), Cstring, (Ptr{Cvoid},), C_NULL)))
        ctx = new(r[])
        finalizer(c -> ccall((:jch_ctx_destroy, LIB), Int32, (Ptr{Cvoid},), c.h), ctx)
        ctx
    end
end

check(ctx::JchCtx, st::Integer) =
    st == 0 || error("libjchemo_hip error $st: ", unsafe_string(ccall((:jch_last_error, LIB), Cstring, (Ptr{Cvoid},), ctx.h)))

const _default = Ref{Union{Nothing, JchCtx}}(nothing)
default_ctx() = (_default[] === nothing && (_default[] = JchCtx(0)); _default[])

"Join this process' context to a row-sharded multi-GPU fit (one Julia process per GPU; `uid` from rank 0's
`unique_id()`, exchanged with MPI.jl / Distributed)."
function unique_id()
    b = zeros(UInt8, 128)
    ccall((:jch_comm_unique_id, LIB), Int32, (Ptr{Cvoid},), b) == 0 || error("jch_comm_unique_id: RCCL not loadable")
    b
end
comm_init!(ctx::JchCtx, uid::Vector{UInt8}, rank::Integer, nranks::Integer) =
    check(ctx, ccall((:jch_ctx_comm_init, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Int32, Int32), ctx.h, uid, rank, nranks))

# ---- the reference package, when it is loaded --------------------------------------------------------
const _JCHEMO_ID = Base.PkgId(Base.UUID("fbca9394-dd0a-4d1c-b066-ae75f6ef1ad5"), "Jchemo")   # Project.toml:1-2 of the reference
const _jchemo = Ref{Union{Nothing, Module}}(nothing)

"The loaded `Jchemo` module, or `nothing`.  Found through its package id, so the load order does not matter."
function jchemo_module()
    if _jchemo[] === nothing
        m = get(Base.loaded_modules, _JCHEMO_ID, nothing)
        m === nothing || attach!(m)
    end
    _jchemo[]
end

# ---- fallback result record: field names / shapes of Jchemo.Plsr (src/plskern.jl:1-14) ---------------
struct Plsr{TT_, WT}
    T::TT_                      # n x nlv   (Matrix{Float64}, or ROCArray for device-resident fits)
    P::Matrix{Float64}
    R::Matrix{Float64}
    W::Matrix{Float64}
    C::Matrix{Float64}
    TT::Vector{Float64}
    xmeans::Vector{Float64}
    xscales::Vector{Float64}
    ymeans::Vector{Float64}
    yscales::Vector{Float64}
    weights::WT
    niter::Union{Array{Float64}, Nothing}
end

"""
    attach!(Jchemo)

Remember the reference module (fits on host arrays then return `Jchemo.Plsr` / `Jchemo.Lwplsr`) and give the
reference's generics methods for the fallback record, so `Jchemo.predict(fm, X)` also works on a device-resident fit.
"""
function attach!(J::Module)
    _jchemo[] === J && return J
    _jchemo[] = J
    Core.eval(J, :(transform(object::$Plsr, X; nlv = nothing) = $transform(object, X; nlv = nlv)))
    Core.eval(J, :(coef(object::$Plsr; nlv = nothing) = $coef(object; nlv = nlv)))
    Core.eval(J, :(predict(object::$Plsr, X; nlv = nothing) = $predict(object, X; nlv = nlv)))
    J
end

# the record a fit hands back (see the module docstring)
function _record(T, P, R, W, C, TT, xm, xs, ym, ys, wn, niter)
    J = jchemo_module()
    if J !== nothing && T isa Matrix{Float64} && wn isa Vector{Float64}
        return Base.invokelatest(getfield(J, :Plsr), T, P, R, W, C, TT, xm, xs, ym, ys, wn, niter)
    end
    Plsr(T, P, R, W, C, TT, xm, xs, ym, ys, wn, niter)
end

struct PlsDesc                                          # == jch_pls_desc (include/jchemo_hip.h)
    n::Int64; p::Int64; q::Int64
    nlv::Int32; scal::Int32; dtype::Int32; loc::Int32; inplace::Int32; reserved::Int32
end

ensure_mat(X::AbstractMatrix) = X                       # src/utility.jl:544-548
ensure_mat(X::AbstractVector) = reshape(X, :, 1)
ensure_mat(X::Number) = reshape([X], 1, 1)

# Host arrays: loc = 0.  Device arrays (AMDGPU.ROCArray{Float64,2}): loc = 1; `pointer(A)` is the device address.
_loc(::Array) = Int32(0)
_loc(A) = Int32(1)                                      # any other strided column-major device array type
_similar(A::Array, dims...) = Array{Float64}(undef, dims...)
_similar(A, dims...) = similar(A, Float64, dims...)
_f64(A::Array{Float64}) = A
_f64(A::Array) = Float64.(A)
_f64(A) = A                                             # device arrays are taken as they are (Float64 required)
"`v` as a vector living where `like` lives (host Vector / device array of `like`'s type)"
_colocate(v, like::Array) = v isa Vector{Float64} ? v : Vector{Float64}(Array(v))
_colocate(v, like) = v isa Array ? copyto!(_similar(like, length(v)), vec(Float64.(v))) : v

# One method per entry point, each with its LITERAL argument-type tuple (`ccall` needs the tuple spelled out where it
# is called; a constant bound to the tuple is a lowering error).  jch_plskern_fit and its same-signature siblings:
for alg in (:plskern, :plsnipals, :plssimp, :plsrosa)
    cname = QuoteNode(Symbol(:jch_, alg, :_fit))
    @eval _fit_call(::Val{$(QuoteNode(alg))}, h, desc, X, ldx, Y, ldy, w, T, P, R, W, C, TT, xm, xs, ym, ys, wn, got) =
        ccall(($cname, LIB), Int32,
              (Ptr{Cvoid}, Ref{PlsDesc}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Float64},
               Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
               Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{Int32}),
              h, desc, X, ldx, Y, ldy, w, T, P, R, W, C, TT, xm, xs, ym, ys, wn, got)
end

# sym: :plskern | :plsnipals | :plssimp | :plsrosa (one C signature) or :plswold (tol, maxit, niter in addition)
function _fit(sym::Symbol, X, Y, weights, nlv, scal, inplace, ctx::JchCtx; tol = sqrt(eps(1.)), maxit = 200)
    n, p = size(X); q = size(Y, 2)
    size(Y, 1) == n || throw(DimensionMismatch("X has $n rows, Y has $(size(Y, 1))"))
    # min(n, p, nlv) as src/plskern.jl:116 — exactly the columns the library fills on one GPU, so T is handed back as
    # allocated (no n x nlv copy); a row shard smaller than nlv of a multi-GPU fit may come back with more (k > n)
    kmax = max(1, min(p, nlv, ctx_nranks(ctx) > 1 ? typemax(Int) : n))
    T = _similar(X, n, kmax); wn = _similar(X, n)
    P = zeros(p, kmax); R = zeros(p, kmax); W = zeros(p, kmax); C = zeros(q, kmax); TT = zeros(kmax)
    xm = zeros(p); xs = zeros(p); ym = zeros(q); ys = zeros(q); niter = zeros(kmax)
    desc = Ref(PlsDesc(n, p, q, nlv, scal ? 1 : 0, 0, _loc(X), inplace ? 1 : 0, 0))
    got = Ref{Int32}(0)
    GC.@preserve X Y weights T wn begin
        w = weights === nothing ? Ptr{Float64}(C_NULL) : pointer(weights)
        st = if sym === :plswold
            ccall((:jch_plswold_fit, LIB), Int32,
                  (Ptr{Cvoid}, Ref{PlsDesc}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Int64, Ptr{Float64}, Float64, Int32, Ptr{Float64},
                   Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
                   Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{Int32}),
                  ctx.h, desc, pointer(X), stride(X, 2), pointer(Y), max(stride(Y, 2), n), w, tol, maxit, pointer(T),
                  P, R, W, C, TT, xm, xs, ym, ys, pointer(wn), niter, got)
        else
            _fit_call(Val(sym), ctx.h, desc, pointer(X), stride(X, 2), pointer(Y), max(stride(Y, 2), n), w, pointer(T),
                      P, R, W, C, TT, xm, xs, ym, ys, pointer(wn), got)
        end
        check(ctx, st)
    end
    k = Int(got[])
    cut(A) = k == size(A, 2) ? A : A[:, 1:k]
    _record(cut(T), cut(P), cut(R), cut(W), cut(C), k == length(TT) ? TT : TT[1:k], xm, xs, ym, ys, wn,
            sym === :plswold ? (k == length(niter) ? niter : niter[1:k]) : nothing)
end

function ctx_nranks(ctx::JchCtx)
    r = Ref{Int32}(0); nr = Ref{Int32}(1)
    ccall((:jch_ctx_comm_info, LIB), Int32, (Ptr{Cvoid}, Ref{Int32}, Ref{Int32}), ctx.h, r, nr)
    Int(nr[])
end

# weights where X lives (`nothing` = ones(n), as the reference's default argument)
_w(weights, X) = weights === nothing ? nothing : _colocate(vec(weights), X)
# non-`!` variants: any real matrix / vector / DataFrame-free input; the library never writes the inputs (inplace = 0),
# so the reference's `copy` (src/plskern.jl:108) is not needed
_in(X) = _f64(ensure_mat(X))

"`plskern(X, Y, weights = ones(n); nlv, scal = false)` — src/plskern.jl:106-110 (inputs untouched)."
plskern(X, Y, weights = nothing; nlv, scal = false, ctx = default_ctx()) =
    _fit(:plskern, _in(X), _in(Y), _w(weights, _in(X)), nlv, scal, false, ctx)
"`plskern!(X::Matrix, Y::Matrix, ...)` — src/plskern.jl:112-178: X, Y are overwritten (centred/scaled)."
plskern!(X, Y, weights = nothing; nlv, scal = false, ctx = default_ctx()) =
    _fit(:plskern, X, Y, _w(weights, X), nlv, scal, true, ctx)
"`plsnipals` — src/plsnipals.jl:31-35."
plsnipals(X, Y, weights = nothing; nlv, scal = false, ctx = default_ctx()) =
    _fit(:plsnipals, _in(X), _in(Y), _w(weights, _in(X)), nlv, scal, false, ctx)
"`plsnipals!` — src/plsnipals.jl:37-97: X, Y end up centred/scaled and deflated."
plsnipals!(X, Y, weights = nothing; nlv, scal = false, ctx = default_ctx()) =
    _fit(:plsnipals, X, Y, _w(weights, X), nlv, scal, true, ctx)

# Sibling algorithms (same row kernels, different small state; include/jchemo_hip.h)
"`plssimp` — src/plssimp.jl:22-26 (`W` is returned equal to `R`, :85-87)."
plssimp(X, Y, weights = nothing; nlv, scal = false, ctx = default_ctx()) =
    _fit(:plssimp, _in(X), _in(Y), _w(weights, _in(X)), nlv, scal, false, ctx)
"`plssimp!` — src/plssimp.jl:28-88."
plssimp!(X, Y, weights = nothing; nlv, scal = false, ctx = default_ctx()) =
    _fit(:plssimp, X, Y, _w(weights, X), nlv, scal, true, ctx)
"`plsrosa` — src/plsrosa.jl:26-30."
plsrosa(X, Y, weights = nothing; nlv, scal = false, ctx = default_ctx()) =
    _fit(:plsrosa, _in(X), _in(Y), _w(weights, _in(X)), nlv, scal, false, ctx)
"`plsrosa!` — src/plsrosa.jl:32-96: X centred/scaled, Y centred/scaled and deflated."
plsrosa!(X, Y, weights = nothing; nlv, scal = false, ctx = default_ctx()) =
    _fit(:plsrosa, X, Y, _w(weights, X), nlv, scal, true, ctx)
"`plswold` — src/plswold.jl:30-34; `niter` filled as :93."
plswold(X, Y, weights = nothing; nlv, tol = sqrt(eps(1.)), maxit = 200, scal = false, ctx = default_ctx()) =
    _fit(:plswold, _in(X), _in(Y), _w(weights, _in(X)), nlv, scal, false, ctx; tol = tol, maxit = maxit)
"`plswold!` — src/plswold.jl:36-111."
plswold!(X, Y, weights = nothing; nlv, tol = sqrt(eps(1.)), maxit = 200, scal = false, ctx = default_ctx()) =
    _fit(:plswold, X, Y, _w(weights, X), nlv, scal, true, ctx; tol = tol, maxit = maxit)

# out = ((X - 1*shift') ./ scale') * B .+ bias'   (shift, scale, B, bias on the host; X and out where X lives)
function _affine(X, shift, scale, B::Matrix{Float64}, bias, ctx)
    X = _in(X); m, p = size(X); k = size(B, 2)
    size(B, 1) == p || throw(DimensionMismatch("X has $p columns, the model has $(size(B, 1))"))
    out = _similar(X, m, k)
    GC.@preserve X out shift scale bias begin
        check(ctx, ccall((:jch_affine_gemm, LIB), Int32,
                         (Ptr{Cvoid}, Int32, Ptr{Float64}, Int64, Int64, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
                          Int64, Ptr{Float64}, Ptr{Float64}, Int64),
                         ctx.h, _loc(X), pointer(X), m, p, stride(X, 2),
                         shift === nothing ? Ptr{Float64}(C_NULL) : pointer(shift),
                         scale === nothing ? Ptr{Float64}(C_NULL) : pointer(scale), B, k,
                         bias === nothing ? Ptr{Float64}(C_NULL) : pointer(bias), pointer(out), m))
    end
    out
end

_nlv_fit(object) = size(object.P, 2)     # (== nco(object.T) of the reference; P is always a host matrix)

"`transform(object, X; nlv)` — src/plskern.jl:187-195 on the GPU; `object`: `Jchemo.Plsr` or `JchemoHIP.Plsr`."
function transform(object, X; nlv = nothing, ctx = default_ctx())
    a = _nlv_fit(object)
    nlv = nlv === nothing ? a : min(nlv, a)
    _affine(X, object.xmeans, object.xscales, nlv == a ? object.R : object.R[:, 1:nlv], nothing, ctx)
end

"`coef(object; nlv)` — src/plskern.jl:207-217 (p x q host glue, as in the reference)"
function coef(object; nlv = nothing)
    a = _nlv_fit(object)
    nlv = nlv === nothing ? a : min(nlv, a)
    beta = object.C[:, 1:nlv]'
    B = Diagonal(1 ./ object.xscales) * object.R[:, 1:nlv] * beta * Diagonal(object.yscales)
    int = object.ymeans' .- object.xmeans' * B
    (B = B, int = int)
end

"""`predict(object, X; nlv)` — src/plskern.jl:226-238 (a `Plsr` record) or src/lwplsr.jl:134-166 (an `Lwplsr` record).
PLSR: the whole nlv range in ONE pass over X (the B blocks concatenated)."""
function predict(object, X; nlv = nothing, ctx = default_ctx())
    hasproperty(object, :metric) && return _predict_lwplsr(object, X, nlv, ctx)
    a = _nlv_fit(object); q = size(object.C, 1)
    rng = nlv === nothing ? (a:a) : (max(0, minimum(nlv)):min(a, maximum(nlv)))
    zs = [coef(object; nlv = k) for k in rng]
    out = _affine(X, nothing, nothing, reduce(hcat, [z.B for z in zs]), reduce(vcat, [vec(z.int) for z in zs]), ctx)
    pred = [out[:, (i - 1) * q + 1:i * q] for i in 1:length(rng)]
    (pred = length(rng) == 1 ? pred[1] : pred,)
end

"""
    explvarx(object, X)

The table of `summary(object::Plsr, X)` (src/plskern.jl:246-260) as a column table `(nlv, var, pvar, cumpvar)` (the
reference wraps the same four columns in a `DataFrame`; `DataFrame(explvarx(fm, X))` gives exactly that), with
`sstot` computed on the GPU.  `object`: either record; X: the data the model was fitted on.
"""
function explvarx(object, X; ctx = default_ctx())
    X = _in(X); n, nlv = size(X, 1), _nlv_fit(object)
    d = _colocate(object.weights, X)          # the (normalised) weights where X lives
    ss = Ref{Float64}(0.0)
    GC.@preserve X d begin
        check(ctx, ccall((:jch_weighted_ss, LIB), Int32,
                         (Ptr{Cvoid}, Int32, Ptr{Float64}, Int64, Int64, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{Float64}),
                         ctx.h, _loc(X), pointer(X), n, size(X, 2), stride(X, 2), pointer(d),
                         object.xmeans, object.xscales, ss))
    end
    tt_adj = vec(sum(object.P .^ 2, dims = 1)) .* object.TT
    pvar = tt_adj / ss[]
    (nlv = collect(1:nlv), var = tt_adj / n, pvar = pvar, cumpvar = cumsum(pvar))
end

"`summary(object::JchemoHIP.Plsr, X)` — src/plskern.jl:246-260: `(explvarx = table,)`.  (A `Jchemo.Plsr` returned by a
fit has the reference's own `summary` method; `explvarx` is the GPU version for either record.)"
Base.summary(object::Plsr, X; ctx = default_ctx()) = (explvarx = explvarx(object, X; ctx = ctx),)

# ---- kNN-LWPLSR (src/lwplsr.jl) -------------------------------------------------------------------
struct Lwplsr                     # fallback record, same fields as the reference's struct (src/lwplsr.jl:1-12)
    X; Y; fm; metric::String; h::Real; k::Int; nlv::Int; tol::Real; scal::Bool; verbose::Bool
end

"`lwplsr(X, Y; nlvdis, metric, h, k, nlv, tol = 1e-4, scal = false)` — src/lwplsr.jl:114-126."
function lwplsr(X, Y; nlvdis, metric, h, k, nlv, tol = 1e-4, scal = false, verbose = false, ctx = default_ctx())
    X = _in(X); Y = _in(Y)
    fm = nlvdis == 0 ? nothing : plskern(X, Y; nlv = nlvdis, scal = scal, ctx = ctx)
    J = jchemo_module()
    if J !== nothing && X isa Array{Float64} && Y isa Array{Float64}
        return Base.invokelatest(getfield(J, :Lwplsr), X, Y, fm, metric, h, k, nlv, tol, scal, verbose)
    end
    Lwplsr(X, Y, fm, metric, h, k, nlv, tol, scal, verbose)
end

function _cov(A, ctx)             # Statistics.cov(A, corrected = false) on the device (src/getknn.jl:38)
    n, d = size(A); S = zeros(d, d)
    GC.@preserve A check(ctx, ccall((:jch_weighted_cov, LIB), Int32,
        (Ptr{Cvoid}, Int32, Ptr{Float64}, Int64, Int64, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
        ctx.h, _loc(A), pointer(A), n, d, stride(A, 2), Ptr{Float64}(C_NULL), S, Ptr{Float64}(C_NULL)))
    S
end

# `predict(object::Lwplsr, X; nlv)` — src/lwplsr.jl:134-166: neighbours, weights and the m local fits in one call.
function _predict_lwplsr(object, X, nlv, ctx)
    X = _in(X); m = size(X, 1); n, p = size(object.X); q = size(object.Y, 2)
    a = object.nlv
    rng = nlv === nothing ? (a:a) : (max(minimum(nlv), 0):min(maximum(nlv), a, p))
    # the space the neighbours are searched in (src/lwplsr.jl:139-150)
    if object.fm === nothing
        if object.scal                        # :141-145  scale(object.X, colstd(object.X)) on both sides
            xs = col_stats(object.X; ctx = ctx).stds
            Dinv = Matrix(Diagonal(1 ./ xs))
            Zt = _affine(object.X, nothing, nothing, Dinv, nothing, ctx); Zq = _affine(X, nothing, nothing, Dinv, nothing, ctx)
        else
            Zt, Zq = object.X, X
        end
    else
        Zt, Zq = object.fm.T, transform(object.fm, X; ctx = ctx)
    end
    Zt = _colocate_mat(Zt, X); Zq = _colocate_mat(Zq, X)
    if object.metric == "mahal"               # src/getknn.jl:37-49
        S = _cov(Zt, ctx); d = size(S, 1)
        Uinv = d == 1 ? fill(1 / sqrt(S[1, 1]), 1, 1) : (isposdef(S) ? Matrix(inv(cholesky(Hermitian(S)).U)) : Matrix(Diagonal(1 ./ diag(S))))
        Zt = _affine(Zt, nothing, nothing, Uinv, nothing, ctx); Zq = _affine(Zq, nothing, nothing, Uinv, nothing, ctx)
    end
    k = min(object.k, n); le = length(rng)
    q <= 16 || error("predict(::Lwplsr): the batched kernel handles q <= 16 responses")
    pred = zeros(q, le, m); ind = zeros(Int32, k, m); dist = zeros(k, m); w = zeros(k, m)   # C layout [m][le][q] == Julia (q, le, m)
    Xt = _colocate_mat(object.X, X); Yt = _colocate_mat(object.Y, X)
    GC.@preserve Xt Yt Zt Zq X check(ctx, ccall((:jch_lwplsr_predict, LIB), Int32,
        (Ptr{Cvoid}, Int32, Ptr{Float64}, Int64, Int64, Int64, Ptr{Float64}, Int64, Int64, Ptr{Float64}, Int64, Ptr{Float64}, Int64,
         Int64, Ptr{Float64}, Int64, Int64, Int32, Float64, Float64, Int32, Int32, Int32, Ptr{Float64}, Ptr{Int32}, Ptr{Float64}, Ptr{Float64}),
        ctx.h, _loc(Xt), pointer(Xt), n, p, stride(Xt, 2), pointer(Yt), q, max(stride(Yt, 2), n), pointer(Zt), stride(Zt, 2),
        pointer(Zq), stride(Zq, 2), size(Zt, 2), pointer(X), m, stride(X, 2), k, object.h, object.tol, object.scal ? 1 : 0,
        first(rng), last(rng), pred, ind, dist, w))
    preds = [permutedims(pred[:, i, :]) for i in 1:le]                                      # m x q per nlv
    (pred = le == 1 ? preds[1] : preds, listnn = [Int.(ind[:, i]) .+ 1 for i in 1:m], listd = [dist[:, i] for i in 1:m],
     listw = [w[:, i] for i in 1:m])
end
# every n-sized operand of one call must live on the same side (`loc` is one flag): bring A where `like` lives
_colocate_mat(A::Array, like::Array) = A
_colocate_mat(A, like::Array) = Array(A)
_colocate_mat(A::Array, like) = copyto!(_similar(like, size(A)...), A)
_colocate_mat(A, like) = A

# ---- caller-supplied column scales (multiblock PLSR, src/mbplsr.jl:77-113) and column statistics
"Weighted column means and uncorrected stds from the device (`colmean`, `colstd`: src/utility.jl:193-195,312-323)."
function col_stats(X, weights = nothing; ctx = default_ctx())
    X = _in(X); n, p = size(X); m = zeros(p); s = zeros(p)
    weights = _w(weights, X)
    GC.@preserve X weights check(ctx, ccall((:jch_col_stats, LIB), Int32,
        (Ptr{Cvoid}, Int32, Ptr{Float64}, Int64, Int64, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
        ctx.h, _loc(X), pointer(X), n, p, stride(X, 2), weights === nothing ? Ptr{Float64}(C_NULL) : pointer(weights), m, s))
    (means = m, stds = s)
end

"`plskern` with column divisors handed in (X centred by its weighted means, divided by `xscales`; Y by `yscales`)."
function plskern_scaled(X, Y, xscales::Vector{Float64}, yscales = nothing, weights = nothing; nlv, ctx = default_ctx())
    X = _in(X); Y = _in(Y); n, p = size(X); q = size(Y, 2); kmax = max(1, min(n, p, nlv))
    weights = _w(weights, X)
    T = _similar(X, n, kmax); wn = _similar(X, n)
    P = zeros(p, kmax); R = zeros(p, kmax); W = zeros(p, kmax); C = zeros(q, kmax); TT = zeros(kmax)
    xm = zeros(p); xs = zeros(p); ym = zeros(q); ys = zeros(q); got = Ref{Int32}(0)
    desc = Ref(PlsDesc(n, p, q, nlv, 0, 0, _loc(X), 0, 0))
    GC.@preserve X Y weights yscales T wn check(ctx, ccall((:jch_plskern_fit_scaled, LIB), Int32,
        (Ptr{Cvoid}, Ref{PlsDesc}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
         Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
         Ptr{Float64}, Ptr{Float64}, Ref{Int32}),
        ctx.h, desc, pointer(X), stride(X, 2), pointer(Y), max(stride(Y, 2), n),
        weights === nothing ? Ptr{Float64}(C_NULL) : pointer(weights), xscales,
        yscales === nothing ? Ptr{Float64}(C_NULL) : pointer(yscales), pointer(T), P, R, W, C, TT, xm, xs, ym, ys, pointer(wn), got))
    k = Int(got[])
    cut(A) = k == size(A, 2) ? A : A[:, 1:k]
    _record(cut(T), cut(P), cut(R), cut(W), cut(C), k == length(TT) ? TT : TT[1:k], xm, xs, ym, ys, wn, nothing)
end

# ---- P2P inbox transport (include/jchemo_hip.h): export -> all-gather the handles (MPI) -> import -> agree -> enable
function p2p_export(ctx::JchCtx, nranks::Integer)
    h = zeros(UInt8, 64)
    check(ctx, ccall((:jch_ctx_p2p_export, LIB), Int32, (Ptr{Cvoid}, Int32, Ptr{Cvoid}), ctx.h, nranks, h))
    h
end
p2p_import(ctx::JchCtx, handles::Vector{UInt8}, rank::Integer, nranks::Integer) =
    ccall((:jch_ctx_p2p_import, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Int32, Int32, UInt32), ctx.h, handles, rank, nranks, 0) == 0
p2p_enable!(ctx::JchCtx, on::Bool) = check(ctx, ccall((:jch_ctx_p2p_enable, LIB), Int32, (Ptr{Cvoid}, Int32), ctx.h, on ? 1 : 0))

function __init__()
    ccall(:jl_generating_output, Cint, ()) == 1 && return nothing   # being precompiled into another image: hook up at run time
    m = get(Base.loaded_modules, _JCHEMO_ID, nothing)     # `using Jchemo` came first: hook up now; otherwise on first use
    m === nothing || attach!(m)
    nothing
end

end # module
