"""
    JchemoHIP

Julia host side of the MI355X-native PLS engine: the same call shapes as Jchemo.jl
(`plskern`, `plskern!`, `plsnipals`, `plsnipals!`, `transform`, `coef`, `predict`, `summary` and a
result with the fields of `Jchemo.Plsr`, src/plskern.jl:1-14 of the reference), implemented as thin
`ccall`s into `libjchemo_hip.so` (C ABI: include/jchemo_hip.h).  AMDGPU.jl is used only as a handle
for device buffers (`ROCArray`); there is no CUDA path and no CPU fallback.

NOT EXECUTED IN THIS REPOSITORY'S CI: the build image has no Julia toolchain (see DESIGN.md).  Every
behaviour below is exercised through the identical C entry points by the ctypes mirror in
`jchemo.jl_amd/jchemo_hip/` (tests/test_gpu_parity.py).
"""
module JchemoHIP

using LinearAlgebra
using Libdl

export Plsr, Lwplsr, plskern, plskern!, plsnipals, plsnipals!, plssimp, plssimp!, plsrosa, plsrosa!, plswold, plswold!, lwplsr, transform, coef, predict, summary_plsr, JchCtx

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
unique_id() = (b = zeros(UInt8, 128); ccall((:jch_comm_unique_id, LIB), Int32, (Ptr{UInt8},), b) == 0 || error("rccl"); b)
comm_init!(ctx::JchCtx, uid::Vector{UInt8}, rank::Integer, nranks::Integer) =
    check(ctx, ccall((:jch_ctx_comm_init, LIB), Int32, (Ptr{Cvoid}, Ptr{UInt8}, Int32, Int32), ctx.h, uid, rank, nranks))

# ---- result record: same field names / shapes as Jchemo.Plsr (src/plskern.jl:1-14) -----------------
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

struct PlsDesc
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

const _FIT_SIG = (Ptr{Cvoid}, Ref{PlsDesc}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Float64},
                  Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
                  Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{Int32})
_entry(name::Symbol) = Libdl.dlsym(Libdl.dlopen(LIB), name)     # jch_plskern_fit and its same-signature siblings

# sym: :plskern | :plsnipals | :plssimp | :plsrosa (one C signature) or :plswold (tol, maxit, niter in addition)
function _fit(sym::Symbol, X, Y, weights, nlv, scal, inplace, ctx::JchCtx; tol = sqrt(eps(1.)), maxit = 200)
    n, p = size(X); q = size(Y, 2)
    size(Y, 1) == n || throw(DimensionMismatch("X has $n rows, Y has $(size(Y, 1))"))
    kmax = max(1, min(p, nlv))
    T = _similar(X, n, kmax); wn = _similar(X, n)
    P = zeros(p, kmax); R = zeros(p, kmax); W = zeros(p, kmax); C = zeros(q, kmax); TT = zeros(kmax)
    xm = zeros(p); xs = zeros(p); ym = zeros(q); ys = zeros(q); niter = zeros(kmax)
    desc = Ref(PlsDesc(n, p, q, nlv, scal ? 1 : 0, 0, _loc(X), inplace ? 1 : 0, 0))
    got = Ref{Int32}(0)
    w = weights === nothing ? C_NULL : pointer(weights)
    GC.@preserve X Y weights T wn begin
        st = if sym === :plswold
            ccall((:jch_plswold_fit, LIB), Int32,
                  (Ptr{Cvoid}, Ref{PlsDesc}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Int64, Ptr{Float64}, Float64, Int32, Ptr{Float64},
                   Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
                   Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{Int32}),
                  ctx.h, desc, pointer(X), stride(X, 2), pointer(Y), max(stride(Y, 2), n), w, tol, maxit, pointer(T),
                  P, R, W, C, TT, xm, xs, ym, ys, pointer(wn), niter, got)
        else
            ccall(_entry(Symbol(:jch_, sym, :_fit)), Int32, _FIT_SIG,
                  ctx.h, desc, pointer(X), stride(X, 2), pointer(Y), max(stride(Y, 2), n), w, pointer(T),
                  P, R, W, C, TT, xm, xs, ym, ys, pointer(wn), got)
        end
        check(ctx, st)
    end
    k = Int(got[])
    Plsr(T[:, 1:k], P[:, 1:k], R[:, 1:k], W[:, 1:k], C[:, 1:k], TT[1:k], xm, xs, ym, ys, wn,
         sym === :plswold ? niter[1:k] : nothing)
end

_w(weights, X) = weights === nothing ? nothing : convert(typeof(_similar(X, 0)), vec(Float64.(weights)))

"`plskern(X, Y, weights = ones(n); nlv, scal = false)` — src/plskern.jl:106-110 (inputs untouched)."
plskern(X, Y, weights = nothing; nlv, scal = false, ctx = default_ctx()) =
    _fit(:plskern, ensure_mat(X), ensure_mat(Y), _w(weights, ensure_mat(X)), nlv, scal, false, ctx)
"`plskern!(X::Matrix, Y::Matrix, ...)` — src/plskern.jl:112-178: X, Y are overwritten (centred/scaled)."
plskern!(X, Y, weights = nothing; nlv, scal = false, ctx = default_ctx()) =
    _fit(:plskern, X, Y, _w(weights, X), nlv, scal, true, ctx)
"`plsnipals` — src/plsnipals.jl:31-35."
plsnipals(X, Y, weights = nothing; nlv, scal = false, ctx = default_ctx()) =
    _fit(:plsnipals, ensure_mat(X), ensure_mat(Y), _w(weights, ensure_mat(X)), nlv, scal, false, ctx)
"`plsnipals!` — src/plsnipals.jl:37-97: X, Y end up centred/scaled and deflated."
plsnipals!(X, Y, weights = nothing; nlv, scal = false, ctx = default_ctx()) =
    _fit(:plsnipals, X, Y, _w(weights, X), nlv, scal, true, ctx)

# Sibling algorithms (same row kernels, different small state; include/jchemo_hip.h)
"`plssimp` — src/plssimp.jl:22-26 (`W` is returned equal to `R`, :85-87)."
plssimp(X, Y, weights = nothing; nlv, scal = false, ctx = default_ctx()) =
    _fit(:plssimp, ensure_mat(X), ensure_mat(Y), _w(weights, ensure_mat(X)), nlv, scal, false, ctx)
"`plssimp!` — src/plssimp.jl:28-88."
plssimp!(X, Y, weights = nothing; nlv, scal = false, ctx = default_ctx()) =
    _fit(:plssimp, X, Y, _w(weights, X), nlv, scal, true, ctx)
"`plsrosa` — src/plsrosa.jl:26-30."
plsrosa(X, Y, weights = nothing; nlv, scal = false, ctx = default_ctx()) =
    _fit(:plsrosa, ensure_mat(X), ensure_mat(Y), _w(weights, ensure_mat(X)), nlv, scal, false, ctx)
"`plsrosa!` — src/plsrosa.jl:32-96: X centred/scaled, Y centred/scaled and deflated."
plsrosa!(X, Y, weights = nothing; nlv, scal = false, ctx = default_ctx()) =
    _fit(:plsrosa, X, Y, _w(weights, X), nlv, scal, true, ctx)
"`plswold` — src/plswold.jl:30-34; `niter` filled as :93."
plswold(X, Y, weights = nothing; nlv, tol = sqrt(eps(1.)), maxit = 200, scal = false, ctx = default_ctx()) =
    _fit(:plswold, ensure_mat(X), ensure_mat(Y), _w(weights, ensure_mat(X)), nlv, scal, false, ctx; tol = tol, maxit = maxit)
"`plswold!` — src/plswold.jl:36-111."
plswold!(X, Y, weights = nothing; nlv, tol = sqrt(eps(1.)), maxit = 200, scal = false, ctx = default_ctx()) =
    _fit(:plswold, X, Y, _w(weights, X), nlv, scal, true, ctx; tol = tol, maxit = maxit)

function _affine(X, shift, scale, B::Matrix{Float64}, bias, ctx)
    X = ensure_mat(X); m, p = size(X); k = size(B, 2)
    size(B, 1) == p || throw(DimensionMismatch("X has $p columns, the model has $(size(B, 1))"))
    out = _similar(X, m, k)
    GC.@preserve X out begin
        check(ctx, ccall((:jch_affine_gemm, LIB), Int32,
                         (Ptr{Cvoid}, Int32, Ptr{Float64}, Int64, Int64, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
                          Int64, Ptr{Float64}, Ptr{Float64}, Int64),
                         ctx.h, _loc(X), pointer(X), m, p, stride(X, 2), shift === nothing ? C_NULL : pointer(shift),
                         scale === nothing ? C_NULL : pointer(scale), B, k, bias === nothing ? C_NULL : pointer(bias),
                         pointer(out), m))
    end
    out
end

"src/plskern.jl:187-195"
function transform(object::Plsr, X; nlv = nothing, ctx = default_ctx())
    a = size(object.P, 2)
    nlv = nlv === nothing ? a : min(nlv, a)
    _affine(X, object.xmeans, object.xscales, object.R[:, 1:nlv], nothing, ctx)
end

"src/plskern.jl:207-217 (p x q host glue, as in the reference)"
function coef(object::Plsr; nlv = nothing)
    a = size(object.P, 2)
    nlv = nlv === nothing ? a : min(nlv, a)
    beta = object.C[:, 1:nlv]'
    B = Diagonal(1 ./ object.xscales) * object.R[:, 1:nlv] * beta * Diagonal(object.yscales)
    int = object.ymeans' .- object.xmeans' * B
    (B = B, int = int)
end

"src/plskern.jl:226-238 — the whole nlv range in ONE pass over X (B blocks concatenated)."
function predict(object::Plsr, X; nlv = nothing, ctx = default_ctx())
    a = size(object.P, 2); q = size(object.C, 1)
    rng = nlv === nothing ? (a:a) : (max(0, minimum(nlv)):min(a, maximum(nlv)))
    zs = [coef(object; nlv = k) for k in rng]
    out = _affine(X, nothing, nothing, reduce(hcat, [z.B for z in zs]), reduce(vcat, [vec(z.int) for z in zs]), ctx)
    pred = [out[:, (i - 1) * q + 1:i * q] for i in 1:length(rng)]
    (pred = length(rng) == 1 ? pred[1] : pred,)
end

"`summary(object::Plsr, X)` — src/plskern.jl:246-260 (named summary_plsr to avoid piracy on Base.summary)."
function summary_plsr(object::Plsr, X; ctx = default_ctx())
    X = ensure_mat(X); n, nlv = size(X, 1), size(object.P, 2)
    ss = Ref{Float64}(0.0)
    GC.@preserve X begin
        check(ctx, ccall((:jch_weighted_ss, LIB), Int32,
                         (Ptr{Cvoid}, Int32, Ptr{Float64}, Int64, Int64, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{Float64}),
                         ctx.h, _loc(X), pointer(X), n, size(X, 2), stride(X, 2), pointer(object.weights),
                         object.xmeans, object.xscales, ss))
    end
    tt_adj = vec(sum(object.P .^ 2, dims = 1)) .* object.TT
    pvar = tt_adj / ss[]
    (explvarx = (nlv = 1:nlv, var = tt_adj / n, pvar = pvar, cumpvar = cumsum(pvar)),)
end

# ---- kNN-LWPLSR (src/lwplsr.jl) -------------------------------------------------------------------
struct Lwplsr                     # same fields as the reference's struct (src/lwplsr.jl:1-12)
    X; Y; fm; metric::String; h::Real; k::Int; nlv::Int; tol::Real; scal::Bool; verbose::Bool
end

"`lwplsr(X, Y; nlvdis, metric, h, k, nlv, tol = 1e-4, scal = false)` — src/lwplsr.jl:114-126."
function lwplsr(X, Y; nlvdis, metric, h, k, nlv, tol = 1e-4, scal = false, verbose = false, ctx = default_ctx())
    X = ensure_mat(X); Y = ensure_mat(Y)
    fm = nlvdis == 0 ? nothing : plskern(X, Y; nlv = nlvdis, scal = scal, ctx = ctx)
    Lwplsr(X, Y, fm, metric, h, k, nlv, tol, scal, verbose)
end

function _cov(A, ctx)             # Statistics.cov(A, corrected = false) on the device (src/getknn.jl:38)
    n, d = size(A); S = zeros(d, d)
    GC.@preserve A check(ctx, ccall((:jch_weighted_cov, LIB), Int32,
        (Ptr{Cvoid}, Int32, Ptr{Float64}, Int64, Int64, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
        ctx.h, _loc(A), pointer(A), n, d, stride(A, 2), C_NULL, S, C_NULL))
    S
end

"`predict(object::Lwplsr, X; nlv)` — src/lwplsr.jl:134-166: neighbours, weights and the m local fits in one call."
function predict(object::Lwplsr, X; nlv = nothing, ctx = default_ctx())
    X = ensure_mat(X); m = size(X, 1); n, p = size(object.X); q = size(object.Y, 2)
    a = object.nlv
    rng = nlv === nothing ? (a:a) : (max(minimum(nlv), 0):min(maximum(nlv), a, p))
    Zt, Zq = object.fm === nothing ? (object.X, X) : (object.fm.T, transform(object.fm, X; ctx = ctx))
    if object.metric == "mahal"
        S = _cov(Zt, ctx); d = size(S, 1)
        Uinv = d == 1 ? fill(1 / sqrt(S[1, 1]), 1, 1) : (isposdef(S) ? Matrix(inv(cholesky(Hermitian(S)).U)) : Matrix(Diagonal(1 ./ diag(S))))
        Zt = _affine(Zt, nothing, nothing, Uinv, nothing, ctx); Zq = _affine(Zq, nothing, nothing, Uinv, nothing, ctx)
    end
    k = min(object.k, n); le = length(rng)
    q <= 8 || error("predict(::Lwplsr): the batched kernel handles q <= 8 responses")
    pred = zeros(q, le, m); ind = zeros(Int32, k, m); dist = zeros(k, m); w = zeros(k, m)   # C layout [m][le][q] == Julia (q, le, m)
    Xt = object.X; Yt = object.Y
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

# ---- caller-supplied column scales (multiblock PLSR, src/mbplsr.jl:77-113) and column statistics
"Weighted column means and uncorrected stds from the device (`colmean`, `colstd`: src/utility.jl:193-195,312-323)."
function col_stats(X, weights = nothing; ctx = default_ctx())
    X = ensure_mat(X); n, p = size(X); m = zeros(p); s = zeros(p)
    w = weights === nothing ? C_NULL : pointer(weights)
    GC.@preserve X weights check(ctx, ccall((:jch_col_stats, LIB), Int32,
        (Ptr{Cvoid}, Int32, Ptr{Float64}, Int64, Int64, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
        ctx.h, _loc(X), pointer(X), n, p, stride(X, 2), w, m, s))
    (means = m, stds = s)
end

"`plskern` with column divisors handed in (X centred by its weighted means, divided by `xscales`; Y by `yscales`)."
function plskern_scaled(X, Y, xscales::Vector{Float64}, yscales = nothing, weights = nothing; nlv, ctx = default_ctx())
    X = ensure_mat(X); Y = ensure_mat(Y); n, p = size(X); q = size(Y, 2); kmax = max(1, min(p, nlv))
    T = _similar(X, n, kmax); wn = _similar(X, n)
    P = zeros(p, kmax); R = zeros(p, kmax); W = zeros(p, kmax); C = zeros(q, kmax); TT = zeros(kmax)
    xm = zeros(p); xs = zeros(p); ym = zeros(q); ys = zeros(q); got = Ref{Int32}(0)
    desc = Ref(PlsDesc(n, p, q, nlv, 0, 0, _loc(X), 0, 0))
    w = weights === nothing ? C_NULL : pointer(weights)
    GC.@preserve X Y weights T wn check(ctx, ccall((:jch_plskern_fit_scaled, LIB), Int32,
        (Ptr{Cvoid}, Ref{PlsDesc}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
         Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
         Ptr{Float64}, Ptr{Float64}, Ref{Int32}),
        ctx.h, desc, pointer(X), stride(X, 2), pointer(Y), max(stride(Y, 2), n), w, xscales,
        yscales === nothing ? C_NULL : pointer(yscales), pointer(T), P, R, W, C, TT, xm, xs, ym, ys, pointer(wn), got))
    k = Int(got[])
    Plsr(T[:, 1:k], P[:, 1:k], R[:, 1:k], W[:, 1:k], C[:, 1:k], TT[1:k], xm, xs, ym, ys, wn, nothing)
end

# ---- P2P inbox transport (include/jchemo_hip.h): export -> all-gather the handles (MPI) -> import -> agree -> enable
p2p_export(ctx::JchCtx, nranks::Integer) = (h = zeros(UInt8, 64); check(ctx, ccall((:jch_ctx_p2p_export, LIB), Int32,
    (Ptr{Cvoid}, Int32, Ptr{UInt8}), ctx.h, nranks, h)); h)
p2p_import(ctx::JchCtx, handles::Vector{UInt8}, rank::Integer, nranks::Integer) =
    ccall((:jch_ctx_p2p_import, LIB), Int32, (Ptr{Cvoid}, Ptr{UInt8}, Int32, Int32, UInt32), ctx.h, handles, rank, nranks, 0) == 0
p2p_enable!(ctx::JchCtx, on::Bool) = check(ctx, ccall((:jch_ctx_p2p_enable, LIB), Int32, (Ptr{Cvoid}, Int32), ctx.h, on ? 1 : 0))

end # module
