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
       lwplsr, transform, coef, predict, explvarx, JchCtx, attach!, nipals_one_pass!,
       msep, rmsep, ssr, bias, r2, cor2, mpar, segmkf, segmts, gridscorelv, gridcvlv,
       Plsrda, dummy, plsrda, Mbplsr, mbplsr, vip, xfit, xresid

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

# ---- tables: the reference returns DataFrames (gridscore.jl:196-220, gridcv.jl:211-227, plskern.jl:258-259) --------------------
const _DATAFRAMES_ID = Base.PkgId(Base.UUID("a93c6f00-e57d-5684-b7b6-d8193f3e46c0"), "DataFrames")   # Project.toml:8 of the reference
"The loaded `DataFrames` module (a dependency of Jchemo, so present whenever Jchemo is), or `nothing`."
dataframes_module() = get(Base.loaded_modules, _DATAFRAMES_ID, nothing)

"""
`_table(cols)`: `cols` is a NamedTuple of equal-length column vectors IN THE REFERENCE'S COLUMN ORDER.  With DataFrames loaded the
result is `DataFrame(cols)` — the type the reference returns —, otherwise the NamedTuple itself (a Tables.jl column table with the
same column names, so `DataFrame(t)` gives the reference's table).
"""
function _table(cols::NamedTuple)
    D = dataframes_module()
    D === nothing ? cols : Base.invokelatest(getfield(D, :DataFrame), cols)
end
_ynames(q) = Tuple(Symbol("y", i) for i in 1:q)                     # `namy = map(string, repeat(["y"], q), 1:q)`

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
function _fit(sym::Symbol, X, Y, weights, nlv, scal, inplace, ctx::JchCtx; tol = sqrt(eps(1.)), maxit = 200, options = 0)
    n, p = size(X); q = size(Y, 2)
    size(Y, 1) == n || throw(DimensionMismatch("X has $n rows, Y has $(size(Y, 1))"))
    # min(n, p, nlv) as src/plskern.jl:116 — exactly the columns the library fills on one GPU, so T is handed back as
    # allocated (no n x nlv copy); a row shard smaller than nlv of a multi-GPU fit may come back with more (k > n)
    kmax = max(1, min(p, nlv, ctx_nranks(ctx) > 1 ? typemax(Int) : n))
    T = _similar(X, n, kmax); wn = _similar(X, n)
    P = zeros(p, kmax); R = zeros(p, kmax); W = zeros(p, kmax); C = zeros(q, kmax); TT = zeros(kmax)
    xm = zeros(p); xs = zeros(p); ym = zeros(q); ys = zeros(q); niter = zeros(kmax)
    desc = Ref(PlsDesc(n, p, q, nlv, scal ? 1 : 0, 0, _loc(X), inplace ? 1 : 0, options))
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

# Diagnostic counters of a ctx (include/jchemo_hip.h): 0 raw-mode fits repeated on the centred copy, 1 kNN-LWPLSR queries refitted per
# query after the neighbour-space kernel's pivot check, 2 queries whose neighbours the screened search found, 3 those of them the
# exact scan redid (the results do not depend on it; ENV["JCH_KNN_SCREEN"] = "0" selects the exact scan for every query)
const COUNTER_PIVOT_REFITS = Int32(0); const COUNTER_LOCW_REFITS = Int32(1)
const COUNTER_KNN_SCREENED = Int32(2); const COUNTER_KNN_SCREEN_REDONE = Int32(3)
function counter(ctx::JchCtx, which::Integer)
    v = Ref{Int64}(0)
    check(ctx, ccall((:jch_ctx_get_counter, LIB), Int32, (Ptr{Cvoid}, Int32, Ref{Int64}), ctx.h, Int32(which), v))
    Int(v[])
end

# weights where X lives (`nothing` = ones(n), as the reference's default argument)
_w(weights, X) = weights === nothing ? nothing : _colocate(vec(weights), X)
# non-`!` variants: any real matrix / vector / DataFrame-free input; the library never writes the inputs (inplace = 0),
# so the reference's `copy` (src/plskern.jl:108) is not needed
_in(X) = _f64(ensure_mat(X))

"`plskern(X, Y, weights = ones(n); nlv, scal = false)` — src/plskern.jl:106-110 (inputs untouched)."
plskern(X, Y, weights = nothing; nlv, scal = false, ctx = default_ctx()) =
    _fit(:plskern, _in(X), _in(Y), _w(weights, _in(X)), nlv, scal, false, ctx; options = _same_x[])
"`plskern!(X::Matrix, Y::Matrix, ...)` — src/plskern.jl:112-178: X, Y are overwritten (centred/scaled)."
plskern!(X, Y, weights = nothing; nlv, scal = false, ctx = default_ctx()) =
    _fit(:plskern, X, Y, _w(weights, X), nlv, scal, true, ctx)
"`plsnipals` — src/plsnipals.jl:31-35."
plsnipals(X, Y, weights = nothing; nlv, scal = false, ctx = default_ctx()) =
    _fit(:plsnipals, _in(X), _in(Y), _w(weights, _in(X)), nlv, scal, false, ctx; options = _nipals_options[] | _same_x[])
"`plsnipals!` — src/plsnipals.jl:37-97: X, Y end up centred/scaled and deflated."
plsnipals!(X, Y, weights = nothing; nlv, scal = false, ctx = default_ctx()) =
    _fit(:plsnipals, X, Y, _w(weights, X), nlv, scal, true, ctx)

# Sibling algorithms (same row kernels, different small state; include/jchemo_hip.h)
"`plssimp` — src/plssimp.jl:22-26 (`W` is returned equal to `R`, :85-87)."
plssimp(X, Y, weights = nothing; nlv, scal = false, ctx = default_ctx()) =
    _fit(:plssimp, _in(X), _in(Y), _w(weights, _in(X)), nlv, scal, false, ctx; options = _same_x[])
"`plssimp!` — src/plssimp.jl:28-88."
plssimp!(X, Y, weights = nothing; nlv, scal = false, ctx = default_ctx()) =
    _fit(:plssimp, X, Y, _w(weights, X), nlv, scal, true, ctx)
"`plsrosa` — src/plsrosa.jl:26-30."
plsrosa(X, Y, weights = nothing; nlv, scal = false, ctx = default_ctx()) =
    _fit(:plsrosa, _in(X), _in(Y), _w(weights, _in(X)), nlv, scal, false, ctx; options = _same_x[])
"`plsrosa!` — src/plsrosa.jl:32-96: X centred/scaled, Y centred/scaled and deflated."
plsrosa!(X, Y, weights = nothing; nlv, scal = false, ctx = default_ctx()) =
    _fit(:plsrosa, X, Y, _w(weights, X), nlv, scal, true, ctx)
"`plswold` — src/plswold.jl:30-34; `niter` filled as :93."
plswold(X, Y, weights = nothing; nlv, tol = sqrt(eps(1.)), maxit = 200, scal = false, ctx = default_ctx()) =
    _fit(:plswold, _in(X), _in(Y), _w(weights, _in(X)), nlv, scal, false, ctx; tol = tol, maxit = maxit, options = _wold_options[] | _nipals_options[] | _same_x[])
"`plswold!` — src/plswold.jl:36-111."
plswold!(X, Y, weights = nothing; nlv, tol = sqrt(eps(1.)), maxit = 200, scal = false, ctx = default_ctx()) =
    _fit(:plswold, X, Y, _w(weights, X), nlv, scal, true, ctx; tol = tol, maxit = maxit, options = _wold_options[])
# jch_pls_desc.reserved for plswold: 0 (default) = finite scores for zero-weight rows (what a zero-weight CV fold needs);
# 2 = JCH_WOLD_REF_ZERO_WEIGHT_NAN, the reference's NaN (src/plswold.jl:107).  A module switch, not a keyword: the keyword list of
# `plswold` stays the reference's (src/plswold.jl:30-31), so higher-order callers can pass it through unchanged.
const _wold_options = Ref{Int32}(0)
"`wold_zero_weight_nan!(true)`: `plswold` gives rows with weight 0 NaN scores as the reference does; `false` (default): finite."
wold_zero_weight_nan!(on::Bool) = (_wold_options[] = on ? Int32(2) : Int32(0); on)

# JCH_NIPALS_ONE_PASS (include/jchemo_hip.h): OPT-IN, never the default — plsnipals / plswold (the non-`!` forms) with ONE pass over X
# per LV: K_{a+1} = K_a - zp_raw c_raw' / tt instead of the reference's recomputation of X'DY from the deflated matrices
# (src/plsnipals.jl:71).  Same results up to rounding.  A module switch for the same reason as above.
const _nipals_options = Ref{Int32}(0)
"`nipals_one_pass!(true)`: `plsnipals` / `plswold` use the one-pass variant (q <= 16, p <= 2048); `false` (default): the reference's schedule."
nipals_one_pass!(on::Bool) = (_nipals_options[] = on ? Int32(4) : Int32(0); on)

# JCH_REUSE_XCOPY (include/jchemo_hip.h): the promise that X — pointer and contents — is what the previous fit on this ctx was given; a
# Float64 plskern-shaped fit then takes X'DY from the row-major copy that fit left in the workspace instead of staging and transposing
# X again.  Set by `gridcvlv` around its second and later fits (the same X, other weights); a module switch like the two above.
const _same_x = Ref{Int32}(0)

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
    hasproperty(object, :lev) && return transform(object.fm, X; nlv = nlv, ctx = ctx)          # Plsrda: src/plsrda.jl:86-88
    hasproperty(object, :bscales) && return _transform_mbplsr(object, X, nlv, ctx)
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

# m x (q * (hi - lo + 1)) predictions for nlv = lo..hi, level-major columns: ONE library call (jch_predict) and one pass over X — up to two
# levels as one GEMM, more as the scores X_c R followed by running sums over the score columns (include/jchemo_hip.h)
function _predict_range(object, X, lo::Integer, hi::Integer, ctx)
    X = _in(X); m, p = size(X); q = size(object.C, 1)
    size(object.R, 1) == p || throw(DimensionMismatch("X has $p columns, the model has $(size(object.R, 1))"))
    out = _similar(X, m, q * (hi - lo + 1))
    R = Matrix{Float64}(object.R); Cm = Matrix{Float64}(object.C)
    xm = Vector{Float64}(vec(object.xmeans)); xs = Vector{Float64}(vec(object.xscales))
    ym = Vector{Float64}(vec(object.ymeans)); ys = Vector{Float64}(vec(object.yscales))
    GC.@preserve X out R Cm xm xs ym ys check(ctx, ccall((:jch_predict, LIB), Int32,
        (Ptr{Cvoid}, Int32, Ptr{Float64}, Int64, Int64, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
         Ptr{Float64}, Int64, Int32, Int32, Ptr{Float64}, Int64),
        ctx.h, _loc(X), pointer(X), m, p, stride(X, 2), pointer(xm), pointer(xs), pointer(ym), pointer(ys), pointer(R), pointer(Cm), q,
        lo, hi, pointer(out), max(m, 1)))
    out
end

"""`predict(object, X; nlv)` — src/plskern.jl:226-238 (a `Plsr` record) or src/lwplsr.jl:134-166 (an `Lwplsr` record).
PLSR: the whole nlv range in ONE pass over X (`jch_predict`)."""
function predict(object, X; nlv = nothing, ctx = default_ctx())
    hasproperty(object, :metric) && return _predict_lwplsr(object, X, nlv, ctx)
    hasproperty(object, :lev) && return _predict_plsrda(object, X, nlv, ctx)
    hasproperty(object, :bscales) && return _predict_mbplsr(object, X, nlv, ctx)
    a = _nlv_fit(object); q = size(object.C, 1)
    rng = nlv === nothing ? (a:a) : (max(0, minimum(nlv)):min(a, maximum(nlv)))
    out = _predict_range(object, X, first(rng), last(rng), ctx)
    pred = [out[:, (i - 1) * q + 1:i * q] for i in 1:length(rng)]
    (pred = length(rng) == 1 ? pred[1] : pred,)
end

"""
    explvarx(object, X)

The table of `summary(object::Plsr, X)` (src/plskern.jl:246-260): a `DataFrame` with the columns `nlv, var, pvar, cumpvar`
when DataFrames is loaded (it is whenever Jchemo is), the same columns as a NamedTuple otherwise; `sstot` computed on the GPU.  `object`: either record; X: the data the model was fitted on.
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
    _table((nlv = collect(1:nlv), var = tt_adj / n, pvar = pvar, cumpvar = cumsum(pvar)))   # src/plskern.jl:258: DataFrame(nlv, var, pvar, cumpvar)
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

# ---- model-constant device data of an Lwplsr object, prepared once (include/jchemo_hip.h: jch_lwplsr_prepare) -------------
# The reference fits `Lwplsr` once and predicts from it many times (src/lwplsr.jl:1-12).  `prepare(fm)` keeps the row-major
# copy of Xtrain, Ytrain and the whitened training scores on the device; `predict(prepared, X)` then only ships the queries.
mutable struct LwplsrPrepared
    object                       # the Lwplsr record (either module's)
    h::Ptr{Cvoid}                # jch_lwplsr_model*
    qmap                         # query block -> coordinates of the neighbour search
    dd::Int
    ctx::JchCtx
    device_map::Bool             # the handle maps the queries itself (jch_lwplsr_add_query_map): predict passes Zq = C_NULL
end

"`prepare(object::Lwplsr; ctx)`: device handle of the model-constant data; release with `release!` (or let the GC do it)."
function prepare(object; ctx = default_ctx())
    Xt = _in(object.X); Yt = _colocate_mat(_in(object.Y), Xt)
    n, p = size(Xt); q = size(Yt, 2)
    Zt, qmap, stages = _knn_train_space(object, Xt, ctx)
    Zt = _colocate_mat(Zt, Xt)
    r = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve Xt Yt Zt check(ctx, ccall((:jch_lwplsr_prepare, LIB), Int32,
        (Ptr{Cvoid}, Int32, Ptr{Float64}, Int64, Int64, Int64, Ptr{Float64}, Int64, Int64, Ptr{Float64}, Int64, Int64, Ref{Ptr{Cvoid}}),
        ctx.h, _loc(Xt), pointer(Xt), n, p, stride(Xt, 2), pointer(Yt), q, max(stride(Yt, 2), n), pointer(Zt), stride(Zt, 2), size(Zt, 2), r))
    pm = LwplsrPrepared(object, r[], qmap, size(Zt, 2), ctx, false)
    finalizer(release!, pm)
    # the query map travels with the handle: two jch_affine_gemm calls per predict (each with an upload of its matrix and a
    # stream synchronisation) become two launches inside jch_lwplsr_predict_prepared
    for (shift, scale, B) in stages
        Bm = Matrix{Float64}(B)
        sh = shift === nothing ? Float64[] : Vector{Float64}(vec(shift))
        sc = scale === nothing ? Float64[] : Vector{Float64}(vec(scale))
        GC.@preserve Bm sh sc check(ctx, ccall((:jch_lwplsr_add_query_map, LIB), Int32,
            (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64, Int64, Ptr{Float64}),
            ctx.h, pm.h, shift === nothing ? Ptr{Float64}(C_NULL) : pointer(sh), scale === nothing ? Ptr{Float64}(C_NULL) : pointer(sc),
            pointer(Bm), size(Bm, 1), size(Bm, 2), Ptr{Float64}(C_NULL)))
    end
    pm.device_map = !isempty(stages)
    pm
end
function release!(pm::LwplsrPrepared)
    pm.h == C_NULL && return nothing
    ccall((:jch_lwplsr_release, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}), C_NULL, pm.h)
    pm.h = C_NULL
    nothing
end

# the space the neighbours are searched in (src/lwplsr.jl:139-150, src/getknn.jl:37-49): training coordinates + query map
function _knn_train_space(object, Xt, ctx)
    stages = Any[]                            # the query map as affine stages (shift, scale, B) for jch_lwplsr_add_query_map
    if object.fm === nothing
        if object.scal                        # :141-145  scale(object.X, colstd(object.X)) on both sides
            xs = col_stats(Xt; ctx = ctx).stds
            Dinv = Matrix(Diagonal(1 ./ xs))
            Zt = _affine(Xt, nothing, nothing, Dinv, nothing, ctx)
            qmap = Xq -> _affine(Xq, nothing, nothing, Dinv, nothing, ctx)
            push!(stages, (nothing, nothing, Dinv))
        else
            Zt = Xt
            qmap = Xq -> Xq
        end
    else
        fmg = object.fm
        Zt = _colocate_mat(fmg.T, Xt)
        qmap = Xq -> transform(fmg, Xq; ctx = ctx)
        push!(stages, (fmg.xmeans, fmg.xscales, fmg.R))
    end
    if object.metric == "mahal"               # src/getknn.jl:37-49
        S = _cov(Zt, ctx); d = size(S, 1)
        Uinv = d == 1 ? fill(1 / sqrt(S[1, 1]), 1, 1) : (isposdef(S) ? Matrix(inv(cholesky(Hermitian(S)).U)) : Matrix(Diagonal(1 ./ diag(S))))
        Zt = _affine(Zt, nothing, nothing, Uinv, nothing, ctx)
        inner = qmap
        qmap = Xq -> _affine(inner(Xq), nothing, nothing, Uinv, nothing, ctx)
        push!(stages, (nothing, nothing, Uinv))
    end
    Zt, qmap, stages
end

"`predict(pm::LwplsrPrepared, X; nlv)` — src/lwplsr.jl:134-166 on the prepared handle."
function predict(pm::LwplsrPrepared, X; nlv = nothing)
    object = pm.object; ctx = pm.ctx
    X = _in(X); m = size(X, 1); n, p = size(object.X); q = size(object.Y, 2)
    a = object.nlv
    rng = nlv === nothing ? (a:a) : (max(minimum(nlv), 0):min(maximum(nlv), a, p))
    Zq = pm.device_map ? X : _colocate_mat(pm.qmap(X), X)    # (device_map: Zq is not read, C_NULL goes down)
    k = min(object.k, n); le = length(rng)
    pred = zeros(q, le, m); ind = zeros(Int32, k, m); dist = zeros(k, m); w = zeros(k, m)   # C layout [m][le][q] == Julia (q, le, m)
    GC.@preserve Zq X check(ctx, ccall((:jch_lwplsr_predict_prepared, LIB), Int32,
        (Ptr{Cvoid}, Ptr{Cvoid}, Int32, Ptr{Float64}, Int64, Ptr{Float64}, Int64, Int64, Int32, Float64, Float64, Int32, Int32, Int32,
         Ptr{Float64}, Ptr{Int32}, Ptr{Float64}, Ptr{Float64}),
        ctx.h, pm.h, _loc(X), pm.device_map ? Ptr{Float64}(C_NULL) : pointer(Zq), stride(Zq, 2), pointer(X), m, stride(X, 2), k, object.h,
        object.tol, object.scal ? 1 : 0, first(rng), last(rng), pred, ind, dist, w))
    preds = [permutedims(pred[:, i, :]) for i in 1:le]
    (pred = le == 1 ? preds[1] : preds, listnn = [Int.(ind[:, i]) .+ 1 for i in 1:m], listd = [dist[:, i] for i in 1:m],
     listw = [w[:, i] for i in 1:m])
end

# `predict(object::Lwplsr, X; nlv)` — src/lwplsr.jl:134-166: neighbours, weights and the m local fits in one call.
function _predict_lwplsr(object, X, nlv, ctx)
    X = _in(X); m = size(X, 1); n, p = size(object.X); q = size(object.Y, 2)
    a = object.nlv
    rng = nlv === nothing ? (a:a) : (max(minimum(nlv), 0):min(maximum(nlv), a, p))
    Xt = _colocate_mat(_in(object.X), X)
    Zt, qmap, _ = _knn_train_space(object, Xt, ctx)
    Zt = _colocate_mat(Zt, X); Zq = _colocate_mat(qmap(X), X)
    k = min(object.k, n); le = length(rng)
    # (no shape limits: outside the batched kernels' envelope the library runs its per-query generic paths, include/jchemo_hip.h)
    pred = zeros(q, le, m); ind = zeros(Int32, k, m); dist = zeros(k, m); w = zeros(k, m)   # C layout [m][le][q] == Julia (q, le, m)
    Yt = _colocate_mat(_in(object.Y), X)
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

# ---- scores from device-side sums (src/scores.jl) ---------------------------------------------------------------------------
# sums[:, c] = (sum e, sum e^2, sum y e, sum y, sum y^2, rows) of prediction column c = level * q + j over the rows with mask != 0
function _score_sums(pred, Y, mask, ctx)
    pred = _in(pred); Y = _colocate_mat(_in(Y), pred)
    m, ncol = size(pred); q = size(Y, 2)
    (size(Y, 1) == m && ncol % q == 0) || throw(DimensionMismatch("predictions are $m x $ncol, Y is $(size(Y, 1)) x $q"))
    mask = mask === nothing ? nothing : _colocate(vec(Float64.(Array(mask))), pred)
    sums = zeros(6, ncol)
    GC.@preserve pred Y mask check(ctx, ccall((:jch_score_sums, LIB), Int32,
        (Ptr{Cvoid}, Int32, Ptr{Float64}, Int64, Int64, Int64, Ptr{Float64}, Int64, Int64, Ptr{Float64}, Ptr{Float64}),
        ctx.h, _loc(pred), pointer(pred), m, ncol, stride(pred, 2), pointer(Y), q, max(stride(Y, 2), m),
        mask === nothing ? Ptr{Float64}(C_NULL) : pointer(mask), sums))
    reshape(sums, 6, q, ncol ÷ q)            # [stat, response, level]
end

# the same sums for the predictions with nlv = lo..hi latent variables straight from the rows' scores T (m x k): running sums over the
# score columns inside the library (jch_score_sums_lv) — the m x (levels q) prediction matrix is never formed
function _score_sums_lv(T, fm, Y, mask, rng, ctx)
    T = _in(T); Y = _colocate_mat(_in(Y), T)
    m = size(T, 1); q = size(Y, 2); k = min(size(T, 2), size(fm.C, 2))
    (size(Y, 1) == m && size(fm.C, 1) == q) || throw(DimensionMismatch("scores are $m x $(size(T, 2)), Y is $(size(Y, 1)) x $q"))
    mask = mask === nothing ? nothing : _colocate(vec(Float64.(Array(mask))), T)
    lo, hi = first(rng), last(rng)
    Cm = Matrix{Float64}(fm.C[:, 1:k]); ym = Vector{Float64}(vec(fm.ymeans)); ys = Vector{Float64}(vec(fm.yscales))
    sums = zeros(6, (hi - lo + 1) * q)
    GC.@preserve T Y mask Cm ym ys check(ctx, ccall((:jch_score_sums_lv, LIB), Int32,
        (Ptr{Cvoid}, Int32, Ptr{Float64}, Int64, Int64, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64, Int64,
         Ptr{Float64}, Int32, Int32, Ptr{Float64}),
        ctx.h, _loc(T), pointer(T), m, k, max(stride(T, 2), m), pointer(Cm), pointer(ym), pointer(ys), pointer(Y), q, max(stride(Y, 2), m),
        mask === nothing ? Ptr{Float64}(C_NULL) : pointer(mask), lo, hi, sums))
    reshape(sums, 6, q, hi - lo + 1)         # [stat, response, level]
end

# one score from the sums: levels x q  (formulas: src/scores.jl:25-32,54-62,155-158,190-196,268,426-429)
function _score_from_sums(name::Symbol, S)
    se, see, sye, sy, syy, cnt = (permutedims(S[i, :, :]) for i in 1:6)
    name === :ssr && return see
    name === :msep && return see ./ cnt
    name === :rmsep && return sqrt.(see ./ cnt)
    name === :bias && return -se ./ cnt
    name === :r2 && return 1 .- (see ./ cnt) ./ (syy ./ cnt .- (sy ./ cnt) .^ 2)
    if name === :cor2
        sp = sy .- se; spp = syy .- 2 .* sye .+ see; spy = syy .- sye          # sums of pred, pred^2, pred * y
        cv = spy ./ cnt .- (sp ./ cnt) .* (sy ./ cnt)
        return cv .^ 2 ./ ((spp ./ cnt .- (sp ./ cnt) .^ 2) .* (syy ./ cnt .- (sy ./ cnt) .^ 2))
    end
    error("unknown score $name")
end

"A score of src/scores.jl as a callable: `rmsep(pred, Y)` -> 1 x q; the grid functions recognise it and use device-side sums."
struct ScoreFun
    name::Symbol
end
(s::ScoreFun)(pred, Y; ctx = default_ctx()) = _score_from_sums(s.name, _score_sums(pred, Y, nothing, ctx))[1:1, :]
const msep = ScoreFun(:msep); const rmsep = ScoreFun(:rmsep); const ssr = ScoreFun(:ssr)
const bias = ScoreFun(:bias); const r2 = ScoreFun(:r2); const cor2 = ScoreFun(:cor2)

# ---- grids (src/gridscore.jl, src/gridcv.jl, src/mpar.jl, src/segm.jl) ---------------------------------------------------------
"`mpar(; kwargs...)` — src/mpar.jl:15-24: all combinations of the parameter values (first keyword fastest), as a NamedTuple of vectors."
function mpar(; kwargs...)
    nam = keys(kwargs); vals = [v isa AbstractVector || v isa AbstractRange || v isa Tuple ? collect(v) : [v] for v in values(kwargs)]
    combs = vec(collect(Iterators.product(vals...)))
    NamedTuple{Tuple(nam)}(Tuple([c[i] for c in combs] for i in 1:length(nam)))
end
"`segmkf(n, K; rep = 1)` — src/segm.jl:44-57."
function segmkf(n::Integer, K::Integer; rep = 1)
    map(1:rep) do _
        perm = _randperm(n)
        [sort(perm[j:K:n]) for j in 1:K]
    end
end
"`segmts(n, m; rep = 1)` — src/segm.jl:135-149."
segmts(n::Integer, m::Integer; rep = 1) = [[sort(_randperm(n)[1:m])] for _ in 1:rep]
_randperm(n) = sortperm(rand(n))
_nlv_range(nlv, p) = max(0, minimum(nlv)):min(p, maximum(nlv))
_pars_rows(pars) = pars === nothing ? [NamedTuple()] : [NamedTuple{keys(pars)}(Tuple(v[i] for v in values(pars))) for i in 1:length(first(values(pars)))]

# m x (length(rng) * q) predictions [pred_rng[1] | pred_rng[2] | ...] (rng contiguous, src/plskern.jl:228): one library call
_pred_matrix(fm, X, rng, ctx) = _predict_range(fm, X, first(rng), min(last(rng), _nlv_fit(fm)), ctx)

# Columns of the reference's result table, in its order (src/gridscore.jl:196-220: `hcat(dat, res)` with dat = the `pars` columns,
# each combination repeated le_nlv times, then `nlv`; res = y1 ... yq), rows combination-major.
function _grid_cols(pars, rng, res::AbstractMatrix)
    rows = _pars_rows(pars)
    cols = NamedTuple()
    if pars !== nothing
        cols = NamedTuple{keys(pars)}(Tuple([r[nm] for r in rows for _ in rng] for nm in keys(pars)))
    end
    cols = merge(cols, (nlv = repeat(collect(rng), length(rows)),))
    merge(cols, NamedTuple{_ynames(size(res, 2))}(Tuple(res[:, j] for j in 1:size(res, 2))))
end

"""
    gridscorelv(Xtrain, Ytrain, X, Y; score, fun, nlv, pars = nothing, verbose = false)

src/gridscore.jl:167-221: one fit at `maximum(nlv)` per parameter combination, the predictions of the whole nlv range from ONE
pass over `X`, the scores from device-side sums.  Returns what the reference returns (:196-220): a `DataFrame` with the columns
`<pars...>, nlv, y1 ... yq`, one row per (combination, nlv), combination-major — as a NamedTuple of those columns when DataFrames
is not loaded.
"""
function gridscorelv(Xtrain, Ytrain, X, Y; score, fun, nlv, pars = nothing, verbose = false, ctx = default_ctx())
    pars === nothing || !(:nlv in keys(pars)) || error("Argument `pars` must not contain `nlv`")
    rng = _nlv_range(nlv, size(ensure_mat(Xtrain), 2))
    verbose && println(pars === nothing ? "-- Nb. combinations = 0." : "-- Nb. combinations = $(length(_pars_rows(pars)))")
    blocks = Matrix{Float64}[]
    for kw in _pars_rows(pars)
        verbose && pars !== nothing && println(pairs(kw)...)
        fm = fun(Xtrain, Ytrain; nlv = maximum(rng), kw...)
        if score isa ScoreFun && hasproperty(fm, :TT) && !hasproperty(fm, :lev)
            kmax = min(maximum(rng), _nlv_fit(fm))
            push!(blocks, kmax == 0 ? _score_from_sums(score.name, _score_sums(_pred_matrix(fm, X, rng, ctx), Y, nothing, ctx)) :
                          _score_from_sums(score.name, _score_sums_lv(transform(fm, X; nlv = kmax, ctx = ctx), fm, Y, nothing, rng, ctx)))
        else                                   # any score(pred, Y) / any model with a `predict`
            pr = predict(fm, X; nlv = rng).pred
            pr = length(rng) == 1 ? [pr] : pr
            push!(blocks, reduce(vcat, [reshape(collect(score(z, Y)), 1, :) for z in pr]))
        end
    end
    verbose && println("-- End.")
    _table(_grid_cols(pars, rng, reduce(vcat, blocks)))
end

"""
    gridcvlv(X, Y; segm, score, fun, nlv, pars = nothing, verbose = false)

src/gridcv.jl:187-228.  The reference copies `rmrow(X, s)` for every segment; here X stays where it is: each fold is ONE
weighted fit with weight 0 on the held-out rows (same means, X'DY and loadings as the fit on the remaining rows), whose scores on
the held-out rows already are their transformed rows, so the predictions for every nlv are running sums over the n x nlv scores.
`score`: one of `msep, rmsep, ssr, bias, r2, cor2`; `fun`: a PLS fit of this module taking `(X, Y, weights; nlv, ...)`.
Returns what the reference returns (:211-227): `(res = table, res_rep = table)` — `res_rep` with the columns
`repl, segm, <pars...>, nlv, y1 ... yq` (one row per replication, segment, combination and nlv), `res` the means of `y1 ... yq` over
replications and segments per `(nlv, <pars...>)` group in order of first appearance (`combine(groupby(res_rep, [:nlv; pars...]), mean)`);
DataFrames when that package is loaded, NamedTuples of the same columns otherwise.
"""
function gridcvlv(X, Y; segm, score, fun, nlv, pars = nothing, verbose = false, ctx = default_ctx())
    score isa ScoreFun || error("gridcvlv: score must be one of msep, rmsep, ssr, bias, r2, cor2")
    pars === nothing || !(:nlv in keys(pars)) || error("Argument `pars` must not contain `nlv`")
    X = _in(X); Y = _colocate_mat(_in(Y), X); n, p = size(X); q = size(Y, 2)
    rng = _nlv_range(nlv, p)
    _same_x[] = Int32(0)                                   # the FIRST fit of the call makes no promise about the workspace
    res_rep = Vector{Vector{Matrix{Float64}}}()
    for (i, listsegm) in enumerate(segm)
        verbose && print("/ repl=", i, " ")
        zres = Matrix{Float64}[]
        for (j, s) in enumerate(listsegm)
            verbose && print("segm=", j, " ")
            held = zeros(n); held[s] .= 1.0
            w = 1.0 .- held
            kfit = min(maximum(rng), n - length(s))                # the reference clamps with the TRAINING rows
            blocks = Matrix{Float64}[]
            for kw in _pars_rows(pars)
                fm = try fun(X, Y, w; nlv = kfit, kw...) catch; _same_x[] = Int32(0); rethrow() end
                _same_x[] = Int32(8)                       # every later fit of this call sees the same X (JCH_REUSE_XCOPY)
                # pred_a = ymeans + sum_{l <= min(a, k)} T_l (C_l .* yscales)' on the held-out rows, level by level inside the library
                push!(blocks, _score_from_sums(score.name, _score_sums_lv(fm.T, fm, Y, held, rng, ctx)))
            end
            push!(zres, reduce(vcat, blocks))
        end
        push!(res_rep, zres)
    end
    _same_x[] = Int32(0)
    verbose && println("/ End.")
    # res_rep: per replication the segments' tables stacked, behind the columns repl, segm (src/gridcv.jl:211-222)
    per = length(rng) * length(_pars_rows(pars))                    # rows per fold
    foldcols = [_grid_cols(pars, rng, z) for zres in res_rep for z in zres]
    repl = reduce(vcat, [fill(i, per * length(zres)) for (i, zres) in enumerate(res_rep)])
    segmc = reduce(vcat, [repeat(1:length(zres), inner = per) for zres in res_rep])
    stacked = NamedTuple{keys(first(foldcols))}(Tuple(reduce(vcat, [c[nm] for c in foldcols]) for nm in keys(first(foldcols))))
    rep_cols = merge((repl = repl, segm = segmc), stacked)
    # res: groupby(res_rep, [:nlv; keys(pars)...]) in order of first appearance = the row order of one fold, means of y1 ... yq (:223-226)
    allfolds = reduce(vcat, res_rep)
    g1 = _grid_cols(pars, rng, sum(allfolds) ./ length(allfolds))
    gkeys = pars === nothing ? (:nlv,) : (:nlv, keys(pars)...)
    res_cols = merge(NamedTuple{gkeys}(Tuple(g1[nm] for nm in gkeys)), NamedTuple{_ynames(q)}(Tuple(g1[nm] for nm in _ynames(q))))
    (res = _table(res_cols), res_rep = _table(rep_cols))
end

# ---- PLSR-DA (src/plsrda.jl) -----------------------------------------------------------------------------------------------------
struct Plsrda                     # fallback record, fields of the reference's struct (src/plsrda.jl:1-5)
    fm; lev; ni
end
"`dummy(y)` — src/utility.jl:509-519: `(Y = n x nlev 0/1 table, lev = sorted levels)`."
function dummy(y)
    y = vec(Array(y)); lev = sort(unique(y))
    (Y = Float64.(y .== permutedims(lev)), lev = lev)
end
"`plsrda(X, y, weights = ones(n); nlv, scal = false)` — src/plsrda.jl:71-77: `plskern` on the dummy table of the classes."
function plsrda(X, y, weights = nothing; nlv, scal = false, ctx = default_ctx())
    res = dummy(y); yv = vec(Array(y))
    ni = [count(==(l), yv) for l in res.lev]
    X = _in(X)
    fm = plskern(X, _colocate_mat(res.Y, X), weights; nlv = nlv, scal = scal, ctx = ctx)
    J = jchemo_module()
    (J !== nothing && fm isa getfield(J, :Plsr)) ? Base.invokelatest(getfield(J, :Plsrda), fm, res.lev, ni) : Plsrda(fm, res.lev, ni)
end
# `predict(object::Plsrda, X; nlv)` — src/plsrda.jl:95-120
function _predict_plsrda(object, X, nlv, ctx)
    a = _nlv_fit(object.fm)
    rng = nlv === nothing ? (a:a) : (max(minimum(nlv), 0):min(maximum(nlv), a))
    post = predict(object.fm, X; nlv = rng, ctx = ctx).pred
    posts = length(rng) == 1 ? [post] : post
    preds = [reshape(object.lev[[argmax(view(Array(z), i, :)) for i in 1:size(z, 1)]], :, 1) for z in posts]
    length(rng) == 1 ? (pred = preds[1], posterior = posts[1]) : (pred = preds, posterior = posts)
end

# ---- multiblock PLSR (src/mbplsr.jl, src/mbplswest.jl:220-254) ------------------------------------------------------------------
struct Mbplsr                     # fields of the reference's struct (src/mbplsr.jl:1-12)
    fm; T; R; C; bscales; xmeans; xscales; ymeans; yscales; weights
end
_hcat(Xbl) = reduce(hcat, [_in(b) for b in Xbl])
"""
    mbplsr(Xbl, Y, weights = ones(n); nlv, bscal = "none", scal = false)

src/mbplsr.jl:64-113.  The reference materialises every centred / scaled / block-scaled block; here the blocks are concatenated
RAW and the whole scaling is ONE vector of column divisors (column std x block scale) handed to `jch_plskern_fit_scaled`.
"""
function mbplsr(Xbl, Y, weights = nothing; nlv, bscal = "none", scal = false, ctx = default_ctx())
    bscal in ("none", "frob") || error("bscal must be \"none\" or \"frob\"")
    X = _hcat(Xbl); Y = _colocate_mat(_in(Y), X)
    widths = [size(ensure_mat(b), 2) for b in Xbl]; edges = cumsum([0; widths])
    st = col_stats(X, weights; ctx = ctx)
    blk(v, k) = v[edges[k] + 1:edges[k + 1]]
    xscales = [scal ? blk(st.stds, k) : ones(widths[k]) for k in 1:length(widths)]
    bscales = bscal == "frob" ? [sqrt(sum((blk(st.stds, k) ./ xscales[k]) .^ 2)) for k in 1:length(widths)] : ones(length(widths))
    div = reduce(vcat, [xscales[k] .* bscales[k] for k in 1:length(widths)])
    ysd = scal ? col_stats(Y, weights; ctx = ctx).stds : nothing
    fm = plskern_scaled(X, Y, div, ysd, weights; nlv = nlv, ctx = ctx)
    xmeans = [blk(fm.xmeans, k) for k in 1:length(widths)]
    # the reference's inner fit sees pre-scaled data with scal = false: its record carries zero means and unit scales
    inner = _record(fm.T, fm.P, fm.R, fm.W, fm.C, fm.TT, zero(fm.xmeans), one.(fm.xscales), zero(fm.ymeans), one.(fm.yscales), fm.weights, nothing)
    Mbplsr(inner, fm.T, fm.R, fm.C, bscales, xmeans, xscales, copy(fm.ymeans), copy(fm.yscales), fm.weights)
end
# `transform(object::Mbplsr, Xbl; nlv)` — src/mbplswest.jl:220-231 as one device GEMM on the raw concatenation
function _transform_mbplsr(object, Xbl, nlv, ctx)
    a = size(object.R, 2); k = nlv === nothing ? a : min(nlv, a)
    div = reduce(vcat, [object.xscales[i] .* object.bscales[i] for i in 1:length(object.xscales)])
    _affine(_hcat(Xbl), reduce(vcat, object.xmeans), div, object.R[:, 1:k], nothing, ctx)
end
# `predict(object::Mbplsr, Xbl; nlv)` — src/mbplswest.jl:239-254: ymeans .+ T[:, 1:nlv] * C[:, 1:nlv]' (no `yscales`, as the reference)
function _predict_mbplsr(object, Xbl, nlv, ctx)
    a = size(object.R, 2); q = size(object.C, 1)
    rng = nlv === nothing ? (a:a) : (max(0, minimum(nlv)):min(a, maximum(nlv)))
    T = _transform_mbplsr(object, Xbl, nothing, ctx)
    Bc = zeros(a, length(rng) * q)
    for (i, k) in enumerate(rng)
        k > 0 && (Bc[1:k, (i - 1) * q + 1:i * q] = object.C[:, 1:k]')
    end
    out = _affine(T, nothing, nothing, Bc, repeat(object.ymeans, length(rng)), ctx)
    preds = [out[:, (i - 1) * q + 1:i * q] for i in 1:length(rng)]
    (pred = length(rng) == 1 ? preds[1] : preds,)
end

# ---- vip / xfit / xresid (src/vip.jl:62-107, src/xfit.jl:37-93) ---------------------------------------------------------------
"`vip(object; nlv)` — src/vip.jl:62-89: from W, C and TT = t'Dt (p x nlv host glue)."
function vip(object; nlv = nothing)
    a = _nlv_fit(object); p = size(object.W, 1); k = nlv === nothing ? a : min(nlv, a)
    W2 = object.W[:, 1:k] .^ 2
    sst = vec(sum(object.C[:, 1:k] .^ 2, dims = 1)) .* object.TT[1:k]          # tr(C_a C_a') t_a'D t_a
    A = vec(sum(sst' .* W2, dims = 2))
    (imp = sqrt.(A ./ (sum(sst) / p)), W2 = W2, sst = sst)
end
"`vip(object, Y; nlv)` — src/vip.jl:91-107: the redundancies rd(Y, T, weights) from ONE weighted covariance of [Y | T] on the device."
function vip(object, Y; nlv = nothing, ctx = default_ctx())
    a = _nlv_fit(object); p = size(object.W, 1); k = nlv === nothing ? a : min(nlv, a)
    T = object.T[:, 1:k]; Y = _colocate_mat(_in(Y), T); q = size(Y, 2)
    A_ = hcat(Y, T); n = size(A_, 1); w = _colocate(object.weights, A_)
    S = zeros(q + k, q + k)
    GC.@preserve A_ w check(ctx, ccall((:jch_weighted_cov, LIB), Int32,
        (Ptr{Cvoid}, Int32, Ptr{Float64}, Int64, Int64, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
        ctx.h, _loc(A_), pointer(A_), n, q + k, stride(A_, 2), pointer(w), S, Ptr{Float64}(C_NULL)))
    # rd(Y, T, weights) (src/angles.jl:97-105): mean over the responses of the squared weighted correlations cor(y_j, t_a)^2
    cyt = S[1:q, q + 1:q + k]; vy = diag(S)[1:q]; vt = diag(S)[q + 1:q + k]
    rdd = vec(sum(cyt .^ 2 ./ (vy .* vt'), dims = 1)) ./ q
    W2 = object.W[:, 1:k] .^ 2
    Aimp = vec(sum(rdd' .* W2, dims = 2))
    (imp = sqrt.(Aimp ./ (sum(rdd) / p)), W2 = W2, rdd = rdd)
end
"`xfit(object, X; nlv)` — src/xfit.jl:37-56: the scores pass over X, then a GEMM on the m x nlv scores, original scale."
function xfit(object, X; nlv = nothing, ctx = default_ctx())
    a = _nlv_fit(object); k = nlv === nothing ? a : min(nlv, a); p = size(object.P, 1)
    k == 0 && return _affine(X, nothing, nothing, zeros(p, p), object.xmeans, ctx)
    Tq = transform(object, X; nlv = k, ctx = ctx)
    _affine(Tq, nothing, nothing, Matrix((object.P[:, 1:k] .* object.xscales)'), object.xmeans, ctx)
end
"`xresid(object, X; nlv)` — src/xfit.jl:86-93: E = X - xfit(X) = cscale(X) (I - R_k P_k') diag(xscales), one device GEMM."
function xresid(object, X; nlv = nothing, ctx = default_ctx())
    a = _nlv_fit(object); k = nlv === nothing ? a : min(nlv, a); p = size(object.P, 1)
    M = Matrix{Float64}(I, p, p) - object.R[:, 1:k] * object.P[:, 1:k]'
    _affine(X, object.xmeans, object.xscales, M .* object.xscales', nothing, ctx)
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
