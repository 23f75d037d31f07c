# KissMCMCHIP.jl -- thin `ccall` shim over libkissmcmc_hip.so (C ABI: include/kissmcmc_hip.h).
#
# EXTENDS KissMCMC.jl, it does not replace it: `emcee` and `metropolis` get one more method each -- for a `pdf` that is
# one of the device log-densities below (menu, `ExprDensity`) or a `HostLogPdf(f)` around any Julia callable (that one
# stays on the host, KMC_HOST_DENSITY: called on each half-step's batch of proposals while moves, draws, accept test
# and storage run on the GPU) -- with KissMCMC's own keywords and return values (reference src/samplers.jl:188-216,
# :59-77).  `make_theta0s` (src/samplers.jl:311-349) and `squash_walkers` (:372-428) are host-side pre/post-processing
# on exactly the containers these methods take and return, so they are KissMCMC's OWN functions, re-exported here, not
# re-implemented: a device log-density is callable on the host with the kernels' formula, which is all `make_theta0s`
# needs.  A plain closure as `pdf` still dispatches to KissMCMC's CPU methods; wrap it, `HostLogPdf(f)`, to move the
# sampler to the GPU.  There is no CPU fallback inside this module.
#
# A package (julia/Project.toml, julia/test/runtests.jl: `] dev path/to/kissmcmc.jl_amd/julia`, `] test KissMCMCHIP` on a machine with Julia and an MI355X).
# NOT EXECUTED in the build environment (no julia binary there).  Every call it makes is mirrored 1:1 by the Python
# ctypes host (kissmcmc.jl_amd/_lib.py, api.py), which is what the tests drive; struct layouts are checked against the
# library when the module loads (`__init__`).
module KissMCMCHIP

import KissMCMC
import KissMCMC: emcee, metropolis, make_theta0s, squash_walkers     # extended (emcee, metropolis) / re-exported as they are

export emcee, make_theta0s, squash_walkers, metropolis, metropolis_chains, GaussianStep, HostProposal, int_acorr, GaussianIso, Exponential, Rosenbrock, LogNormal, MvNormal2, ExprDensity, CDensity, HostLogPdf

using LinearAlgebra: inv

const LIB = get(ENV, "KMC_LIB_PATH", joinpath(@__DIR__, "..", "..", "libkissmcmc_hip.so"))     # julia/src/ -> the package directory that holds the built library

# ---- C structs: field for field include/kissmcmc_hip.h (kmc_config, kmc_outputs).  Built by keyword, so a new field
#      of the header needs one line here and no call site changes; `__init__` compares sizeof with the library's.
#      (They come first: a ccall's argument types are resolved when the enclosing method is defined.) ---
Base.@kwdef struct KmcConfig
    dtype::Int32 = 0                                  # KMC_F64 = 0, KMC_F32 = 1
    density::Int32 = 0
    params::NTuple{8,Float64} = ntuple(_ -> 0.0, 8)
    nwalkers::Int64 = 0
    ndim::Int64 = 0
    ngenerations::Int64 = 0
    nburnin::Int64 = 0
    nthin::Int64 = 1
    a_scale::Float64 = 2.0
    seed::UInt64 = 0
    flags::UInt32 = 0
    device::Int32 = 0
    shard_rank::Int32 = 0
    shard_count::Int32 = 1
    user_density::Ptr{Cvoid} = C_NULL
    island_gens::Int32 = 0                            # KMC_ISLANDS (opt-in island mode); 0 = defaults
    island_size::Int32 = 0
    host_logpdf::Ptr{Cvoid} = C_NULL                  # KMC_HOST_DENSITY callback ...
    host_user::Ptr{Cvoid} = C_NULL                    # ... and its context
    host_accepted::Ptr{Cvoid} = C_NULL                # KMC_HOST_DENSITY: accept outcomes per half-step (blobs), or NULL
    deal_rank::Int32 = 0                              # dealt sub-ensembles (multi-GPU, opt-in); deal_count = 0: off
    deal_count::Int32 = 0
end

Base.@kwdef mutable struct KmcOutputs
    chain::Ptr{Float64} = C_NULL
    chain_logp::Ptr{Float64} = C_NULL
    accept_ratio::Ptr{Float64} = C_NULL
    naccept::Ptr{Int64} = C_NULL
    final_pos::Ptr{Float64} = C_NULL
    final_logp::Ptr{Float64} = C_NULL
    sum::Ptr{Float64} = C_NULL
    sumsq::Ptr{Float64} = C_NULL
    nmoment::Int64 = 0
    nsamples::Int64 = 0
    device_ms::Float64 = 0.0
    blobs::Ptr{Float64} = C_NULL
end

const KMC_STORE_CHAIN = UInt32(1) << 0
const KMC_STORE_LOGP = UInt32(1) << 1
const KMC_CHAIN_BY_WALKER = UInt32(1) << 12     # chain delivered as [walker][sample][dim]: thetas[w][k] are contiguous
const KMC_STORE_BLOBS = UInt32(1) << 13         # a CDensity(body; nblob=m): the blob of every stored sample (src/samplers.jl:270, :117)

last_error() = unsafe_string(ccall((:kmc_last_error, LIB), Cstring, ()))

# ---- menu densities: callable on the host with the same formula the kernels use ----------------
abstract type DeviceLogPdf end
struct GaussianIso <: DeviceLogPdf; mu::Float64; sigma::Float64; end
GaussianIso() = GaussianIso(0.0, 1.0)
struct Exponential <: DeviceLogPdf; rate::Float64; end
Exponential() = Exponential(1.0)
struct Rosenbrock <: DeviceLogPdf; a::Float64; b::Float64; scale::Float64; end
Rosenbrock() = Rosenbrock(1.0, 100.0, 20.0)
struct LogNormal <: DeviceLogPdf; mu::Float64; sigma::Float64; end
struct MvNormal2 <: DeviceLogPdf; mean::Vector{Float64}; prec::Matrix{Float64}; end
MvNormal2(mean, cov::AbstractMatrix) = MvNormal2(collect(Float64, mean), inv(Matrix{Float64}(cov)))

(d::GaussianIso)(x) = -0.5 * sum(((x .- d.mu) ./ d.sigma) .^ 2)
(d::Exponential)(x) = any(x .< 0) ? -Inf : -d.rate * sum(x)          # README.md:15
(d::Rosenbrock)(x) = -sum(d.b .* (x[2:end] .- x[1:end-1] .^ 2) .^ 2 .+ (d.a .- x[1:end-1]) .^ 2) / d.scale  # test/runtests.jl:68
(d::LogNormal)(x) = any(x .<= 0) ? -Inf : sum(-log.(x) .- 0.5 .* ((log.(x) .- d.mu) ./ d.sigma) .^ 2)
(d::MvNormal2)(x) = -0.5 * ((x .- d.mean)' * d.prec * (x .- d.mean))

density_id(::GaussianIso) = Cint(0); params(d::GaussianIso) = [d.mu, d.sigma]
density_id(::Exponential) = Cint(1); params(d::Exponential) = [d.rate]
density_id(::Rosenbrock) = Cint(2);  params(d::Rosenbrock) = [d.a, d.b, d.scale]
density_id(::LogNormal) = Cint(3);   params(d::LogNormal) = [d.mu, d.sigma]
density_id(::MvNormal2) = Cint(4);   params(d::MvNormal2) = [d.mean[1], d.mean[2], d.prec[1,1], 0.5 * (d.prec[1,2] + d.prec[2,1]), d.prec[2,2]]

# Runtime-compiled user density (hiprtc): log p = sum_d term(x_d) + sum_{d<n-1} pair(x_d, x_{d+1}); see
# kmc_user_density_create in include/kissmcmc_hip.h.  The host call evaluates it on the device.
mutable struct ExprDensity <: DeviceLogPdf
    handle::Ptr{Cvoid}; p::Vector{Float64}; nblob::Int
    ExprDensity(handle::Ptr{Cvoid}, p::Vector{Float64}, nblob::Int=0) = new(handle, p, nblob)   # (an existing handle: CDensity below)
    function ExprDensity(term::String, pair::Union{String,Nothing}=nothing; params=Float64[])
        h = Ref{Ptr{Cvoid}}(C_NULL)
        st = ccall((:kmc_user_density_create, LIB), Cint, (Cstring, Cstring, Ref{Ptr{Cvoid}}), term, pair === nothing ? "" : pair, h)
        st == 0 || error("kmc_user_density_create failed: $(last_error())")
        d = new(h[], collect(Float64, params), 0)
        finalizer(x -> ccall((:kmc_user_density_destroy, LIB), Cvoid, (Ptr{Cvoid},), x.handle), d)
        return d
    end
end
# The general device form: the BODY of `double logpdf(const double* x, int n, const double* p) { BODY }` (C++), any coupling
# between the dimensions; runs one walker per lane (kmc_user_density_create_body).  `nblob=m`: the reference's
# `pdf(theta) -> (p, blob)` of hasblob=true (src/samplers.jl:150-151) on the device -- the body is then that of
# `double logpdf(const double* x, int n, const double* p, double* blob)` and fills blob[0..m) (kmc_user_density_create_body_blob).
function CDensity(body::String; params=Float64[], nblob::Int=0)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    st = nblob > 0 ? ccall((:kmc_user_density_create_body_blob, LIB), Cint, (Cstring, Cint, Ref{Ptr{Cvoid}}), body, nblob, h) :
                     ccall((:kmc_user_density_create_body, LIB), Cint, (Cstring, Ref{Ptr{Cvoid}}), body, h)
    st == 0 || error("kmc_user_density_create_body failed: $(last_error())")
    d = ExprDensity(h[], collect(Float64, params), nblob)
    finalizer(x -> ccall((:kmc_user_density_destroy, LIB), Cvoid, (Ptr{Cvoid},), x.handle), d)
    return d
end
density_id(::ExprDensity) = Cint(100); params(d::ExprDensity) = d.p
user_handle(d::ExprDensity) = d.handle
user_handle(::DeviceLogPdf) = C_NULL
function (d::ExprDensity)(x)
    xs = collect(Float64, x isa Number ? [x] : x); out = Ref(0.0)
    p8 = ntuple(i -> i <= length(d.p) ? d.p[i] : 0.0, 8)
    cfg = Ref(KmcConfig(density=Cint(100), params=p8, nwalkers=2, ndim=length(xs), user_density=d.handle))
    st = ccall((:kmc_logpdf_eval_host, LIB), Cint, (Ref{KmcConfig}, Ptr{Float64}, Ref{Float64}, Int64), cfg, xs, out, 1)
    st == 0 || error(last_error()); out[]
end

# Any Julia callable as the log-density (the reference's closure, src/samplers.jl:257), evaluated on
# the host: kmc_config.host_logpdf points at `host_trampoline`, host_user at the wrapper object.
# With hasblob the closure returns (p, blob) (src/samplers.jl:150-151): the blobs stay on the host --
# `batch` holds those of the rows evaluated last, `blob0s` the walkers' current ones (:264), `blobs` the
# per-walker storage (:238, :270) -- and follow the device's accept decisions, reported per half-step
# through kmc_config.host_accepted (`accepted_trampoline`).
mutable struct HostLogPdf{F} <: DeviceLogPdf
    f::F; scalar::Bool; hasblob::Bool
    batch::Vector{Any}; blob0s::Vector{Any}; blobs::Vector{Any}
    init_blobs::Any; reduce_blob!::Any; nsamples::Int
end
HostLogPdf(f; hasblob=false) = HostLogPdf(f, false, hasblob, Any[], Any[], Any[], nothing, nothing, 0)
(d::HostLogPdf)(x) = d.f(x)
density_id(::HostLogPdf) = Cint(101); params(::HostLogPdf) = Float64[]
function host_trampoline(rows::Ptr{Float64}, nrows::Int64, ndim::Int64, out::Ptr{Float64}, user::Ptr{Cvoid})::Cint
    try
        d = unsafe_pointer_to_objref(user)
        X = unsafe_wrap(Array, rows, (ndim, nrows))       # column w = proposal of walker w
        d.hasblob && resize!(d.batch, nrows)
        for w in 1:nrows
            r = d.scalar ? d.f(X[1, w]) : d.f(X[:, w])
            if d.hasblob
                unsafe_store!(out, Float64(r[1]), w); d.batch[w] = r[2]                 # p1, blob1 = pdf(theta1)  :257
            else
                unsafe_store!(out, Float64(r), w)
            end
        end
        return Cint(0)
    catch
        return Cint(1)                                     # kmc_emcee_run then fails with KMC_ERR_BAD_ARG
    end
end
function accepted_trampoline(accepted::Ptr{UInt8}, nrows::Int64, row0::Int64, generation::Int64, stored::Int32, user::Ptr{Cvoid})::Cint
    try
        d = unsafe_pointer_to_objref(user)
        if isempty(d.blob0s)        # first half-step: the last full evaluation was the initial one (:209-210)
            error("initial blobs missing")
        end
        for i in 1:nrows
            unsafe_load(accepted, i) != 0 && (d.blob0s[row0 + i] = d.batch[i])          # :264
        end
        if stored != 0
            for w in (row0 + 1):(row0 + nrows)
                d.reduce_blob!(d.blobs[w], d.blob0s[w])                                  # :270
            end
        end
        return Cint(0)
    catch
        return Cint(1)
    end
end

"""
    emcee(pdf::DeviceLogPdf, theta0s; niter=10^5, nburnin=niter÷2, nthin=1, a_scale=2.0,
          use_progress_meter=true, hasblob=false, init_blobs, reduce_blob!, seed=rand(UInt64), device=0, dtype=:f64)

Same meaning as KissMCMC.emcee (src/samplers.jl:188-197); returns
`(thetas, accept_ratio, logdensities, blobs)` with `thetas[w][k]` (src/samplers.jl:292).
`hasblob=true` needs a `pdf` that returns a blob: a host closure (`HostLogPdf(f; hasblob=true)`, blobs of any type, kept on
the host) or a `CDensity(body; nblob=m)` (m doubles computed and carried on the device; `blobs[w][k]::Vector{Float64}`).
"""
function emcee(pdf::DeviceLogPdf, theta0s; niter=10^5, nburnin=niter ÷ 2, nthin=1, a_scale=2.0,
               use_progress_meter=true, hasblob=false,
               init_blobs=(blob0, nsamples) -> sizehint!(typeof(blob0)[], nsamples),      # init_output_vector :80-85
               reduce_blob! =(blobs, blob) -> push!(blobs, blob),                         # :196
               seed=rand(UInt64), device=0, dtype=:f64)     # dtype=:f32: float rows on the device (KMC_F32), built-in densities
    device_blobs = hasblob && pdf isa ExprDensity && pdf.nblob > 0
    hasblob && !device_blobs && !(pdf isa HostLogPdf && pdf.hasblob) &&
        error("hasblob=true needs a pdf that returns a blob: HostLogPdf(f; hasblob=true) or CDensity(body; nblob=m)")
    nwalkers = length(theta0s)
    scalar = theta0s[1] isa Number
    ndim = length(theta0s[1])
    @assert a_scale > 1                                                                  # :200
    @assert iseven(nwalkers) "Use an even number of walkers."                           # :202
    niter_walker = niter ÷ nwalkers                                                      # :203
    nburnin_walker = nburnin ÷ nwalkers                                                  # :204
    @assert nwalkers >= ndim + 2 "Use more walkers: at least DOF+2, but better many more."  # :205
    nsamples = max(0, (niter_walker - nburnin_walker) ÷ nthin)                           # :234

    theta = Matrix{Float64}(undef, ndim, nwalkers)        # column-major [dim, walker] == C row-major [walker][dim]
    for w in 1:nwalkers, d in 1:ndim
        theta[d, w] = scalar ? theta0s[w] : theta0s[w][d]                                # :198 deep copy
    end
    p = params(pdf); p8 = ntuple(i -> i <= length(p) ? p[i] : 0.0, 8)
    host_fn, host_ctx, acc_fn = C_NULL, C_NULL, C_NULL
    if pdf isa HostLogPdf
        pdf.scalar = scalar
        host_fn = @cfunction(host_trampoline, Cint, (Ptr{Float64}, Int64, Int64, Ptr{Float64}, Ptr{Cvoid}))
        host_ctx = pointer_from_objref(pdf)
        if pdf.hasblob
            # the initial evaluations (:209-210) here, so that blob storage exists before the run; the library
            # evaluates the same rows once more for its own log-pdfs
            tmp = [pdf.f(scalar ? theta0s[w] : collect(Float64, theta0s[w])) for w in 1:nwalkers]
            pdf.blob0s = Any[t[2] for t in tmp]
            pdf.blobs = Any[init_blobs(pdf.blob0s[w], nsamples) for w in 1:nwalkers]      # :238
            pdf.init_blobs, pdf.reduce_blob!, pdf.nsamples = init_blobs, reduce_blob!, nsamples
            acc_fn = @cfunction(accepted_trampoline, Cint, (Ptr{UInt8}, Int64, Int64, Int64, Int32, Ptr{Cvoid}))
        end
    end
    chain = Array{Float64}(undef, ndim * nsamples * nwalkers)
    clogp = Array{Float64}(undef, nsamples * nwalkers)
    acc = Vector{Float64}(undef, nwalkers)
    nblob = device_blobs ? pdf.nblob : 0
    bl = Array{Float64}(undef, nblob * nsamples * nwalkers)          # device blobs: [walker][sample][nblob] (src/samplers.jl:270)
    out = KmcOutputs(chain=pointer(chain), chain_logp=pointer(clogp), accept_ratio=pointer(acc), blobs=(device_blobs ? pointer(bl) : C_NULL))
    # The chain in the reference's own order (thetas[w][k], :219-221), reordered on the device.  A chain too large for
    # the device is streamed into these arrays while sampling (kmc_emcee_run decides, KMC_STREAM_CHAIN) and then arrives
    # sample-major: that combination is refused (status 9) and the call is repeated without the flag.
    st = 9; by_walker = true
    for flag in (KMC_CHAIN_BY_WALKER, UInt32(0))
        by_walker = flag != 0
        cfg = Ref(KmcConfig(dtype=(dtype == :f32 ? 1 : 0), density=density_id(pdf), params=p8, nwalkers=nwalkers, ndim=ndim,
                            ngenerations=niter_walker, nburnin=nburnin_walker, nthin=nthin, a_scale=a_scale, seed=UInt64(seed),
                            flags=KMC_STORE_CHAIN | KMC_STORE_LOGP | flag, device=Int32(device), user_density=user_handle(pdf),
                            host_logpdf=host_fn, host_user=host_ctx, host_accepted=acc_fn))
        st = GC.@preserve pdf theta chain clogp acc bl ccall((:kmc_emcee_run, LIB), Cint,
                                                       (Ref{KmcConfig}, Ptr{Float64}, Ref{KmcOutputs}), cfg, theta, out)
        (st == 9 && by_walker && occursin("KMC_CHAIN_BY_WALKER", last_error())) || break
    end
    st == 0 || error("kmc_emcee_run failed ($st): $(last_error())")
    # column-major views of the C arrays: [dim, sample, walker] (by walker) or [dim, walker, sample]
    ch = by_walker ? reshape(chain, ndim, nsamples, nwalkers) : permutedims(reshape(chain, ndim, nwalkers, nsamples), (1, 3, 2))
    lp = by_walker ? reshape(clogp, nsamples, nwalkers) : permutedims(reshape(clogp, nwalkers, nsamples), (2, 1))
    thetas = scalar ? [ch[1, :, w] for w in 1:nwalkers] :
                      [[ch[:, k, w] for k in 1:nsamples] for w in 1:nwalkers]
    logdensities = [lp[:, w] for w in 1:nwalkers]
    if device_blobs
        # blob0s = the blobs of the initial evaluations (:209-210); then the caller's init_blobs / reduce_blob! over each
        # walker's stored series, in order (:238, :270)
        lp0 = Vector{Float64}(undef, nwalkers); b0 = Matrix{Float64}(undef, nblob, nwalkers)
        cfg0 = Ref(KmcConfig(density=density_id(pdf), params=p8, nwalkers=nwalkers, ndim=ndim, user_density=user_handle(pdf), device=Int32(device)))
        st0 = ccall((:kmc_logpdf_blob_eval_host, LIB), Cint, (Ref{KmcConfig}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64), cfg0, theta, lp0, b0, nwalkers)
        st0 == 0 || error("kmc_logpdf_blob_eval_host failed ($st0): $(last_error())")
        B = by_walker ? reshape(bl, nblob, nsamples, nwalkers) : permutedims(reshape(bl, nblob, nwalkers, nsamples), (1, 3, 2))
        blobs = [init_blobs(b0[:, w], nsamples) for w in 1:nwalkers]
        for w in 1:nwalkers, k in 1:nsamples
            reduce_blob!(blobs[w], B[:, k, w])
        end
        return thetas, acc, logdensities, blobs
    end
    return thetas, acc, logdensities, (pdf isa HostLogPdf && pdf.hasblob) ? pdf.blobs : nothing   # :292
end
# (a bare closure as `pdf` is KissMCMC.emcee's own CPU method; `emcee(HostLogPdf(f; hasblob), theta0s; ...)` runs the
#  sampler on the GPU with `f` evaluated on the host)

# ---- many-chain Metropolis: metropolis / _metropolis, src/samplers.jl:59-128 ---------------------
"Symmetric proposal `theta -> scale .* randn(n) .+ theta` (the one all reference tests use, test/runtests.jl:54,59,64,75)."
struct GaussianStep; scale::Vector{Float64}; end
GaussianStep(c::Real) = GaussianStep([Float64(c)])

Base.@kwdef struct KmcMetropolisConfig
    dtype::Int32 = 0
    density::Int32 = 0
    params::NTuple{8,Float64} = ntuple(_ -> 0.0, 8)
    nchains::Int64 = 0
    ndim::Int64 = 0
    niter::Int64 = 0
    nburnin::Int64 = 0
    nthin::Int64 = 1
    step::Ptr{Float64} = C_NULL
    seed::UInt64 = 0
    flags::UInt32 = 0
    device::Int32 = 0
    user_density::Ptr{Cvoid} = C_NULL
    host_logpdf::Ptr{Cvoid} = C_NULL                  # host route: `pdf` is a closure (density == KMC_HOST_DENSITY) ...
    host_user::Ptr{Cvoid} = C_NULL
    host_accepted::Ptr{Cvoid} = C_NULL                # ... accept outcomes per iteration (blobs)
    host_propose::Ptr{Cvoid} = C_NULL                 # ... `sample_ppdf` is a closure (then `step` may be NULL)
end

Base.@kwdef mutable struct KmcMetropolisOutputs
    chain::Ptr{Float64} = C_NULL
    chain_logp::Ptr{Float64} = C_NULL
    accept_ratio::Ptr{Float64} = C_NULL
    naccept::Ptr{Int64} = C_NULL
    final_pos::Ptr{Float64} = C_NULL
    final_logp::Ptr{Float64} = C_NULL
    chain_sum::Ptr{Float64} = C_NULL
    chain_sumsq::Ptr{Float64} = C_NULL
    nsamples::Int64 = 0
    device_ms::Float64 = 0.0
    blobs::Ptr{Float64} = C_NULL
    final_blob::Ptr{Float64} = C_NULL
end

# Layout drift between these mirrors and the library fails here, when the module loads -- not inside the first real ccall.
function __init__()
    for (n, T) in ((ccall((:kmc_sizeof_config, LIB), Cint, ()), KmcConfig), (ccall((:kmc_sizeof_metropolis_config, LIB), Cint, ()), KmcMetropolisConfig),
                   (ccall((:kmc_sizeof_outputs, LIB), Cint, ()), KmcOutputs), (ccall((:kmc_sizeof_metropolis_outputs, LIB), Cint, ()), KmcMetropolisOutputs))
        n == sizeof(T) || error("$LIB: $(T) is $(sizeof(T)) bytes here but $n in the library (include/kissmcmc_hip.h changed: update the struct in KissMCMCHIP.jl)")
    end
end

# `sample_ppdf` as ANY Julia callable (src/samplers.jl:41, :98), kept on the host: called once per iteration on every
# chain's current state.  The host callbacks of one run share one context object, `MetroCtx`.
mutable struct HostProposal{F}; f::F; scalar::Bool; end
HostProposal(f) = HostProposal(f, false)
mutable struct MetroCtx; pdf::Any; prop::Any; end
function metro_pdf_trampoline(rows::Ptr{Float64}, nrows::Int64, ndim::Int64, out::Ptr{Float64}, user::Ptr{Cvoid})::Cint
    ctx = unsafe_pointer_to_objref(user)::MetroCtx
    return host_trampoline(rows, nrows, ndim, out, pointer_from_objref(ctx.pdf))
end
function metro_propose_trampoline(rows::Ptr{Float64}, nrows::Int64, ndim::Int64, out::Ptr{Float64}, user::Ptr{Cvoid})::Cint
    try
        d = (unsafe_pointer_to_objref(user)::MetroCtx).prop
        X = unsafe_wrap(Array, rows, (ndim, nrows)); Y = unsafe_wrap(Array, out, (ndim, nrows))
        for c in 1:nrows
            Y[:, c] .= d.scalar ? d.f(X[1, c]) : d.f(X[:, c])                            # theta1 = sample_ppdf(theta0)  :98
        end
        return Cint(0)
    catch
        return Cint(1)
    end
end

"""
    metropolis_chains(pdf::DeviceLogPdf, sample_ppdf::Union{GaussianStep,HostProposal}, theta0s; niter=10^5, nburnin=niter÷2, nthin=1, seed, device)

One independent Metropolis chain (src/samplers.jl:96-126) per element of `theta0s`; `niter`/`nburnin` count steps per
chain.  With a menu / `ExprDensity` `pdf` and a `GaussianStep` the chains run inside one kernel, one chain per GPU lane;
a `HostLogPdf(f)` and / or a `HostProposal(g)` keep those closures on the host (one batch call per iteration) while the
accept test, counters and storage stay on the device.  Returns `(thetas, accept_ratio, logdensities, nothing)` shaped
like `emcee`'s output (`thetas[chain][sample]`), so `squash_walkers` applies.
"""
function metropolis_chains(pdf::DeviceLogPdf, sample_ppdf::Union{GaussianStep,HostProposal}, theta0s; niter=10^5, nburnin=niter ÷ 2, nthin=1,
                           hasblob=false, init_blobs=(blob0, nsamples) -> sizehint!(typeof(blob0)[], nsamples),
                           reduce_blob! =(blobs, blob) -> push!(blobs, blob), seed=rand(UInt64), device=0)
    device_blobs = hasblob && pdf isa ExprDensity && pdf.nblob > 0 && sample_ppdf isa GaussianStep
    hasblob && !device_blobs &&
        error("hasblob=true for metropolis: a CDensity(body; nblob=m) with a GaussianStep (blobs carried on the device); host closures with blobs are not wired in this shim (the C ABI carries them: kmc_metropolis_config.host_accepted)")
    nchains = length(theta0s); scalar = theta0s[1] isa Number; ndim = length(theta0s[1])
    nsamples = niter > nburnin ? (niter - nburnin) ÷ nthin : 0                            # :88
    theta = Matrix{Float64}(undef, ndim, nchains)          # column-major [dim, chain] == C row-major [chain][dim]
    for c in 1:nchains, d in 1:ndim
        theta[d, c] = scalar ? theta0s[c] : theta0s[c][d]                                # :68 deep copy
    end
    step = Float64[]
    prop_fn = C_NULL
    if sample_ppdf isa GaussianStep
        step = length(sample_ppdf.scale) == 1 ? fill(sample_ppdf.scale[1], ndim) : copy(sample_ppdf.scale)
        @assert length(step) == ndim
    else
        sample_ppdf.scalar = scalar
        prop_fn = @cfunction(metro_propose_trampoline, Cint, (Ptr{Float64}, Int64, Int64, Ptr{Float64}, Ptr{Cvoid}))
    end
    pdf_fn = C_NULL
    if pdf isa HostLogPdf
        pdf.scalar = scalar
        pdf_fn = @cfunction(metro_pdf_trampoline, Cint, (Ptr{Float64}, Int64, Int64, Ptr{Float64}, Ptr{Cvoid}))
    end
    ctx = MetroCtx(pdf, sample_ppdf)
    p = params(pdf); p8 = ntuple(i -> i <= length(p) ? p[i] : 0.0, 8)
    chain = Array{Float64}(undef, ndim, nsamples, nchains)      # column-major == C [chain][sample][dim] (KMC_CHAIN_BY_WALKER)
    clogp = Array{Float64}(undef, nsamples, nchains)
    acc = Vector{Float64}(undef, nchains)
    nblob = device_blobs ? pdf.nblob : 0
    bl = Array{Float64}(undef, nblob, nsamples, nchains)         # blobs[chain][sample] (:117), nblob doubles each
    out = KmcMetropolisOutputs(chain=pointer(chain), chain_logp=pointer(clogp), accept_ratio=pointer(acc), blobs=(device_blobs ? pointer(bl) : C_NULL))
    st = GC.@preserve pdf sample_ppdf ctx theta step chain clogp acc bl begin
        cfg = Ref(KmcMetropolisConfig(density=density_id(pdf), params=p8, nchains=nchains, ndim=ndim, niter=niter, nburnin=nburnin,
                                      nthin=nthin, step=(isempty(step) ? Ptr{Float64}(C_NULL) : pointer(step)), seed=UInt64(seed),
                                      flags=KMC_STORE_CHAIN | KMC_STORE_LOGP | KMC_CHAIN_BY_WALKER | (device_blobs ? KMC_STORE_BLOBS : UInt32(0)), device=Int32(device), user_density=user_handle(pdf),
                                      host_logpdf=pdf_fn, host_user=pointer_from_objref(ctx), host_propose=prop_fn))
        ccall((:kmc_metropolis_run, LIB), Cint, (Ref{KmcMetropolisConfig}, Ptr{Float64}, Ref{KmcMetropolisOutputs}), cfg, theta, out)
    end
    st == 0 || error("kmc_metropolis_run failed ($st): $(last_error())")
    thetas = scalar ? [chain[1, :, c] for c in 1:nchains] :
                      [[chain[:, k, c] for k in 1:nsamples] for c in 1:nchains]
    blobs = nothing
    if device_blobs       # p0, blob0 = pdf(theta0) (:70) for init_blobs (:90), then reduce_blob! over each chain's stored series (:117)
        lp0 = Vector{Float64}(undef, nchains); b0 = Matrix{Float64}(undef, nblob, nchains)
        cfg0 = Ref(KmcConfig(density=density_id(pdf), params=p8, nwalkers=nchains + isodd(nchains), ndim=ndim, user_density=user_handle(pdf), device=Int32(device)))
        st0 = ccall((:kmc_logpdf_blob_eval_host, LIB), Cint, (Ref{KmcConfig}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64), cfg0, theta, lp0, b0, nchains)
        st0 == 0 || error("kmc_logpdf_blob_eval_host failed ($st0): $(last_error())")
        blobs = [init_blobs(b0[:, c], nsamples) for c in 1:nchains]
        for c in 1:nchains, k in 1:nsamples
            reduce_blob!(blobs[c], bl[:, k, c])
        end
    end
    return thetas, acc, [clogp[:, c] for c in 1:nchains], blobs
end

"""
    metropolis(pdf::DeviceLogPdf, sample_ppdf::Union{GaussianStep,HostProposal}, theta0; niter=10^5, nburnin=niter÷2, nthin=1, ...)

One more method of KissMCMC.metropolis, with its signature and return value (src/samplers.jl:59-77, :128), for one chain
on the GPU (a single lane: a drop-in, not a fast path -- use `metropolis_chains` for throughput).  Two plain closures,
`metropolis(pdf, sample_ppdf, theta0)`, remain KissMCMC's own CPU method.
"""
function metropolis(pdf::DeviceLogPdf, sample_ppdf::Union{GaussianStep,HostProposal}, theta0; use_progress_meter=true, kw...)
    thetas, acc, logd, blobs = metropolis_chains(pdf, sample_ppdf, [theta0]; kw...)
    return thetas[1], acc[1], logd[1], blobs === nothing ? nothing : blobs[1]             # :128
end

"""
    int_acorr(thetas; c=5, device=0)

Integrated autocorrelation time per dimension (src/analysis.jl:140-167, commented out in the reference; followed as
written) of `thetas[walker][sample]` as `emcee` / `metropolis_chains` return it.  Returns `(tau, converged)`.
"""
function int_acorr(thetas; c=5, device=0)
    @assert c > 1
    nw = length(thetas); ns = length(thetas[1]); nd = length(thetas[1][1])
    chain = Array{Float64}(undef, nd, nw, ns)              # column-major == C [sample][walker][dim]
    for w in 1:nw, k in 1:ns, d in 1:nd
        chain[d, w, k] = nd == 1 ? thetas[w][k][1] : thetas[w][k][d]
    end
    tau = Vector{Float64}(undef, nd); conv = Vector{Float64}(undef, nd)
    st = ccall((:kmc_int_acorr, LIB), Cint, (Ptr{Float64}, Int64, Int64, Int64, Float64, Cint, Ptr{Float64}, Ptr{Float64}),
               chain, ns, nw, nd, Float64(c), Cint(device), tau, conv)
    st == 0 || error("kmc_int_acorr failed ($st): $(last_error())")
    return tau, conv
end

# make_theta0s (src/samplers.jl:311-349) and squash_walkers (src/samplers.jl:372-428): KissMCMC's own, imported above.

end # module
