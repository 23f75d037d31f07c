# KissMCMCHIP.jl -- thin `ccall` shim over libkissmcmc_hip.so (C ABI: include/kissmcmc_hip.h).
#
# Keeps the call surface of KissMCMC.jl's emcee path (reference src/samplers.jl:188-216, :311-349,
# :372-428): `emcee(pdf, theta0s; niter, nburnin, nthin, a_scale, ...)`, `make_theta0s`,
# `squash_walkers`.  `pdf` is one of the menu densities below or an `ExprDensity` (evaluated on the
# device), or ANY Julia callable: that one stays on the host (`HostLogPdf`, KMC_HOST_DENSITY) and is
# called on each half-step's batch of proposals while the moves, draws, accept test and storage run
# on the GPU.  There is no CPU fallback for the sampler itself.
#
# NOT EXECUTED in the build environment (no julia binary there).  Every call it makes is mirrored
# 1:1 by the Python ctypes host (kissmcmc.jl_amd/_lib.py, api.py), which is what the tests drive.
module KissMCMCHIP

export emcee, make_theta0s, squash_walkers, metropolis, metropolis_chains, GaussianStep, int_acorr, GaussianIso, Exponential, Rosenbrock, LogNormal, MvNormal2, ExprDensity, HostLogPdf

using Statistics: mean, median, std
using LinearAlgebra: inv

const LIB = get(ENV, "KMC_LIB_PATH", joinpath(@__DIR__, "..", "libkissmcmc_hip.so"))

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
    handle::Ptr{Cvoid}; p::Vector{Float64}
    function ExprDensity(term::String, pair::Union{String,Nothing}=nothing; params=Float64[])
        h = Ref{Ptr{Cvoid}}(C_NULL)
        st = ccall((:kmc_user_density_create, LIB), Cint, (Cstring, Cstring, Ref{Ptr{Cvoid}}), term, pair === nothing ? "" : pair, h)
        st == 0 || error("kmc_user_density_create failed: $(last_error())")
        d = new(h[], collect(Float64, params))
        finalizer(x -> ccall((:kmc_user_density_destroy, LIB), Cvoid, (Ptr{Cvoid},), x.handle), d)
        return d
    end
end
density_id(::ExprDensity) = Cint(100); params(d::ExprDensity) = d.p
user_handle(d::ExprDensity) = d.handle
user_handle(::DeviceLogPdf) = C_NULL
function (d::ExprDensity)(x)
    xs = collect(Float64, x isa Number ? [x] : x); out = Ref(0.0)
    p8 = ntuple(i -> i <= length(d.p) ? d.p[i] : 0.0, 8)
    cfg = Ref(KmcConfig(0, Cint(100), p8, 2, length(xs), 0, 0, 1, 2.0, UInt64(0), 0, 0, 0, 1, d.handle, 0, 0, C_NULL, C_NULL, C_NULL))
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

# ---- C structs (layout checked against the header by tests/test_c_abi.py on the Python mirror) ---
struct KmcConfig
    dtype::Int32; density::Int32
    params::NTuple{8,Float64}
    nwalkers::Int64; ndim::Int64; ngenerations::Int64; nburnin::Int64; nthin::Int64
    a_scale::Float64; seed::UInt64
    flags::UInt32; device::Int32; shard_rank::Int32; shard_count::Int32
    user_density::Ptr{Cvoid}
    island_gens::Int32; island_size::Int32     # KMC_ISLANDS (opt-in island mode); 0 = defaults
    host_logpdf::Ptr{Cvoid}; host_user::Ptr{Cvoid}   # KMC_HOST_DENSITY callback and its context
    host_accepted::Ptr{Cvoid}                         # KMC_HOST_DENSITY: accept outcomes per half-step (blobs), or NULL
end

mutable struct KmcOutputs
    chain::Ptr{Float64}; chain_logp::Ptr{Float64}; accept_ratio::Ptr{Float64}; naccept::Ptr{Int64}
    final_pos::Ptr{Float64}; final_logp::Ptr{Float64}; sum::Ptr{Float64}; sumsq::Ptr{Float64}
    nmoment::Int64; nsamples::Int64; device_ms::Float64
end

last_error() = unsafe_string(ccall((:kmc_last_error, LIB), Cstring, ()))

"""
    emcee(pdf::DeviceLogPdf, theta0s; niter=10^5, nburnin=niter÷2, nthin=1, a_scale=2.0,
          use_progress_meter=true, hasblob=false, init_blobs, reduce_blob!, seed=rand(UInt64), device=0, dtype=:f64)

Same meaning as KissMCMC.emcee (src/samplers.jl:188-197); returns
`(thetas, accept_ratio, logdensities, blobs)` with `thetas[w][k]` (src/samplers.jl:292).
`hasblob=true` needs a host closure as `pdf` (blobs are host objects).
"""
function emcee(pdf::DeviceLogPdf, theta0s; niter=10^5, nburnin=niter ÷ 2, nthin=1, a_scale=2.0,
               use_progress_meter=true, hasblob=false,
               init_blobs=(blob0, nsamples) -> sizehint!(typeof(blob0)[], nsamples),      # init_output_vector :80-85
               reduce_blob! =(blobs, blob) -> push!(blobs, blob),                         # :196
               seed=rand(UInt64), device=0, dtype=:f64)     # dtype=:f32: float rows on the device (KMC_F32), built-in densities
    hasblob && !(pdf isa HostLogPdf && pdf.hasblob) &&
        error("hasblob=true needs a host closure as pdf: device densities return the log-pdf alone")
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
    cfg = Ref(KmcConfig(dtype == :f32 ? 1 : 0, density_id(pdf), p8, nwalkers, ndim, niter_walker, nburnin_walker, nthin,
                        a_scale, UInt64(seed), 0x3, Int32(device), 0, 1, user_handle(pdf), 0, 0, host_fn, host_ctx, acc_fn))   # flags: STORE_CHAIN | STORE_LOGP
    chain = Array{Float64}(undef, ndim, nwalkers, nsamples)
    clogp = Array{Float64}(undef, nwalkers, nsamples)
    acc = Vector{Float64}(undef, nwalkers)
    out = KmcOutputs(pointer(chain), pointer(clogp), pointer(acc), C_NULL, C_NULL, C_NULL, C_NULL, C_NULL, 0, 0, 0.0)
    st = GC.@preserve pdf theta chain clogp acc ccall((:kmc_emcee_run, LIB), Cint,
                                                   (Ref{KmcConfig}, Ptr{Float64}, Ref{KmcOutputs}), cfg, theta, out)
    st == 0 || error("kmc_emcee_run failed ($st): $(last_error())")
    thetas = scalar ? [[chain[1, w, k] for k in 1:nsamples] for w in 1:nwalkers] :
                      [[chain[:, w, k] for k in 1:nsamples] for w in 1:nwalkers]
    logdensities = [[clogp[w, k] for k in 1:nsamples] for w in 1:nwalkers]
    return thetas, acc, logdensities, (pdf isa HostLogPdf && pdf.hasblob) ? pdf.blobs : nothing   # :292
end
# arbitrary closure: evaluated on the host
emcee(pdf, theta0s; hasblob=false, kw...) = emcee(HostLogPdf(pdf; hasblob=hasblob), theta0s; hasblob=hasblob, kw...)

# ---- many-chain Metropolis: metropolis / _metropolis, src/samplers.jl:59-128 ---------------------
"Symmetric proposal `theta -> scale .* randn(n) .+ theta` (the one all reference tests use, test/runtests.jl:54,59,64,75)."
struct GaussianStep; scale::Vector{Float64}; end
GaussianStep(c::Real) = GaussianStep([Float64(c)])

struct KmcMetropolisConfig
    dtype::Int32; density::Int32
    params::NTuple{8,Float64}
    nchains::Int64; ndim::Int64; niter::Int64; nburnin::Int64; nthin::Int64
    step::Ptr{Float64}; seed::UInt64
    flags::UInt32; device::Int32
    user_density::Ptr{Cvoid}
end

mutable struct KmcMetropolisOutputs
    chain::Ptr{Float64}; chain_logp::Ptr{Float64}; accept_ratio::Ptr{Float64}; naccept::Ptr{Int64}
    final_pos::Ptr{Float64}; final_logp::Ptr{Float64}; chain_sum::Ptr{Float64}; chain_sumsq::Ptr{Float64}
    nsamples::Int64; device_ms::Float64
end

"""
    metropolis_chains(pdf::DeviceLogPdf, sample_ppdf::GaussianStep, theta0s; niter=10^5, nburnin=niter÷2, nthin=1, seed, device)

One independent Metropolis chain (src/samplers.jl:96-126) per element of `theta0s`, one chain per GPU lane;
`niter`/`nburnin` count steps per chain.  Returns `(thetas, accept_ratio, logdensities, nothing)` shaped like
`emcee`'s output (`thetas[chain][sample]`), so `squash_walkers` applies.
"""
function metropolis_chains(pdf::DeviceLogPdf, sample_ppdf::GaussianStep, theta0s; niter=10^5, nburnin=niter ÷ 2, nthin=1,
                           hasblob=false, seed=rand(UInt64), device=0)
    hasblob && error("hasblob=true is not supported by the HIP samplers")
    pdf isa HostLogPdf && error("the many-chain Metropolis kernel needs a device log-density (menu or ExprDensity)")
    nchains = length(theta0s); scalar = theta0s[1] isa Number; ndim = length(theta0s[1])
    nsamples = niter > nburnin ? (niter - nburnin) ÷ nthin : 0                            # :88
    theta = Matrix{Float64}(undef, ndim, nchains)          # column-major [dim, chain] == C row-major [chain][dim]
    for c in 1:nchains, d in 1:ndim
        theta[d, c] = scalar ? theta0s[c] : theta0s[c][d]                                # :68 deep copy
    end
    step = length(sample_ppdf.scale) == 1 ? fill(sample_ppdf.scale[1], ndim) : copy(sample_ppdf.scale)
    @assert length(step) == ndim
    p = params(pdf); p8 = ntuple(i -> i <= length(p) ? p[i] : 0.0, 8)
    chain = Array{Float64}(undef, ndim, nchains, nsamples)
    clogp = Array{Float64}(undef, nchains, nsamples)
    acc = Vector{Float64}(undef, nchains)
    out = KmcMetropolisOutputs(pointer(chain), pointer(clogp), pointer(acc), C_NULL, C_NULL, C_NULL, C_NULL, C_NULL, 0, 0.0)
    st = GC.@preserve pdf theta step chain clogp acc begin
        cfg = Ref(KmcMetropolisConfig(0, density_id(pdf), p8, nchains, ndim, niter, nburnin, nthin, pointer(step), UInt64(seed),
                                      0x3, Int32(device), user_handle(pdf)))                 # flags: STORE_CHAIN | STORE_LOGP
        ccall((:kmc_metropolis_run, LIB), Cint, (Ref{KmcMetropolisConfig}, Ptr{Float64}, Ref{KmcMetropolisOutputs}), cfg, theta, out)
    end
    st == 0 || error("kmc_metropolis_run failed ($st): $(last_error())")
    thetas = scalar ? [[chain[1, c, k] for k in 1:nsamples] for c in 1:nchains] :
                      [[chain[:, c, k] for k in 1:nsamples] for c in 1:nchains]
    return thetas, acc, [[clogp[c, k] for k in 1:nsamples] for c in 1:nchains], nothing
end

"""
    metropolis(pdf::DeviceLogPdf, sample_ppdf::GaussianStep, theta0; niter=10^5, nburnin=niter÷2, nthin=1, ...)

KissMCMC.metropolis' signature and return value (src/samplers.jl:59-77, :128) for one chain (a single lane:
a drop-in, not a fast path -- use `metropolis_chains` for throughput).
"""
function metropolis(pdf::DeviceLogPdf, sample_ppdf::GaussianStep, theta0; use_progress_meter=true, kw...)
    thetas, acc, logd, _ = metropolis_chains(pdf, sample_ppdf, [theta0]; kw...)
    return thetas[1], acc[1], logd[1], nothing                                           # :128
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

"src/samplers.jl:311-349 (host side, runs once)."
function make_theta0s(theta0::T, ball_radius, pdf, nwalkers; ball_radius_halfing_steps=7, ntries=100, hasblob=false) where T
    npara = length(theta0)
    if ball_radius isa Number && !(T <: Number)
        ball_radius = ones(npara) * ball_radius
    end
    @assert length(ball_radius) == npara
    theta0s = T[]
    for i = 1:nwalkers
        for k = 1:ball_radius_halfing_steps
            ball_radius *= 1 / 2^(k - 1)
            for _ = 1:ntries
                tmp = npara == 1 ? theta0 .+ randn() .* ball_radius : theta0 .+ randn(npara) .* ball_radius
                p0 = hasblob ? pdf(tmp)[1] : pdf(tmp)                                   # :333-337
                if p0 > -Inf
                    push!(theta0s, tmp)
                    break
                end
            end
            length(theta0s) == i && break
        end
        length(theta0s) == i || error("Could not find suitable initial theta.  PDF is zero in too many places inside ball.")
    end
    return theta0s
end

"src/samplers.jl:372-428 (host post-processing)."
function squash_walkers(thetas, accept_ratio, logdensities=nothing, blobs=nothing;
                        drop_low_accept_ratio=false, drop_fact=2, verbose=true, order=false, merge_blobs! =append!)
    nwalkers = length(accept_ratio)
    walkers2keep = if drop_low_accept_ratio
        ma, sa = median(accept_ratio), std(accept_ratio)
        verbose && println("Median accept ratio is $ma, standard deviation is $sa\n")
        [nc for nc in 1:nwalkers if !(accept_ratio[nc] <= ma - drop_fact * sa)]
    else
        collect(1:nwalkers)
    end
    t = reduce(vcat, thetas[walkers2keep])
    l = logdensities === nothing ? nothing : reduce(vcat, logdensities[walkers2keep])
    b = nothing
    if blobs !== nothing                                                                  # :408-413
        b = deepcopy(blobs[walkers2keep[1]])
        for w in walkers2keep[2:end]
            merge_blobs!(b, blobs[w])
        end
    end
    if order
        ns = length(thetas[1])
        perm = sortperm(repeat(1:ns, length(walkers2keep)))
        t = t[perm]
        l === nothing || (l = l[perm])
        b === nothing || (b = b[perm])
    end
    return t, mean(accept_ratio[walkers2keep]), l, b
end

end # module
