# Times the REAL KissMCMC.emcee (the reference package) on the configurations bench.py reports, for anyone who
# has Julia and KissMCMC.jl installed.  It has NOT been executed in the environment this repository was built in
# (no Julia there: DESIGN.md §5) and no number from it is claimed anywhere; the CPU figure bench.py prints is the
# C/OpenMP restatement under oracle/ ("kind": "port"), not this.
#
#   JULIA_NUM_THREADS=auto julia bench/reference_julia.jl [C1|C2|C3] [generations]
#
# Metric: walker-steps/s = log-density evaluations per second = niter / wall time (src/samplers.jl:159: niter is the
# total number of log-density evaluations).
using KissMCMC
using Random

config = length(ARGS) >= 1 ? ARGS[1] : "C1"

if config == "C1"          # README.md:15-25, exactly
    logpdf(x::T) where {T} = x < 0 ? -convert(T, Inf) : -x
    nwalkers, gens = 100, 1000
    theta0s = make_theta0s(0.5, 0.1, logpdf, nwalkers)
elseif config == "C2"      # 65 536 walkers x 32-dim isotropic Gaussian, started at stationarity
    logpdf = x -> -0.5 * sum(abs2, x)
    nwalkers, gens = 65536, 100
    theta0s = [randn(32) for _ in 1:nwalkers]
elseif config == "C3"      # 16 384 walkers x 64-dim chained Rosenbrock / 20 (test/runtests.jl:68 at N = 2)
    logpdf = x -> -sum(100 .* (x[2:end] .- x[1:end-1] .^ 2) .^ 2 .+ (1 .- x[1:end-1]) .^ 2) / 20
    nwalkers, gens = 16384, 100
    theta0s = [0.1 .* randn(64) for _ in 1:nwalkers]
else
    error("unknown configuration $config (C1, C2 or C3)")
end
gens = length(ARGS) >= 2 ? parse(Int, ARGS[2]) : gens
niter = nwalkers * gens

emcee(logpdf, theta0s; niter=2 * nwalkers, use_progress_meter=false)          # compile
t = @elapsed begin
    thetas, accept_ratio, logdensities, blobs = emcee(logpdf, theta0s; niter=niter, use_progress_meter=false)
end
println("KissMCMC.emcee $config: $nwalkers walkers x $gens generations on $(Threads.nthreads()) threads: ",
        round(t, digits=3), " s, ", round(niter / t, sigdigits=4), " walker-steps/s, mean accept ratio ",
        round(sum(accept_ratio) / length(accept_ratio), digits=3))
