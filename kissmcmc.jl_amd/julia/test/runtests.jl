# KissMCMCHIP test set: KissMCMC.jl's own emcee test (reference test/emcee.jl:17-48, over the four cases without blobs of test/runtests.jl:52-79) with the
# log-densities as DEVICE densities, so that `emcee` dispatches to the HIP path, plus the README call (reference README.md:13-22).  Same shape checks, same
# acceptance bound, same truths and tolerances as the reference; the truths are closed forms here (the reference takes them from Distributions.jl, which this
# package does not depend on).  `make_theta0s` and `squash_walkers` are KissMCMC's own functions throughout.
#
# Needs Julia, KissMCMC.jl, an MI355X and the built library (python -m kissmcmc_jl_amd.build); never executed in the build environment of this repository,
# which has no julia -- tests/test_c_abi.py checks there, statically, that every name used below exists in the module.
using Test, Statistics
using KissMCMCHIP

stds(x) = sqrt.(var(x))                      # works for scalars and vectors of samples

# one case: a device density, its moments, the reference's settings (test/runtests.jl:16-35: ball_radius 0.1, 100 walkers, tolerance 0.3 unless stated)
Base.@kwdef struct DeviceCase
    name::String
    pdf
    theta0
    mean_
    std_
    median_ = nothing
    niter::Int = 10^4
    nwalkers::Int = 100
    ball_radius = 0.1
    tole::Float64 = 0.3
end

e = exp(1.0)
cases = [
    DeviceCase(name="Normal(-5, 3)", pdf=GaussianIso(-5.0, 3.0), theta0=-4.0, mean_=-5.0, std_=3.0, median_=-5.0),                       # test/runtests.jl:53-56
    DeviceCase(name="LogNormal(0, 1)", pdf=LogNormal(0.0, 1.0), theta0=0.4, mean_=sqrt(e), std_=sqrt((e - 1) * e), median_=1.0, niter=10^7),   # :57-61
    DeviceCase(name="2-D normal", pdf=MvNormal2([0.5, -0.25], [0.47 1.8; 1.8 7.0]), theta0=[0.4, 0.3], mean_=[0.5, -0.25],
               std_=[sqrt(0.47), sqrt(7.0)], niter=10^5),                                                                                      # :62-67
    DeviceCase(name="Rosenbrock / 20", pdf=Rosenbrock(1.0, 100.0, 20.0), theta0=[0.0, 0.0], mean_=[0.98, 10.3], std_=[3.1, 13.8],
               niter=10^7, tole=0.6),                                                                                                          # :68-79
]

@testset "emcee on the device" begin
    for tc in cases
        @testset "$(tc.name)" begin
            theta0s = make_theta0s(tc.theta0, tc.ball_radius, tc.pdf, tc.nwalkers)                        # KissMCMC's own (the density is callable on the host)
            samples = emcee(tc.pdf, theta0s; niter=tc.niter, use_progress_meter=false, seed=UInt64(20260105))
            @test length.(samples[1:3]) == (tc.nwalkers, tc.nwalkers, tc.nwalkers)
            @test samples[4] === nothing
            @test length(samples[1][1]) == tc.niter ÷ tc.nwalkers ÷ 2                                       # half of every walker's steps is burn-in
            thetas, accept_ratio, logdensities, blobs = squash_walkers(samples...; verbose=false)
            @test blobs === nothing
            @test length(thetas) == tc.niter ÷ 2
            @test length(logdensities) == tc.niter ÷ 2
            @test accept_ratio > 0.1
            @test all(abs.(mean(thetas) .- tc.mean_) .< abs.(tc.std_ .* tc.tole))
            @test all(abs.(stds(thetas) .- tc.std_) .< abs.(tc.std_ .* tc.tole))
            tc.median_ !== nothing && @test abs(median(thetas) - tc.median_) < abs(tc.std_ * tc.tole)
        end
    end
end

@testset "the README call" begin
    # reference README.md:13-22 with the exponential as a device density: 100 walkers, niter = 10^5
    logpdf = Exponential(1.0)
    theta0 = make_theta0s(0.5, 0.1, logpdf, 100)
    thetas, accept_ratio, logdensities = emcee(logpdf, theta0; niter=10^5, use_progress_meter=false)
    thetas, accept_ratio, logdensities = squash_walkers(thetas, accept_ratio, logdensities; verbose=false)
    @test length(thetas) == 10^5 ÷ 2
    @test all(thetas .>= 0)
    @test isapprox(mean(thetas), 1.0; atol=0.05) && isapprox(std(thetas), 1.0; atol=0.08)
    @test accept_ratio > 0.5
end

@testset "the reference's asserts" begin
    # src/samplers.jl:200-205, same conditions and messages on the device path
    pdf = GaussianIso()
    @test_throws AssertionError emcee(pdf, make_theta0s(0.0, 0.1, pdf, 10); niter=100, a_scale=1.0)
    @test_throws AssertionError emcee(pdf, make_theta0s(0.0, 0.1, pdf, 11); niter=100)
    @test_throws AssertionError emcee(pdf, make_theta0s(zeros(4), 0.1, pdf, 4); niter=100)
end

@testset "a log-density written by the caller" begin
    # the closure of src/samplers.jl:257 as a C function body, compiled at run time (CDensity), against the built-in Gaussian: same seed, same chain
    # (the built-in Gaussian multiplies by 1/sigma: the body does the same, so that both round alike and every accept decision agrees)
    body = CDensity("double s = 0; for (int i = 0; i < n; ++i) { const double t = (x[i] - p[0]) * p[1]; s += t * t; } return -0.5 * s;"; params=[-5.0, 1.0 / 3.0])
    theta0s = make_theta0s(-4.0, 0.1, GaussianIso(-5.0, 3.0), 100)
    a = emcee(body, theta0s; niter=10^4, use_progress_meter=false, seed=UInt64(7))
    b = emcee(GaussianIso(-5.0, 3.0), theta0s; niter=10^4, use_progress_meter=false, seed=UInt64(7))
    @test a[1] == b[1] && a[2] == b[2]
end
