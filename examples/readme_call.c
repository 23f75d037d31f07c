/*
 * The reference README's call sequence (README.md:15-27) through the C ABI alone -- no Python, no torch:
 *
 *   logpdf(x) = x<0 ? -Inf : -x
 *   thetase, accept_ratioe = emcee(logpdf, make_theta0s(0.5, 0.1, logpdf, 100), niter=10^5)
 *   thetas, accept_ratio   = squash_walkers(thetase, accept_ratioe)
 *
 * Build:  gcc -O2 -Iinclude examples/readme_call.c -o readme_call -Lkissmcmc.jl_amd -lkissmcmc_hip -lm \
 *             -Wl,-rpath,$PWD/kissmcmc.jl_amd
 * This is what a `ccall` binding does; tests/test_gpu_dropin.py builds and runs it.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "kissmcmc_hip.h"

int main(void)
{
    enum { NWALKERS = 100, NITER = 100000 };
    kmc_config cfg = {0};
    cfg.dtype = KMC_F64;
    cfg.density = KMC_EXPONENTIAL;                 /* README.md:15 */
    cfg.params[0] = 1.0;
    cfg.nwalkers = NWALKERS;
    cfg.ndim = 1;
    cfg.ngenerations = NITER / NWALKERS;           /* niter_walker   = niter ÷ nwalkers        src/samplers.jl:203 */
    cfg.nburnin = (NITER / 2) / NWALKERS;          /* nburnin_walker = (niter ÷ 2) ÷ nwalkers  src/samplers.jl:190,204 */
    cfg.nthin = 1;
    cfg.a_scale = 2.0;
    cfg.seed = 2024;
    cfg.flags = KMC_STORE_CHAIN | KMC_MOMENTS;
    if (kmc_validate(&cfg) != KMC_OK) { fprintf(stderr, "%s\n", kmc_last_error()); return 2; }

    /* make_theta0s(0.5, 0.1, logpdf, 100): 0.5 + 0.1 randn(), redrawn while logpdf = -Inf  (src/samplers.jl:311-349) */
    double theta0[NWALKERS];
    srand(7);
    for (int w = 0; w < NWALKERS; ++w) {
        do {
            const double u1 = (rand() + 1.0) / ((double)RAND_MAX + 2.0), u2 = (rand() + 1.0) / ((double)RAND_MAX + 2.0);
            theta0[w] = 0.5 + 0.1 * sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
        } while (theta0[w] < 0.0);
    }

    const long nsamples = (cfg.ngenerations - cfg.nburnin) / cfg.nthin;            /* src/samplers.jl:234 */
    double* chain = malloc(sizeof(double) * nsamples * NWALKERS);                /* [sample][walker][1] */
    double accept_ratio[NWALKERS], sum[1], sumsq[1];
    kmc_outputs out = {0};
    out.chain = chain;
    out.accept_ratio = accept_ratio;
    out.sum = sum;
    out.sumsq = sumsq;
    const kmc_status st = kmc_emcee_run(&cfg, theta0, &out);
    if (st != KMC_OK) { fprintf(stderr, "kmc_emcee_run: %s\n", kmc_last_error()); return 1; }

    /* squash_walkers: all walkers' samples in one vector, mean accept ratio  (src/samplers.jl:372-428) */
    double mean = 0.0, acc = 0.0;
    for (long i = 0; i < nsamples * NWALKERS; ++i) mean += chain[i];
    mean /= (double)(nsamples * NWALKERS);
    for (int w = 0; w < NWALKERS; ++w) acc += accept_ratio[w];
    acc /= NWALKERS;
    const double var = sumsq[0] / (double)out.nmoment - (sum[0] / (double)out.nmoment) * (sum[0] / (double)out.nmoment);
    printf("samples %ld mean %.4f var %.4f accept_ratio %.4f device_ms %.3f\n", nsamples * NWALKERS, mean, var, acc, out.device_ms);
    free(chain);
    /* exponential(1): mean 1, variance 1; acceptance about 0.745 */
    return (fabs(mean - 1.0) < 0.1 && fabs(var - 1.0) < 0.25 && fabs(acc - 0.745) < 0.03 && nsamples * NWALKERS == NITER / 2) ? 0 : 3;
}
