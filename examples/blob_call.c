/*
 * The reference's blob test case (test/runtests.jl:80-107 through test/emcee.jl:21-45) through the C ABI alone:
 *
 *   pdf = x -> (-(x+5)^2/(2*3.0^2), blob)        # hasblob=true: the log-density AND a blob  (src/samplers.jl:150-151)
 *   samples = emcee(pdf, theta0s; niter=10^4, hasblob=true)
 *
 * with the blob computed ON THE DEVICE: a body density that also fills blob[0..m) (kmc_user_density_create_body_blob).  Here
 * m = 3 and blob = {x, x^2, the log-density}, so that the program can check the carried blobs against the stored samples:
 * blob0s[nc] = blob1 exactly when theta0s[nc] = theta1 (src/samplers.jl:261-264), stored with every sample (:270).
 *
 * Build:  gcc -O2 -Iinclude examples/blob_call.c -o blob_call -Lkissmcmc.jl_amd -lkissmcmc_hip -lm -Wl,-rpath,$PWD/kissmcmc.jl_amd
 * tests/test_gpu_deviceblobs.py builds and runs it.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "kissmcmc_hip.h"

int main(void)
{
    enum { NWALKERS = 100, NITER = 10000, NBLOB = 3 };
    kmc_user_density* ud = NULL;
    if (kmc_user_density_create_body_blob("const double t = x[0] + 5.0; const double lp = -(t * t) / (2.0 * 3.0 * 3.0);"
                                          "blob[0] = x[0]; blob[1] = x[0] * x[0]; blob[2] = lp; return lp;", NBLOB, &ud) != KMC_OK) {
        fprintf(stderr, "kmc_user_density_create_body_blob: %s\n", kmc_last_error());
        return 2;
    }
    if (kmc_user_density_nblob(ud) != NBLOB) return 2;
    kmc_config cfg = {0};
    cfg.dtype = KMC_F64;
    cfg.density = KMC_USER_DENSITY;
    cfg.user_density = ud;
    cfg.nwalkers = NWALKERS;
    cfg.ndim = 1;
    cfg.ngenerations = NITER / NWALKERS;           /* src/samplers.jl:203 */
    cfg.nburnin = (NITER / 2) / NWALKERS;          /* :190, :204 */
    cfg.nthin = 1;
    cfg.a_scale = 2.0;
    cfg.seed = 8;
    cfg.flags = KMC_CHAIN_BY_WALKER;               /* thetas[w][k], blobs[w][k]: the reference's order (:219-221, :238) */

    double theta0[NWALKERS];                        /* make_theta0s(-4.0, 0.1, pdf, 100): test/runtests.jl:88, :23 */
    srand(11);
    for (int w = 0; w < NWALKERS; ++w) {
        const double u1 = (rand() + 1.0) / ((double)RAND_MAX + 2.0), u2 = (rand() + 1.0) / ((double)RAND_MAX + 2.0);
        theta0[w] = -4.0 + 0.1 * sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
    }
    /* pdf.(theta0s): the initial log-densities and blobs (:209-210) */
    double lp0[NWALKERS], blob0[NWALKERS * NBLOB];
    if (kmc_logpdf_blob_eval_host(&cfg, theta0, lp0, blob0, NWALKERS) != KMC_OK) { fprintf(stderr, "%s\n", kmc_last_error()); return 1; }
    for (int w = 0; w < NWALKERS; ++w)
        if (blob0[w * NBLOB] != theta0[w] || blob0[w * NBLOB + 2] != lp0[w]) return 4;

    const long ns = (cfg.ngenerations - cfg.nburnin) / cfg.nthin;                  /* :234 */
    double* chain = malloc(sizeof(double) * ns * NWALKERS);
    double* clogp = malloc(sizeof(double) * ns * NWALKERS);
    double* blobs = malloc(sizeof(double) * ns * NWALKERS * NBLOB);
    double accept_ratio[NWALKERS];
    kmc_outputs out = {0};
    out.chain = chain; out.chain_logp = clogp; out.blobs = blobs; out.accept_ratio = accept_ratio;
    const kmc_status st = kmc_emcee_run(&cfg, theta0, &out);
    if (st != KMC_OK) { fprintf(stderr, "kmc_emcee_run: %s\n", kmc_last_error()); return 1; }

    long bad = 0;
    double mean = 0.0, acc = 0.0;
    for (long i = 0; i < ns * NWALKERS; ++i) {      /* i = w * ns + k */
        if (blobs[i * NBLOB] != chain[i] || blobs[i * NBLOB + 1] != chain[i] * chain[i] || blobs[i * NBLOB + 2] != clogp[i]) ++bad;
        mean += chain[i];
    }
    mean /= (double)(ns * NWALKERS);
    for (int w = 0; w < NWALKERS; ++w) acc += accept_ratio[w];
    acc /= NWALKERS;
    printf("samples %ld blobs-that-do-not-follow-their-walker %ld mean %.3f accept_ratio %.3f device_ms %.3f\n", ns * NWALKERS, bad, mean, acc, out.device_ms);
    free(chain); free(clogp); free(blobs);
    kmc_user_density_destroy(ud);
    /* Normal(-5, 3): |mean + 5| < 0.3 * 3 (test_mean_std, tolerance 0.3); accept_ratio > 0.1 (test/emcee.jl:43) */
    return (bad == 0 && ns * NWALKERS == NITER / 2 && fabs(mean + 5.0) < 0.9 && acc > 0.1) ? 0 : 3;
}
