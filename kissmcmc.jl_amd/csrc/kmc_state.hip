// kmc_state.hip -- a sampler's state in and out: the initial ensemble and its log-pdfs (reference src/samplers.jl:198,
// :209-210), the device-side initial ball (:311-349), checkpoint / resume, the read-outs (positions, log-pdfs, accept
// statistics :291, streaming moments), the deal of dealt sub-ensembles, and the one-shot `emcee` call (:188-293).
#include <signal.h>

#include <cmath>
#include <cstdlib>

#define KMC_DEFINE_STATE_KERNELS
#include "kmc_sampler.hpp"

using namespace kmc;
using namespace kmc_host;

namespace kmc_host {
// Initial log-pdfs of the rows in d_pos (src/samplers.jl:209-210) into d_logp, on the sampler's stream.  KMC_F32: the
// log-pdf kernels read double rows, so the float rows are widened (exactly) into a scratch buffer first.
__global__ __launch_bounds__(256) void widen_rows(const float* src, double* dst, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = (double)src[i];
}
__global__ __launch_bounds__(256) void narrow_rows(const double* src, float* dst, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = (float)src[i];
}

kmc_status eval_initial_logp(kmc_sampler* s, double* logp_out = nullptr)      // (logp_out: somewhere else than d_logp -- a caller that only wants the blobs)
{
    const size_t nw = (size_t)s->nrows, nd = (size_t)s->cfg.ndim;
    double* rows = s->d_pos;
    double* scratch = nullptr;
    if (s->f32) {
        const int64_t n = (int64_t)(nw * (size_t)s->ld);
        HIP_TRY(hipMalloc((void**)&scratch, (size_t)n * sizeof(double)));
        hipLaunchKernelGGL(widen_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s->stream, reinterpret_cast<const float*>(s->d_pos), scratch, n);
        rows = scratch;
    }
    const LogpdfArgs la{rows, logp_out ? logp_out : s->d_logp, (int64_t)nw, (int32_t)nd, (int32_t)s->ld, s->dp, s->d_blob};   // (blobs of the initial evaluations, :209-210)
    hipError_t e = hipSuccess;
    if (s->user) {
        e = launch_module(s->uk.logpdf, (unsigned)((nw + 255) / 256), 256u, s->stream, la);
    } else {
        hipLaunchKernelGGL(s->logpdf_fn, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, s->stream, la);
        e = hipGetLastError();
    }
    if (scratch) {
        if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
        (void)hipFree(scratch);
    }
    HIP_TRY(e);
    return KMC_OK;
}

// Everything set_positions does after the rows are in place: initial log-pdfs (src/samplers.jl:209-210)
// unless they are supplied, counters, accumulators, finiteness check.
kmc_status reset_run_state(kmc_sampler* s, bool eval_logp, int64_t generation, uint32_t klast_value)
{
    const size_t nw = (size_t)s->nrows;
    if (eval_logp) KMC_TRY(eval_initial_logp(s));
    std::vector<double> lp(nw);
    HIP_TRY(copy_sync(lp.data(), s->d_logp, nw * sizeof(double), hipMemcpyDeviceToHost, s->stream));
    if (s->d_msum) {
        HIP_TRY(hipMemsetAsync(s->d_msum, 0, (size_t)s->macc_elems * sizeof(double), s->stream));
        HIP_TRY(hipMemsetAsync(s->d_msumsq, 0, (size_t)s->macc_elems * sizeof(double), s->stream));
        if (s->d_klast) HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)s->d_klast, (int)klast_value, nw, s->stream));
        if (s->d_mcnt) HIP_TRY(hipMemsetAsync(s->d_mcnt, 0, 2 * (size_t)s->mring_waves * sizeof(uint32_t), s->stream));
        if (s->d_isum) {
            const size_t ne = (size_t)s->nislands * 4 * (size_t)s->island_K;
            HIP_TRY(hipMemsetAsync(s->d_isum, 0, ne * sizeof(double), s->stream));
            HIP_TRY(hipMemsetAsync(s->d_isumsq, 0, ne * sizeof(double), s->stream));
        }
        s->carry_sum.clear(); s->carry_sumsq.clear();
    }
    HIP_TRY(hipStreamSynchronize(s->stream));
    s->pos2_current = false;
    s->fused_cur = 0;                  // (the caller's state is in d_pos / d_logp: whatever a failed run left in copy 1 is void)
    s->generation = generation;
    s->launches = 0;
    s->have_run_events = false;
    if (s->stream_chain) { HIP_TRY(hipStreamSynchronize(s->copy_stream)); s->blocks_copied = 0; s->blocks_waited = 0; s->flushed_done = -1; }
    for (size_t w = 0; w < nw; ++w)
        if (!std::isfinite(lp[w])) {
            s->positions_set = false;
            return fail(KMC_ERR_NONFINITE_LOGP, "walker " + std::to_string(w) + " has a non-finite initial log-pdf");
        }
    s->positions_set = true;
    return KMC_OK;
}

}  // namespace kmc_host
// Device-side make_theta0s: src/samplers.jl:311-349 (see init_ball in kmc_kernels.hpp).
KMC_EXPORT kmc_status kmc_sampler_init_ball(kmc_sampler* s, const double* theta0, const double* ball_radius,
                                            uint64_t seed, int halving_steps, int ntries)
{
    if (!s || !theta0 || !ball_radius || halving_steps < 1 || ntries < 1) return fail(KMC_ERR_BAD_ARG, "bad argument");
    if (s->host_eval) return fail(KMC_ERR_UNSUPPORTED, "kmc_sampler_init_ball evaluates the density on the device; with KMC_HOST_DENSITY build the ball on the host");
    if (s->push) return fail(KMC_ERR_UNSUPPORTED, "kmc_sampler_init_ball fills this rank's rows only; with KMC_P2P_PUSH use kmc_sampler_set_positions (the peers' copies must be filled too)");
    HIP_TRY(hipSetDevice(s->cfg.device));
    HIP_TRY(hipStreamSynchronize(s->stream));
    const size_t nd = (size_t)s->cfg.ndim;
    double* d_par = nullptr;
    unsigned long long* d_fail = nullptr;
    HIP_TRY(hipMalloc((void**)&d_par, 2 * nd * sizeof(double)));
    hipError_t e = hipMalloc((void**)&d_fail, sizeof(unsigned long long));
    if (e == hipSuccess) e = copy_sync(d_par, theta0, nd * sizeof(double), hipMemcpyHostToDevice, s->stream);
    if (e == hipSuccess) e = copy_sync(d_par + nd, ball_radius, nd * sizeof(double), hipMemcpyHostToDevice, s->stream);
    if (e == hipSuccess) e = fill_sync(d_fail, 0, sizeof(unsigned long long), s->stream);
    const size_t nelem = (size_t)s->nrows * (size_t)s->ld;
    double* d_ball = s->d_pos;                  // KMC_F32: the ball is drawn in double, then rounded into the float rows
    if (s->f32 && e == hipSuccess) e = hipMalloc((void**)&d_ball, nelem * sizeof(double));
    if (e == hipSuccess) e = fill_sync(d_ball, 0, nelem * sizeof(double), s->stream);
    InitBallFn fn = s->user ? nullptr : init_ball_fn(s->cfg.density);
    const int pieces = s->p2p ? 2 : 1;
    for (int piece = 0; piece < pieces && e == hipSuccess; ++piece) {
        InitBallArgs a{};
        const int64_t rows = s->p2p ? s->h_loc : s->nrows;
        a.pos = d_ball + (size_t)piece * (size_t)s->h_loc * (size_t)s->ld;
        a.logp = s->d_logp + (size_t)piece * (size_t)s->h_loc;
        a.theta0 = d_par; a.radius = d_par + nd;
        a.nrows = rows;
        a.row_walker0 = s->p2p ? (int64_t)piece * s->h + s->active_begin
                               : (s->cfg.deal_count > 0 ? (int64_t)s->cfg.deal_rank * s->nrows : 0);   // rows of ONE global ball
        a.ndim = (int32_t)nd; a.ld = (int32_t)s->ld;
        a.halving_steps = halving_steps; a.ntries = ntries;
        a.seed_lo = (uint32_t)seed; a.seed_hi = (uint32_t)(seed >> 32);
        a.dp = s->dp;
        a.fail = d_fail;
        a.blob = s->d_blob;
        const unsigned grid = (unsigned)((rows + 255) / 256);
        if (s->user) e = launch_module(s->uk.init_ball, grid, 256u, s->stream, a);
        else { hipLaunchKernelGGL(fn, dim3(grid), dim3(256), 0, s->stream, a); e = hipGetLastError(); }
    }
    unsigned long long nfail = 0;
    if (s->f32 && e == hipSuccess) {
        hipLaunchKernelGGL(narrow_rows, dim3((unsigned)((nelem + 255) / 256)), dim3(256), 0, s->stream, d_ball, reinterpret_cast<float*>(s->d_pos), (int64_t)nelem);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
    if (e == hipSuccess) e = copy_sync(&nfail, d_fail, sizeof(nfail), hipMemcpyDeviceToHost, s->stream);
    (void)hipFree(d_par);
    (void)hipFree(d_fail);
    if (s->f32) (void)hipFree(d_ball);
    HIP_TRY(e);
    if (nfail != 0) {
        s->positions_set = false;
        return fail(KMC_ERR_NONFINITE_LOGP, "Could not find suitable initial theta.  PDF is zero in too many places inside ball. (" +
                                            std::to_string(nfail) + " walkers)");
    }
    HIP_TRY(hipMemsetAsync(s->d_naccept, 0, (size_t)s->nrows * sizeof(uint32_t), s->stream));
    HIP_TRY(hipMemsetAsync(s->d_gen, 0, 64, s->stream));
    if (s->p2p) {
        HIP_TRY(fill_sync(s->d_flags, 0, 4096, s->stream));
        HIP_TRY(fill_sync(s->d_err, 0, 64, s->stream));
    }
    s->dev_gen = 0;
    s->moment_base = 0;
    if (s->d_ids) {
        hipLaunchKernelGGL(deal_init_ids, dim3((unsigned)((s->nrows + 255) / 256)), dim3(256), 0, s->stream, s->d_ids, s->nrows,
                           (uint32_t)((uint64_t)s->cfg.deal_rank * (uint64_t)s->nrows));
        HIP_TRY(hipGetLastError());
    }
    return reset_run_state(s, /*eval_logp=*/s->f32, 0, 0u);      // KMC_F32: the log-pdfs of the rows as rounded
}

// Checkpoint / resume: restore (positions, log-pdfs, acceptance counters, generation).  The random
// stream is a pure function of (seed, generation, walker), so the continued run is bit-identical to an
// uninterrupted one.  Moments and the chain restart at the restored generation.
KMC_EXPORT kmc_status kmc_sampler_set_state(kmc_sampler* s, const double* pos_host, const double* logp_host,
                                            const int64_t* naccept_host, int64_t generation)
{
    if (!s || !pos_host || !logp_host || generation < 0) return fail(KMC_ERR_BAD_ARG, "bad argument");
    // (stored blobs are chain storage too: after a resume past burn-in kmc_sampler_get_blobs would return rows never written)
    if (s->d_chain || s->d_chain_logp || s->d_chain_blob)
        return fail(KMC_ERR_UNSUPPORTED, "kmc_sampler_set_state: not with chain / blob storage (download the chain before checkpointing)");
    if (s->p2p && s->cfg.shard_count > 1)
        return fail(KMC_ERR_UNSUPPORTED, "kmc_sampler_set_state: P2P progress flags restart at 0; restore is single-GPU for now");
    HIP_TRY(hipSetDevice(s->cfg.device));
    HIP_TRY(hipStreamSynchronize(s->stream));
    const size_t nw = (size_t)s->nrows;
    HIP_TRY(upload_rows(s, s->d_pos, pos_host, nw));
    HIP_TRY(copy_sync(s->d_logp, logp_host, nw * sizeof(double), hipMemcpyHostToDevice, s->stream));
    std::vector<uint32_t> na(nw, 0u);
    if (naccept_host) for (size_t i = 0; i < nw; ++i) na[i] = (uint32_t)naccept_host[i];
    HIP_TRY(copy_sync(s->d_naccept, na.data(), nw * sizeof(uint32_t), hipMemcpyHostToDevice, s->stream));
    if (s->p2p) {
        HIP_TRY(fill_sync(s->d_flags, 0, 4096, s->stream));
        HIP_TRY(fill_sync(s->d_err, 0, 64, s->stream));
    }
    if (s->d_blob) {
        // the blobs of the restored positions: evaluated again (same kernel and order as the initial evaluation: same bits);
        // the restored log-pdfs stay as given
        double* scratch = nullptr;
        HIP_TRY(hipMalloc((void**)&scratch, nw * sizeof(double)));
        const kmc_status bst = eval_initial_logp(s, scratch);
        const hipError_t be = hipStreamSynchronize(s->stream);
        (void)hipFree(scratch);
        KMC_TRY(bst);
        HIP_TRY(be);
    }
    s->generation = generation;            // the device counter follows at the next graph replay
    const int64_t done = samples_done(s);
    s->moment_base = done;
    return reset_run_state(s, /*eval_logp=*/false, generation, (uint32_t)done);
}

KMC_EXPORT kmc_status kmc_sampler_set_positions(kmc_sampler* s, const double* theta_host)
{
    if (!s || !theta_host) return fail(KMC_ERR_BAD_ARG, "null argument");
    reinstall_abort_backtrace();                                 // (diagnostics: somebody may have replaced the handler)
    HIP_TRY(hipSetDevice(s->cfg.device));
    const size_t nw = (size_t)s->nrows, nd = (size_t)s->cfg.ndim;
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (!s->p2p) {
        HIP_TRY(upload_rows(s, s->d_pos, theta_host, nw));   // :198 (caller's array untouched)
    } else {
        // theta_host is the GLOBAL ensemble; keep this shard's slice of each half: local rows
        // [0,h_loc) = global [begin, begin+h_loc), local [h_loc,2h_loc) = global [h+begin, ...)
        const size_t hl = (size_t)s->h_loc;
        HIP_TRY(upload_rows(s, s->d_pos, theta_host + (size_t)s->active_begin * nd, hl));
        HIP_TRY(upload_rows(s, s->d_pos + hl * (size_t)s->ld, theta_host + ((size_t)s->h + (size_t)s->active_begin) * nd, hl));
        if (s->push) {      // the local copies of the other shards (block 1 + q = rank q's rows)
            for (int q = 0; q < s->cfg.shard_count; ++q) {
                if (q == s->cfg.shard_rank) continue;
                double* blk = s->d_pos + (size_t)(1 + q) * nw * (size_t)s->ld;
                HIP_TRY(upload_rows(s, blk, theta_host + (size_t)q * hl * nd, hl));
                HIP_TRY(upload_rows(s, blk + hl * (size_t)s->ld, theta_host + ((size_t)s->h + (size_t)q * hl) * nd, hl));
            }
        }
        HIP_TRY(fill_sync(s->d_flags, 0, 4096, s->stream));     // callers barrier across ranks before running
        HIP_TRY(fill_sync(s->d_err, 0, 64, s->stream));
    }
    if (s->host_eval) {                                          // :209-210, on the caller's thread
        std::vector<double> lp0(nw);
        if (s->cfg.host_logpdf(theta_host, (int64_t)nw, (int64_t)nd, lp0.data(), s->cfg.host_user) != 0) {
            s->positions_set = false;
            return fail(KMC_ERR_BAD_ARG, "the host log-pdf callback failed on the initial ensemble");
        }
        HIP_TRY(copy_sync(s->d_logp, lp0.data(), nw * sizeof(double), hipMemcpyHostToDevice, s->stream));
    } else {
        KMC_TRY(eval_initial_logp(s));                           // :209-210
    }
    std::vector<double> lp(nw);
    HIP_TRY(copy_sync(lp.data(), s->d_logp, nw * sizeof(double), hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipMemsetAsync(s->d_naccept, 0, nw * sizeof(uint32_t), s->stream));
    HIP_TRY(hipMemsetAsync(s->d_gen, 0, 64, s->stream));
    if (s->d_msum) {
        HIP_TRY(hipMemsetAsync(s->d_msum, 0, (size_t)s->macc_elems * sizeof(double), s->stream));
        HIP_TRY(hipMemsetAsync(s->d_msumsq, 0, (size_t)s->macc_elems * sizeof(double), s->stream));
        if (s->d_klast) HIP_TRY(hipMemsetAsync(s->d_klast, 0, nw * sizeof(uint32_t), s->stream));
        if (s->d_mcnt) HIP_TRY(hipMemsetAsync(s->d_mcnt, 0, 2 * (size_t)s->mring_waves * sizeof(uint32_t), s->stream));
        if (s->d_isum) {
            const size_t ne = (size_t)s->nislands * 4 * (size_t)s->island_K;
            HIP_TRY(hipMemsetAsync(s->d_isum, 0, ne * sizeof(double), s->stream));
            HIP_TRY(hipMemsetAsync(s->d_isumsq, 0, ne * sizeof(double), s->stream));
        }
        s->carry_sum.clear(); s->carry_sumsq.clear();
    }
    if (s->d_ids) {                                              // dealt sub-ensembles: slot i holds global walker r S + i
        hipLaunchKernelGGL(deal_init_ids, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, s->stream, s->d_ids, (int64_t)nw,
                           (uint32_t)((uint64_t)s->cfg.deal_rank * (uint64_t)nw));
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipStreamSynchronize(s->stream));
    s->pos2_current = false;
    s->fused_cur = 0;                  // (the caller's state is in d_pos / d_logp: whatever a failed run left in copy 1 is void)
    s->generation = 0;
    s->dev_gen = 0;
    s->moment_base = 0;
    s->launches = 0;
    s->have_run_events = false;
    if (s->stream_chain) { HIP_TRY(hipStreamSynchronize(s->copy_stream)); s->blocks_copied = 0; s->blocks_waited = 0; s->flushed_done = -1; }
    for (size_t w = 0; w < nw; ++w)
        if (!std::isfinite(lp[w])) {
            s->positions_set = false;
            return fail(KMC_ERR_NONFINITE_LOGP,
                        "walker " + std::to_string(w) + " has a non-finite initial log-pdf");
        }
    s->positions_set = true;
    return KMC_OK;
}

KMC_EXPORT kmc_status kmc_sampler_get_positions(kmc_sampler* s, double* host)
{
    if (!s || !host) return fail(KMC_ERR_BAD_ARG, "null argument");
    HIP_TRY(hipSetDevice(s->cfg.device));
    HIP_TRY(hipStreamSynchronize(s->stream));
    KMC_TRY(check_p2p_err(s));
    HIP_TRY(download_rows(s, host, s->d_pos, (size_t)s->nrows));
    return KMC_OK;
}

KMC_EXPORT kmc_status kmc_sampler_get_logp(kmc_sampler* s, double* host)
{
    if (!s || !host) return fail(KMC_ERR_BAD_ARG, "null argument");
    HIP_TRY(hipSetDevice(s->cfg.device));
    HIP_TRY(hipStreamSynchronize(s->stream));
    KMC_TRY(check_p2p_err(s));
    HIP_TRY(copy_sync(host, s->d_logp, (size_t)s->nrows * sizeof(double), hipMemcpyDeviceToHost, s->stream));
    return KMC_OK;
}

KMC_EXPORT kmc_status kmc_sampler_get_naccept(kmc_sampler* s, int64_t* host)
{
    if (!s || !host) return fail(KMC_ERR_BAD_ARG, "null argument");
    HIP_TRY(hipSetDevice(s->cfg.device));
    HIP_TRY(hipStreamSynchronize(s->stream));
    KMC_TRY(check_p2p_err(s));
    std::vector<uint32_t> tmp((size_t)s->nrows);
    HIP_TRY(copy_sync(tmp.data(), s->d_naccept, tmp.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream));
    for (size_t i = 0; i < tmp.size(); ++i) host[i] = (int64_t)tmp[i];
    return KMC_OK;
}

KMC_EXPORT kmc_status kmc_sampler_get_accept_ratio(kmc_sampler* s, double* host)
{
    if (!s || !host) return fail(KMC_ERR_BAD_ARG, "null argument");
    std::vector<int64_t> na((size_t)s->nrows);
    KMC_TRY(kmc_sampler_get_naccept(s, na.data()));
    const double denom = (double)(s->generation - s->cfg.nburnin);   // :291 (0 -> inf/nan like the reference)
    for (size_t i = 0; i < na.size(); ++i) host[i] = (double)na[i] / denom;
    return KMC_OK;
}

KMC_EXPORT kmc_status kmc_sampler_get_moments(kmc_sampler* s, double* sum, double* sumsq, int64_t* n)
{
    if (!s) return fail(KMC_ERR_BAD_ARG, "null sampler");
    if (!s->d_msum) return fail(KMC_ERR_BAD_ARG, "sampler was created without KMC_MOMENTS");
    HIP_TRY(hipSetDevice(s->cfg.device));
    if (s->fused) {
        // one launch per generation: per-walker sums [nwalkers][ld] (laid out like the rows), added up in walker order -- or, lane-striped rows with
        // K == 2 and L = 8 / 16 / 32, per-wave accumulators in the vector kernels' transposed layout (kmc_kernels.hpp: FoldT), added up in wave order
        HIP_TRY(hipStreamSynchronize(s->stream));
        const int64_t nd = s->cfg.ndim, nw = s->cfg.nwalkers, ld = s->ld;
        std::vector<double> S((size_t)nd, 0.0), Q((size_t)nd, 0.0);
        if (s->fused_fold) {
            const int L = s->fused_L, NVL = 8 * L / 64;
            const int64_t nwaves = s->nislands;
            std::vector<double> hs((size_t)(nwaves * NVL * 64));
            HIP_TRY(copy_sync(hs.data(), s->d_isum, hs.size() * sizeof(double), hipMemcpyDeviceToHost, s->stream));
            for (int64_t w = 0; w < nwaves; ++w)
                for (int r = 0; r < NVL; ++r)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int b3 = (lane >> 3) & 1, b4 = (lane >> 4) & 1, b5 = (lane >> 5) & 1;
                        const int v = L == 8 ? 4 * b3 + 2 * b4 + b5 : L == 16 ? 4 * b4 + 2 * b5 + r : 4 * b5 + r;
                        const int64_t d = 2 * ((int64_t)((v >> 1) & 1) * L + (lane & (L - 1))) + (v & 1);
                        if (d >= nd) continue;
                        const double x = hs[(size_t)((w * NVL + r) * 64 + lane)];
                        if (v >> 2) Q[(size_t)d] += x; else S[(size_t)d] += x;
                    }
        } else {
            std::vector<double> hs((size_t)(ld * nw)), hq((size_t)(ld * nw));
            HIP_TRY(copy_sync(hs.data(), s->d_isum, hs.size() * sizeof(double), hipMemcpyDeviceToHost, s->stream));
            HIP_TRY(copy_sync(hq.data(), s->d_isumsq, hq.size() * sizeof(double), hipMemcpyDeviceToHost, s->stream));
            for (int64_t w = 0; w < nw; ++w)
                for (int64_t d = 0; d < nd; ++d) { S[(size_t)d] += hs[(size_t)(w * ld + d)]; Q[(size_t)d] += hq[(size_t)(w * ld + d)]; }
        }
        if (s->fused_L > 0) {
            // the lane-striped form credits a value when it is replaced (sojourn weights, like the two-launch kernels): every walker's CURRENT value
            // still stands for the samples taken since its last move -- credited here, on the host, leaving the device state as it is
            std::vector<double> hp((size_t)(ld * nw));
            std::vector<uint32_t> kl((size_t)nw);
            HIP_TRY(copy_sync(hp.data(), s->d_pos, hp.size() * sizeof(double), hipMemcpyDeviceToHost, s->stream));
            HIP_TRY(copy_sync(kl.data(), s->d_klast, kl.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream));
            const uint32_t done = (uint32_t)samples_done(s);
            for (int64_t w = 0; w < nw; ++w) {
                const double wgt = (double)(done - kl[(size_t)w]);
                if (wgt == 0.0) continue;
                for (int64_t d = 0; d < nd; ++d) {
                    const double x = hp[(size_t)(w * ld + d)];
                    S[(size_t)d] += x * wgt;
                    Q[(size_t)d] += (x * x) * wgt;
                }
            }
        }
        for (int64_t d = 0; d < nd; ++d) {
            if (sum) sum[d] = S[(size_t)d];
            if (sumsq) sumsq[d] = Q[(size_t)d];
        }
        if (n) *n = (samples_done(s) - s->moment_base) * s->nlocal;
        return KMC_OK;
    }
    if (s->islands || s->resident) {
        HIP_TRY(hipStreamSynchronize(s->stream));
        const int64_t nd = s->cfg.ndim;
        const size_t per = 4 * (size_t)s->island_K, ne = (size_t)s->nislands * per;
        std::vector<double> hs(ne), hq(ne);
        HIP_TRY(copy_sync(hs.data(), s->d_isum, ne * sizeof(double), hipMemcpyDeviceToHost, s->stream));
        HIP_TRY(copy_sync(hq.data(), s->d_isumsq, ne * sizeof(double), hipMemcpyDeviceToHost, s->stream));
        std::vector<double> S((size_t)nd, 0.0), Q((size_t)nd, 0.0);
        for (int64_t b = 0; b < s->nislands; ++b)
            for (size_t e = 0; e < per; ++e)
                if ((int64_t)e < nd) { S[e] += hs[(size_t)b * per + e]; Q[e] += hq[(size_t)b * per + e]; }
        for (int64_t d = 0; d < nd; ++d) {
            if (sum) sum[d] = S[d];
            if (sumsq) sumsq[d] = Q[d];
        }
        if (n) *n = (samples_done(s) - s->moment_base) * s->nlocal;
        return KMC_OK;
    }
    KMC_TRY(flush_moments_now(s));
    HIP_TRY(hipStreamSynchronize(s->stream));
    KMC_TRY(check_p2p_err(s));
    const int64_t nd = s->cfg.ndim;
    std::vector<double> hs((size_t)s->macc_elems), hq((size_t)s->macc_elems);
    HIP_TRY(copy_sync(hs.data(), s->d_msum, hs.size() * sizeof(double), hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(copy_sync(hq.data(), s->d_msumsq, hq.size() * sizeof(double), hipMemcpyDeviceToHost, s->stream));
    std::vector<double> S((size_t)nd, 0.0), Q((size_t)nd, 0.0);
    if (s->plan.vec && s->plan.K == 2 && (s->plan.L == 8 || s->plan.L == 16 || s->plan.L == 32)) {
        // transposed fold (kmc_kernels.hpp, FoldT): every lane of a wave owns NVL of the wave's 8 L / 64 * 64 sums
        const int L = s->plan.L, NVL = 8 * L / 64;
        const int64_t nwaves = s->macc_stride / 64;
        for (int64_t w = 0; w < nwaves; ++w)
            for (int r = 0; r < NVL; ++r)
                for (int lane = 0; lane < 64; ++lane) {
                    const int b3 = (lane >> 3) & 1, b4 = (lane >> 4) & 1, b5 = (lane >> 5) & 1;
                    const int v = L == 8 ? 4 * b3 + 2 * b4 + b5 : L == 16 ? 4 * b4 + 2 * b5 + r : 4 * b5 + r;
                    const int64_t d = 2 * ((int64_t)((v >> 1) & 1) * L + (lane & (L - 1))) + (v & 1);
                    if (d >= nd) continue;
                    const double x = hs[(size_t)((w * NVL + r) * 64 + lane)];
                    if (v >> 2) Q[(size_t)d] += x; else S[(size_t)d] += x;
                }
    } else if (s->plan.vec) {
        const int L = s->plan.L, K = s->plan.K;
        for (int k = 0; k < K; ++k)
            for (int64_t t = 0; t < s->macc_stride; ++t) {
                const int64_t d0 = 2 * ((int64_t)k * L + (t % L));
                const int64_t idx = 2 * ((int64_t)k * s->macc_stride + t);
                if (d0 < nd) { S[d0] += hs[idx]; Q[d0] += hq[idx]; }
                if (d0 + 1 < nd) { S[d0 + 1] += hs[idx + 1]; Q[d0 + 1] += hq[idx + 1]; }
            }
    } else if (s->user && s->uk.staged != nullptr && staged_tile_moments((int)nd)) {
        // half_step_staged: one accumulator row [ld] per wave
        for (int64_t w = 0; w < s->macc_stride / 64; ++w)
            for (int64_t d = 0; d < nd; ++d) {
                S[d] += hs[w * s->ld + d];
                Q[d] += hq[w * s->ld + d];
            }
    } else {
        for (int64_t d = 0; d < nd; ++d)
            for (int64_t t = 0; t < s->macc_stride; ++t) {
                S[d] += hs[d * s->macc_stride + t];
                Q[d] += hq[d * s->macc_stride + t];
            }
    }
    if (!s->carry_sum.empty())                    // what the sampler credited while it ran one launch per generation (unfuse)
        for (int64_t d = 0; d < nd; ++d) { S[(size_t)d] += s->carry_sum[(size_t)d]; Q[(size_t)d] += s->carry_sumsq[(size_t)d]; }
    for (int64_t d = 0; d < nd; ++d) {
        if (sum) sum[d] = S[d];
        if (sumsq) sumsq[d] = Q[d];
    }
    if (n) *n = (samples_done(s) - s->moment_base) * s->nlocal;
    return KMC_OK;
}

// Blobs of a body density with blobs (include/kissmcmc_hip.h): the current ones, and the stored series (KMC_STORE_BLOBS).
KMC_EXPORT kmc_status kmc_sampler_get_blobs(kmc_sampler* s, double* current, double* stored, int by_walker)
{
    if (!s) return fail(KMC_ERR_BAD_ARG, "null sampler");
    if (s->nblob == 0) return fail(KMC_ERR_BAD_ARG, "this sampler's density returns no blobs (kmc_user_density_create_body_blob)");
    HIP_TRY(hipSetDevice(s->cfg.device));
    HIP_TRY(hipStreamSynchronize(s->stream));
    const size_t nb = (size_t)s->nblob;
    if (current) HIP_TRY(copy_sync(current, s->d_blob, (size_t)s->nrows * nb * sizeof(double), hipMemcpyDeviceToHost, s->stream));
    if (stored) {
        if (!s->d_chain_blob && s->nsamples > 0) return fail(KMC_ERR_BAD_ARG, "sampler was created without KMC_STORE_BLOBS");
        const int64_t K = samples_done(s);
        if (K > 0) {
            if (by_walker) KMC_TRY(download_by_walker(s->d_chain_blob, false, s->nlocal, (int64_t)nb, (int64_t)nb, K, stored, s->stream));
            else HIP_TRY(copy_sync(stored, s->d_chain_blob, (size_t)K * (size_t)s->nlocal * nb * sizeof(double), hipMemcpyDeviceToHost, s->stream));
        }
    }
    return KMC_OK;
}

KMC_EXPORT uint64_t kmc_deal_seed(uint64_t seed, int32_t deal_rank) { return deal_seed(seed, deal_rank); }

KMC_EXPORT kmc_status kmc_deal_perm(uint64_t seed, int64_t epoch, int32_t deal_rank, int64_t S, int64_t* A, int64_t* C)
{
    if (!A || !C || S < 2 || epoch < 0 || deal_rank < 0) return fail(KMC_ERR_BAD_ARG, "bad argument");
    deal_perm(seed, epoch, deal_rank, S, A, C);
    return KMC_OK;
}

DealArgs deal_args(kmc_sampler* s, void* buf)
{
    DealArgs a{};
    a.pos = s->d_pos; a.logp = s->d_logp; a.naccept = s->d_naccept; a.ids = s->d_ids;
    a.buf = static_cast<double*>(buf);
    a.S = s->nrows; a.A = 1; a.C = 0;
    a.ndim = (int32_t)s->cfg.ndim; a.ld = (int32_t)s->ld;
    return a;
}

KMC_EXPORT kmc_status kmc_sampler_deal_pack(kmc_sampler* s, int64_t epoch, void* send_dev)
{
    if (!s || !send_dev || epoch < 0) return fail(KMC_ERR_BAD_ARG, "bad argument");
    if (!s->d_ids) return fail(KMC_ERR_BAD_ARG, "sampler was created without kmc_config.deal_count");
    if (!s->positions_set) return fail(KMC_ERR_BAD_ARG, "kmc_sampler_set_positions has not succeeded yet");
    HIP_TRY(hipSetDevice(s->cfg.device));
    KMC_TRY(flush_moments_now(s));          // the accumulators are per slot: settle them before walkers change slots
    DealArgs a = deal_args(s, send_dev);
    deal_perm(s->user_seed, epoch, s->cfg.deal_rank, s->nrows, &a.A, &a.C);
    const int64_t n = a.S * ((int64_t)a.ndim + 2);
    hipLaunchKernelGGL(deal_pack, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s->stream, a);
    HIP_TRY(hipGetLastError());
    return KMC_OK;
}

KMC_EXPORT kmc_status kmc_sampler_deal_unpack(kmc_sampler* s, const void* recv_dev)
{
    if (!s || !recv_dev) return fail(KMC_ERR_BAD_ARG, "bad argument");
    if (!s->d_ids) return fail(KMC_ERR_BAD_ARG, "sampler was created without kmc_config.deal_count");
    HIP_TRY(hipSetDevice(s->cfg.device));
    const DealArgs a = deal_args(s, const_cast<void*>(recv_dev));
    const int64_t n = a.S * ((int64_t)a.ndim + 2);
    hipLaunchKernelGGL(deal_unpack, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s->stream, a);
    HIP_TRY(hipGetLastError());
    return KMC_OK;
}

KMC_EXPORT kmc_status kmc_sampler_get_walker_ids(kmc_sampler* s, int64_t* host)
{
    if (!s || !host) return fail(KMC_ERR_BAD_ARG, "null argument");
    HIP_TRY(hipSetDevice(s->cfg.device));
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (!s->d_ids) { for (int64_t i = 0; i < s->nrows; ++i) host[i] = i; return KMC_OK; }
    std::vector<uint32_t> tmp((size_t)s->nrows);
    HIP_TRY(copy_sync(tmp.data(), s->d_ids, tmp.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream));
    for (size_t i = 0; i < tmp.size(); ++i) host[i] = (int64_t)tmp[i];
    return KMC_OK;
}

// dealt sub-ensembles, after kmc_sampler_set_state: which global walker each slot holds (a function of the restored generation's
// epoch alone: replay kmc_deal_perm on the host, distributed.deal_slot_ids)
KMC_EXPORT kmc_status kmc_sampler_set_walker_ids(kmc_sampler* s, const int64_t* host)
{
    if (!s || !host) return fail(KMC_ERR_BAD_ARG, "null argument");
    if (!s->d_ids) return fail(KMC_ERR_BAD_ARG, "sampler was created without kmc_config.deal_count");
    HIP_TRY(hipSetDevice(s->cfg.device));
    HIP_TRY(hipStreamSynchronize(s->stream));
    std::vector<uint32_t> tmp((size_t)s->nrows);
    for (size_t i = 0; i < tmp.size(); ++i) {
        if (host[i] < 0 || host[i] >= (int64_t)1 << 32) return fail(KMC_ERR_BAD_ARG, "walker index out of range");
        tmp[i] = (uint32_t)host[i];
    }
    HIP_TRY(copy_sync(s->d_ids, tmp.data(), tmp.size() * sizeof(uint32_t), hipMemcpyHostToDevice, s->stream));
    return KMC_OK;
}

KMC_EXPORT kmc_status kmc_emcee_run(const kmc_config* cfg, const double* theta0, kmc_outputs* out)
{
    if (!cfg || !theta0 || !out) return fail(KMC_ERR_BAD_ARG, "null argument");
    kmc_config c = *cfg;
    if (out->chain) c.flags |= KMC_STORE_CHAIN;
    if (out->chain_logp) c.flags |= KMC_STORE_LOGP;
    if (out->blobs) c.flags |= KMC_STORE_BLOBS;
    if (out->sum || out->sumsq) c.flags |= KMC_MOMENTS;
    c.shard_rank = 0;
    c.shard_count = 1;
    const bool has_blobs = c.density == KMC_USER_DENSITY && c.user_density && static_cast<const kmc_user_density*>(c.user_density)->nblob > 0;
    if ((out->chain || out->chain_logp) && !has_blobs && c.dtype == KMC_F64 && !(c.flags & (KMC_ISLANDS | KMC_P2P)) && c.nthin > 0 && c.ngenerations > c.nburnin) {
        // a chain that does not fit the device is streamed to the caller's buffers while sampling (KMC_STREAM_CHAIN)
        size_t free_b = 0, total_b = 0;
        const size_t ns = (size_t)((c.ngenerations - c.nburnin) / c.nthin), nw_ = (size_t)c.nwalkers;
        const size_t need = ns * nw_ * ((out->chain ? (size_t)(c.ndim + (c.ndim & 1)) * sizeof(double) : 0) + (out->chain_logp ? sizeof(double) : 0));
        if (hipSetDevice(c.device) == hipSuccess && hipMemGetInfo(&free_b, &total_b) == hipSuccess && need > free_b / 10 * 8) c.flags |= KMC_STREAM_CHAIN;
        else (void)hipGetLastError();
    }
    kmc_sampler* s = nullptr;
    KMC_TRY(kmc_sampler_create(&c, &s));
    kmc_status st = KMC_OK;
    if (s->stream_chain) st = kmc_sampler_set_chain_host(s, out->chain, out->chain_logp);
    if (st == KMC_OK) st = kmc_sampler_set_positions(s, theta0);
    if (st == KMC_OK) st = kmc_sampler_run(s, c.ngenerations);
    if (st == KMC_OK) st = kmc_sampler_sync(s);
    if (st == KMC_OK) st = kmc_sampler_last_run_ms(s, &out->device_ms);
    if (st == KMC_OK && (out->chain || out->chain_logp))
        st = (c.flags & KMC_CHAIN_BY_WALKER) ? kmc_sampler_get_chain_by_walker(s, out->chain, out->chain_logp) : kmc_sampler_get_chain(s, out->chain, out->chain_logp);
    if (st == KMC_OK && out->blobs) st = kmc_sampler_get_blobs(s, nullptr, out->blobs, (c.flags & KMC_CHAIN_BY_WALKER) ? 1 : 0);
    if (st == KMC_OK && out->accept_ratio) st = kmc_sampler_get_accept_ratio(s, out->accept_ratio);
    if (st == KMC_OK && out->naccept) st = kmc_sampler_get_naccept(s, out->naccept);
    if (st == KMC_OK && out->final_pos) st = kmc_sampler_get_positions(s, out->final_pos);
    if (st == KMC_OK && out->final_logp) st = kmc_sampler_get_logp(s, out->final_logp);
    out->nmoment = 0;
    if (st == KMC_OK && (out->sum || out->sumsq)) st = kmc_sampler_get_moments(s, out->sum, out->sumsq, &out->nmoment);
    out->nsamples = s->nsamples;
    kmc_sampler_destroy(s);
    return st;
}

// ------------------------------------------------------------------------------------------
// stateless op
// ------------------------------------------------------------------------------------------
KMC_EXPORT kmc_status kmc_logpdf_eval(const kmc_config* cfg, const double* pos_dev, double* logp_dev,
                                      int64_t nrows, void* hip_stream)
{
    if (!cfg || !pos_dev || !logp_dev || nrows < 0) return fail(KMC_ERR_BAD_ARG, "bad argument");
    DensityParams dp;
    KMC_TRY(digest_params(*cfg, &dp));
    if (nrows == 0) return KMC_OK;
    const LogpdfArgs la{pos_dev, logp_dev, nrows, (int32_t)cfg->ndim, (int32_t)cfg->ndim, dp, nullptr};
    const unsigned grid = (unsigned)((nrows + 255) / 256);
    if (cfg->density == KMC_USER_DENSITY) {
        UserKernels uk;
        KMC_TRY(load_user(static_cast<kmc_user_density*>(cfg->user_density), false, 0, 0, 0, false, &uk, 0, false, 0, false, cfg->ndim));
        const hipError_t e = launch_module(uk.logpdf, grid, 256u, (hipStream_t)hip_stream, la);
        if (e == hipSuccess) (void)hipStreamSynchronize((hipStream_t)hip_stream);   // (uk's hold on the module ends with this scope)
        HIP_TRY(e);
        return KMC_OK;
    }
    if (cfg->density == KMC_HOST_DENSITY) return fail(KMC_ERR_UNSUPPORTED, "KMC_HOST_DENSITY is evaluated by the caller, not on the device");
    HalfStepFn v, g;
    LogpdfFn lp = nullptr;
    if (!lookup(cfg->density, 0, 0, 1, false, false, false, &v, &g, &lp)) return fail(KMC_ERR_BAD_ARG, "unknown density id");
    hipLaunchKernelGGL(lp, dim3(grid), dim3(256), 0, (hipStream_t)hip_stream, la);
    HIP_TRY(hipGetLastError());
    return KMC_OK;
}

// Host-buffer convenience: logp[i] = log pdf(pos[i]) for dense host rows (used by the host shims for
// make_theta0s' `pdf(theta) > -Inf` test with runtime-compiled densities).
KMC_EXPORT kmc_status kmc_logpdf_eval_host(const kmc_config* cfg, const double* pos_host, double* logp_host, int64_t nrows)
{
    if (!cfg || !pos_host || !logp_host || nrows < 0) return fail(KMC_ERR_BAD_ARG, "bad argument");
    if (nrows == 0) return KMC_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        (void)hipGetLastError();
        return fail(KMC_ERR_NO_DEVICE, "no HIP device visible");
    }
    HIP_TRY(hipSetDevice(cfg->device));
    ScopedStream ss;
    HIP_TRY(ss.create());
    double *dpos = nullptr, *dlp = nullptr;
    const size_t nb = (size_t)nrows * (size_t)cfg->ndim * sizeof(double);
    HIP_TRY(hipMalloc(&dpos, nb));
    hipError_t e = hipMalloc(&dlp, (size_t)nrows * sizeof(double));
    kmc_status st = KMC_OK;
    if (e == hipSuccess) e = copy_sync(dpos, pos_host, nb, hipMemcpyHostToDevice, ss.st);
    if (e == hipSuccess) st = kmc_logpdf_eval(cfg, dpos, dlp, nrows, ss.st);
    if (e == hipSuccess && st == KMC_OK) e = hipStreamSynchronize(ss.st);
    if (e == hipSuccess && st == KMC_OK) e = copy_sync(logp_host, dlp, (size_t)nrows * sizeof(double), hipMemcpyDeviceToHost, ss.st);
    (void)hipFree(dpos);
    (void)hipFree(dlp);
    if (st != KMC_OK) return st;
    HIP_TRY(e);
    return KMC_OK;
}

// ... and the blobs with them (a body density with blobs; the reference's `pdf.(theta0s)` under hasblob=true, src/samplers.jl:209-210,
// and make_theta0s' `pdf(theta)[1]`, :336): blob_host [nrows][nblob].
KMC_EXPORT kmc_status kmc_logpdf_blob_eval_host(const kmc_config* cfg, const double* pos_host, double* logp_host, double* blob_host, int64_t nrows)
{
    if (!cfg || !pos_host || !logp_host || !blob_host || nrows < 0) return fail(KMC_ERR_BAD_ARG, "bad argument");
    const kmc_user_density* ud = cfg->density == KMC_USER_DENSITY ? static_cast<const kmc_user_density*>(cfg->user_density) : nullptr;
    if (!ud || ud->nblob == 0) return fail(KMC_ERR_BAD_ARG, "kmc_logpdf_blob_eval_host needs a body density with blobs (kmc_user_density_create_body_blob)");
    if (nrows == 0) return KMC_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { (void)hipGetLastError(); return fail(KMC_ERR_NO_DEVICE, "no HIP device visible"); }
    HIP_TRY(hipSetDevice(cfg->device));
    DensityParams dp;
    KMC_TRY(digest_params(*cfg, &dp));
    ScopedStream ss;
    HIP_TRY(ss.create());
    const size_t nb = (size_t)nrows * (size_t)cfg->ndim * sizeof(double), bb = (size_t)nrows * (size_t)ud->nblob * sizeof(double);
    char* buf = nullptr;
    HIP_TRY(hipMalloc((void**)&buf, nb + (size_t)nrows * sizeof(double) + bb));
    double* dpos = reinterpret_cast<double*>(buf);
    double* dlp = dpos + (size_t)nrows * (size_t)cfg->ndim;
    double* dbl = dlp + nrows;
    UserKernels uk;
    kmc_status st = load_user(const_cast<kmc_user_density*>(ud), false, 0, 0, 0, false, &uk, 0, false, 0, false, cfg->ndim);
    hipError_t e = hipSuccess;
    if (st == KMC_OK) {
        e = copy_sync(dpos, pos_host, nb, hipMemcpyHostToDevice, ss.st);
        const LogpdfArgs la{dpos, dlp, nrows, (int32_t)cfg->ndim, (int32_t)cfg->ndim, dp, dbl};
        if (e == hipSuccess) e = launch_module(uk.logpdf, (unsigned)((nrows + 255) / 256), 256u, ss.st, la);
        if (e == hipSuccess) e = hipStreamSynchronize(ss.st);
        if (e == hipSuccess) e = copy_sync(logp_host, dlp, (size_t)nrows * sizeof(double), hipMemcpyDeviceToHost, ss.st);
        if (e == hipSuccess) e = copy_sync(blob_host, dbl, bb, hipMemcpyDeviceToHost, ss.st);
    }
    (void)hipFree(buf);
    if (st != KMC_OK) return st;
    HIP_TRY(e);
    return KMC_OK;
}

// The same on the chain a sampler holds on the device (KMC_STORE_CHAIN; the samples stored so far): no host round trip.
KMC_EXPORT kmc_status kmc_sampler_int_acorr(kmc_sampler* s, double c, double* tau, double* converged)
{
    if (!s) return fail(KMC_ERR_BAD_ARG, "null sampler");
    if (!s->d_chain) return fail(KMC_ERR_BAD_ARG, "sampler was created without KMC_STORE_CHAIN");
    if (s->stream_chain) return fail(KMC_ERR_UNSUPPORTED, "KMC_STREAM_CHAIN: the chain is on the host; use kmc_int_acorr on it");
    if (s->ld != s->cfg.ndim) return fail(KMC_ERR_UNSUPPORTED, "odd ndim: rows are padded on the device; use kmc_int_acorr on the downloaded chain");
    if (s->f32) return fail(KMC_ERR_UNSUPPORTED, "KMC_F32: the device chain is float; use kmc_int_acorr on the downloaded chain");
    HIP_TRY(hipSetDevice(s->cfg.device));
    HIP_TRY(hipStreamSynchronize(s->stream));
    const int64_t ns = samples_done(s);
    KMC_TRY(int_acorr_check(ns, s->nlocal, s->cfg.ndim, c, tau, converged));
    return int_acorr_device(s->d_chain, ns, s->nlocal, s->cfg.ndim, c, tau, converged);
}

