// kmc_kernels.hpp -- the stretch-move half-step kernels (gfx950, wave64).
//
// One launch = one half-step of reference src/samplers.jl:246-273: every walker of the active
// half proposes once against a partner drawn from the complementary half, which is read-only
// for the duration of the launch.  The kernel boundary is the join of src/samplers.jl:273.
//
// State layout in HBM (row-major, like the reference's theta0s[walker][dim]):
//   pos     double [nwalkers][ndim]     first half = rows [0,h), second half = rows [h,2h)
//   logp    double [nwalkers]
//   naccept uint32 [nwalkers]
//
// Two kernels:
//   half_step_vec<Density, L, K, ITER>  ndim == 2*L*K.  A walker's 16*L*K-byte row is striped
//       over L lanes of one wave (16 B per lane per chunk), so both the own row and the randomly
//       drawn partner row are read as fully coalesced 16-B-per-lane segments; the log-pdf is a
//       cross-lane reduction; the proposal stays in registers until the accept decision and is
//       stored only on accept.  Each L-lane group walks ITER walkers with all loads issued
//       up front.
//   half_step_generic<Density>          any ndim; one walker per lane, scalar loops.  Used for
//       the reference's own 1-D/2-D cases and odd sizes.
#pragma once
#include "kmc_device.hpp"

namespace kmc {

struct HalfStepArgs {
    double*        pos;
    double*        logp;
    uint32_t*      naccept;
    const int64_t* gen_base;     // device generation counter (graph replay) or nullptr
    int64_t        gen_offset;   // generation = gen_offset + (gen_base ? *gen_base : 0)
    int64_t        nburnin;
    int64_t        nthin;
    int64_t        nsamples;     // stored-sample capacity
    int64_t        nhalf;        // h = nwalkers / 2 (global)
    int64_t        active_begin; // first active index (within the half) of this shard
    int32_t        n_active;     // number of active walkers of this shard
    int32_t        half;         // 0: update [0,h) against [h,2h); 1: swapped       (:247)
    int32_t        ndim;
    int32_t        pad_;
    DrawConsts     dc;
    DensityParams  dp;
    double*        chain;        // [nsamples][chain_rows][ndim] or nullptr          (:269)
    double*        chain_logp;   // [nsamples][chain_rows] or nullptr                (:271)
    int64_t        chain_rows;   // rows per sample slot
    int64_t        chain_row0;   // row of this launch's first active walker in a slot
    double*        msum;         // per-thread moment accumulators or nullptr
    double*        msumsq;
    int64_t        macc_stride;  // threads in the accumulator grid
};

struct Schedule {
    int64_t gen;
    bool    count;   // post burn-in: count acceptances            (:265, :285-288)
    bool    sample;  // this generation's state is a stored sample (:268)
    int64_t slot;    // its index k
};

__device__ __forceinline__ Schedule schedule_of(const HalfStepArgs& a)
{
    Schedule s;
    s.gen = a.gen_offset + (a.gen_base ? *a.gen_base : 0);
    const int64_t n = s.gen + 1 - a.nburnin;          // the reference's loop variable n (:245)
    s.count = n > 0;
    s.sample = false;
    s.slot = 0;
    if (n > 0) {
        if (a.nthin == 1) { s.sample = true; s.slot = n - 1; }
        else if (n % a.nthin == 0) { s.sample = true; s.slot = n / a.nthin - 1; }
        if (s.slot >= a.nsamples) s.sample = false;
    }
    return s;
}

template <int L>
__device__ __forceinline__ double group_sum(double v)
{
#pragma unroll
    for (int m = 1; m < L; m <<= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// ------------------------------------------------------------------------------------------
// Vector kernel.
// ------------------------------------------------------------------------------------------
template <class Dens, int L, int K, int ITER>
__global__ __launch_bounds__(256) void half_step_vec(const HalfStepArgs a)
{
    static_assert(L >= 1 && L <= 64 && (L & (L - 1)) == 0, "L must be a power of two <= 64");
    const int tid = blockIdx.x * 256 + threadIdx.x;
    const int j   = threadIdx.x & (L - 1);
    const int grp = tid / L;
    const int ndim = 2 * L * K;
    const Schedule sch = schedule_of(a);
    const uint64_t step = 2ull * (uint64_t)sch.gen + (uint64_t)a.half;
    const int64_t act0 = (int64_t)a.half * a.nhalf + a.active_begin;   // global index of active walker 0
    const int64_t oth0 = (int64_t)(1 - a.half) * a.nhalf;

    bool     valid[ITER];
    int64_t  gw[ITER];
    Draw     dr[ITER];
    double   p0[ITER];
    double2  xc[ITER][K], xo[ITER][K];

#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const int i = grp * ITER + it;
        valid[it] = i < a.n_active;
        gw[it] = act0 + (valid[it] ? i : a.n_active - 1);
        const double2* own = reinterpret_cast<const double2*>(a.pos + gw[it] * ndim);
#pragma unroll
        for (int k = 0; k < K; ++k) xc[it][k] = own[k * L + j];
        p0[it] = a.logp[gw[it]];
    }
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        dr[it] = draw_step(a.dc, step, (uint64_t)gw[it]);
        const double2* oth = reinterpret_cast<const double2*>(a.pos + (oth0 + dr[it].partner) * ndim);
#pragma unroll
        for (int k = 0; k < K; ++k) xo[it][k] = oth[k * L + j];
    }

    const bool do_mom = sch.sample && a.msum != nullptr;
    double2 ms[K], mq[K];
#pragma unroll
    for (int k = 0; k < K; ++k) { ms[k] = make_double2(0.0, 0.0); mq[k] = make_double2(0.0, 0.0); }

#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        double2 y[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {                                   // :255
            y[k].x = fma(dr[it].z, xc[it][k].x - xo[it][k].x, xo[it][k].x);
            y[k].y = fma(dr[it].z, xc[it][k].y - xo[it][k].y, xo[it][k].y);
        }
        const double S  = group_sum<L>(Dens::template frag_partial<L, K>(y, j, ndim, a.dp));
        const double p1 = Dens::finish(S, a.dp);                         // :257
        const bool acc = accept_test(dr[it], p1, p0[it]) && valid[it];   // :260
        if (acc) {                                                       // :261-265
            double2* own = reinterpret_cast<double2*>(a.pos + gw[it] * ndim);
#pragma unroll
            for (int k = 0; k < K; ++k) own[k * L + j] = y[k];
            if (j == 0) {
                a.logp[gw[it]] = p1;
                if (sch.count) atomicAdd(&a.naccept[gw[it]], 1u);
            }
        }
        if (sch.sample) {                                                // :268-271
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const double2 cur = acc ? y[k] : xc[it][k];
                if (valid[it]) {
                    ms[k].x += cur.x; ms[k].y += cur.y;
                    mq[k].x += cur.x * cur.x; mq[k].y += cur.y * cur.y;
                }
            }
            if (a.chain != nullptr && valid[it]) {
                const int64_t row = sch.slot * a.chain_rows + a.chain_row0 + (grp * ITER + it);
                double2* dst = reinterpret_cast<double2*>(a.chain + row * ndim);
#pragma unroll
                for (int k = 0; k < K; ++k) dst[k * L + j] = acc ? y[k] : xc[it][k];
            }
            if (a.chain_logp != nullptr && valid[it] && j == 0) {
                const int64_t row = sch.slot * a.chain_rows + a.chain_row0 + (grp * ITER + it);
                a.chain_logp[row] = acc ? p1 : p0[it];
            }
        }
    }
    if (do_mom) {
        double2* s = reinterpret_cast<double2*>(a.msum);
        double2* q = reinterpret_cast<double2*>(a.msumsq);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int64_t idx = (int64_t)k * a.macc_stride + tid;
            double2 sv = s[idx], qv = q[idx];
            sv.x += ms[k].x; sv.y += ms[k].y;
            qv.x += mq[k].x; qv.y += mq[k].y;
            s[idx] = sv; q[idx] = qv;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Generic kernel: one walker per lane, any ndim.
// ------------------------------------------------------------------------------------------
template <class Dens>
__global__ __launch_bounds__(256) void half_step_generic(const HalfStepArgs a)
{
    const int tid = blockIdx.x * 256 + threadIdx.x;
    if (tid >= a.n_active) return;
    const int ndim = a.ndim;
    const Schedule sch = schedule_of(a);
    const uint64_t step = 2ull * (uint64_t)sch.gen + (uint64_t)a.half;
    const int64_t gw = (int64_t)a.half * a.nhalf + a.active_begin + tid;
    const Draw dr = draw_step(a.dc, step, (uint64_t)gw);
    double* own = a.pos + gw * ndim;
    const double* oth = a.pos + ((int64_t)(1 - a.half) * a.nhalf + dr.partner) * ndim;
    const double p0 = a.logp[gw];

    typename Dens::Seq q;
    Dens::seq_init(q);
    for (int d = 0; d < ndim; ++d) {
        const double y = fma(dr.z, own[d] - oth[d], oth[d]);            // :255
        Dens::seq_add(q, y, d, a.dp);
    }
    const double p1 = Dens::seq_finish(q, ndim, a.dp);                   // :257
    const bool acc = accept_test(dr, p1, p0);                           // :260

    const bool do_mom = sch.sample && a.msum != nullptr;
    const bool do_chain = sch.sample && a.chain != nullptr;
    const int64_t row = sch.slot * a.chain_rows + a.chain_row0 + tid;
    if (acc || do_mom || do_chain) {
        for (int d = 0; d < ndim; ++d) {
            const double xcd = own[d];
            const double cur = acc ? fma(dr.z, xcd - oth[d], oth[d]) : xcd;
            if (acc) own[d] = cur;                                      // :261
            if (do_chain) a.chain[row * ndim + d] = cur;                // :269
            if (do_mom) {
                const int64_t idx = (int64_t)d * a.macc_stride + tid;
                a.msum[idx] += cur;
                a.msumsq[idx] += cur * cur;
            }
        }
    }
    if (acc) {
        a.logp[gw] = p1;                                                // :262
        if (sch.count) a.naccept[gw] += 1u;                             // :265
    }
    if (sch.sample && a.chain_logp != nullptr) a.chain_logp[row] = acc ? p1 : p0;   // :271
}

// Initial log-pdfs, src/samplers.jl:209.
template <class Dens>
__global__ __launch_bounds__(256) void logpdf_rows(const double* __restrict__ pos, double* __restrict__ logp,
                                                   int64_t nrows, int ndim, DensityParams dp)
{
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= nrows) return;
    typename Dens::Seq q;
    Dens::seq_init(q);
    for (int d = 0; d < ndim; ++d) Dens::seq_add(q, pos[r * ndim + d], d, dp);
    logp[r] = Dens::seq_finish(q, ndim, dp);
}

__global__ void bump_generation(int64_t* gen, int64_t by) { *gen += by; }

}  // namespace kmc
