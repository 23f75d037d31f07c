// kmc_metropolis.hpp -- many independent Metropolis chains, one chain per lane (gfx950).
//
// The reference's `metropolis` / `_metropolis` (src/samplers.jl:59-128) is one serial Markov chain:
// nothing to parallelise inside it.  The data-parallel form is MANY chains at once -- every lane
// runs the reference's loop (src/samplers.jl:96-126) on its own chain, state in registers, the whole
// iteration range of a launch inside the kernel (chains never talk to each other, so there is no
// join and no kernel boundary per step).  Same log-density menu as the emcee path (the Seq
// interface: element by element in index order, so log-pdfs match a scalar CPU loop).
//
// Proposal: the symmetric Gaussian step every reference test uses, `theta -> c .* randn(n) .+ theta`
// (test/runtests.jl:54,59,64,75), with one scale per dimension.
//
// Random stream (the build's contract; the reference never seeds): Philox4x32-10,
//   key     = {seed_lo ^ 0x4d455452 ("METR"), seed_hi}
//   counter = {it_lo, it_hi, chain, block}      it = 0-based iteration, chain = chain index
//   block 0:  words w0,w1 -> normal pair for dimensions 0,1;  (w2 << 20 | w3 >> 12) -> the accept uniform
//   block b>=1: words w0,w1 -> dimensions 4b-2, 4b-1;  w2,w3 -> dimensions 4b, 4b+1
//   normal pair from words (a, b): u1 = (a + 1/2) 2^-32, u2 = (b + 1/2) 2^-32,
//       r = sqrt(-2 log u1), n0 = r cos(2 pi u2), n1 = r sin(2 pi u2)           (Box-Muller)
// A chain's result is a pure function of (seed, chain index, inputs): independent of launch geometry
// and of how the iteration range is cut into launches.
#pragma once
#include "kmc_device.hpp"

namespace kmc {

struct MetropolisArgs {
    double*       pos;         // [nchains][ndim] theta0 of every chain (in/out)
    double*       logp;        // [nchains] p0
    uint32_t*     naccept;     // [nchains] accepted steps with n > 0                  (:106, :124)
    double*       chain;       // [nsamples][nchains][ndim] or nullptr                 (:117)
    double*       chain_logp;  // [nsamples][nchains] or nullptr                       (:119)
    double*       csum;        // [nchains][ndim] per-chain sum over the stored samples, or nullptr
    double*       csumsq;
    const double* step;        // [ndim] proposal scale per dimension
    double*       xt;          // any-ndim kernel: the chains' state, DIMENSION-major [ndim][nchains] (coalesced over lanes)
    double*       yt;          // any-ndim kernel: the proposals, [ndim][nchains]
    double*       st1;         // any-ndim kernel: per-chain sums, [ndim][nchains], or nullptr
    double*       st2;
    int64_t       nchains;
    int64_t       it0, it1;    // iterations of this launch (0-based); reference n = it + 1 - nburnin (:96)
    int64_t       nburnin, nthin, nsamples;
    int64_t       cnt0;        // steps with n > 0 since the last stored sample, at it0
    int64_t       slot0;       // samples stored before it0
    int32_t       ndim;
    uint32_t      seed_lo, seed_hi;
    DensityParams dp;
    // blobs of a body density (BodyBlobDensity, NB doubles per evaluation; the any-ndim kernel): hasblob=true, src/samplers.jl:70-72
    double*       blob;        // [nchains][NB]: blob0 of every chain (:72, :103)
    double*       chain_blob;  // [nsamples][nchains][NB] (reduce_blob!, :117) or nullptr
};

__device__ __forceinline__ void normal_pair(uint32_t a, uint32_t b, double& n0, double& n1)
{
    const double u1 = ((double)a + 0.5) * 0x1.0p-32;
    const double u2 = ((double)b + 0.5) * 0x1.0p-32;
    const double r = sqrt(-2.0 * log_pos_normal(u1));     // u1 in (0,1): positive, normal
    double sn, cs;
    sincospi(2.0 * u2, &sn, &cs);        // = sincos(2 pi u2) without the argument reduction (agrees to rounding)
    n0 = r * cs;
    n1 = r * sn;
}

__device__ __forceinline__ double normal_first(uint32_t a, uint32_t b)
{   // n0 of normal_pair alone (1-D chains)
    const double u1 = ((double)a + 0.5) * 0x1.0p-32;
    const double u2 = ((double)b + 0.5) * 0x1.0p-32;
    return sqrt(-2.0 * log_pos_normal(u1)) * cospi(2.0 * u2);
}

// ND > 0: ndim <= ND, the chain lives in registers.
template <class Dens, int ND>
__device__ __forceinline__ void metropolis_chains_body(const MetropolisArgs& a)
{
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= a.nchains) return;
    const int ndim = a.ndim;
    double x[ND], y[ND], sc[ND], s1[ND], s2[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d) {
        x[d]  = d < ndim ? a.pos[c * ndim + d] : 0.0;
        sc[d] = d < ndim ? a.step[d] : 0.0;
        s1[d] = (a.csum != nullptr && d < ndim) ? a.csum[c * ndim + d] : 0.0;
        s2[d] = (a.csum != nullptr && d < ndim) ? a.csumsq[c * ndim + d] : 0.0;
    }
    double   p0  = a.logp[c];
    uint32_t na  = a.naccept[c];
    int64_t  cnt = a.cnt0, slot = a.slot0;
    int64_t  n   = a.it0 + 1 - a.nburnin;                                    // :96
    const uint32_t k0 = a.seed_lo ^ 0x4d455452u, k1 = a.seed_hi;
    for (int64_t it = a.it0; it < a.it1; ++it, ++n) {
        const U4 w = philox4x32_10((uint32_t)it, (uint32_t)((uint64_t)it >> 32), (uint32_t)c, 0u, k0, k1);
        double nr[ND + 4];
        if constexpr (ND == 1) nr[0] = normal_first(w.x, w.y);
        else normal_pair(w.x, w.y, nr[0], nr[1]);
        if constexpr (ND > 2) {
#pragma unroll
            for (int b = 1; 4 * b - 2 < ND; ++b) {
                if (4 * b - 2 < ndim) {
                    const U4 v = philox4x32_10((uint32_t)it, (uint32_t)((uint64_t)it >> 32), (uint32_t)c, (uint32_t)b, k0, k1);
                    normal_pair(v.x, v.y, nr[4 * b - 2], nr[4 * b - 1]);
                    normal_pair(v.z, v.w, nr[4 * b], nr[4 * b + 1]);
                }
            }
        }
        typename Dens::Seq q;
        Dens::seq_init(q);
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            if (d < ndim) {
                y[d] = fma(sc[d], nr[d], x[d]);                              // :98  theta1 = sample_ppdf(theta0)
                Dens::seq_add(q, y[d], d, a.dp);
            }
        }
        const double p1 = Dens::seq_finish(q, ndim, a.dp);                   // :99
        const uint64_t kk = ((uint64_t)w.z << 20) | (uint64_t)(w.w >> 12);
        const double lu = log_pos_normal(((double)kk + 0.5) * 0x1.0p-52);
        if (p1 - p0 > lu) {                                                  // :101, note the strict >
#pragma unroll
            for (int d = 0; d < ND; ++d) x[d] = y[d];                        // :102
            p0 = p1;                                                         // :104
            na += n > 0 ? 1u : 0u;                                           // :105; counters restart at n == 0 (:122-125)
        }
        if (n > 0) {                                                         // :108 rem(n, nthin) == 0, :112 n > 0
            if (++cnt == a.nthin) {
                cnt = 0;
                if (slot < a.nsamples) {
                    if (a.chain != nullptr) {
#pragma unroll
                        for (int d = 0; d < ND; ++d)
                            if (d < ndim) a.chain[(slot * a.nchains + c) * ndim + d] = x[d];     // :113
                    }
                    if (a.chain_logp != nullptr) a.chain_logp[slot * a.nchains + c] = p0;        // :115
#pragma unroll
                    for (int d = 0; d < ND; ++d) { s1[d] += x[d]; s2[d] += x[d] * x[d]; }
                }
                ++slot;
            }
        }
    }
#pragma unroll
    for (int d = 0; d < ND; ++d) {
        if (d < ndim) {
            a.pos[c * ndim + d] = x[d];
            if (a.csum != nullptr) { a.csum[c * ndim + d] = s1[d]; a.csumsq[c * ndim + d] = s2[d]; }
        }
    }
    a.logp[c] = p0;
    a.naccept[c] = na;
}

// ------------------------------------------------------------------------------------------------
// FEW chains (the reference's own call is ONE): the draws come from a table.
//
// A chain is serial, and with the draws made in the loop above a step is ~1400 cycles of Philox, Box-Muller and a
// logarithm around ~100 cycles of proposal, density and accept test: 1.5e6 steps/s however few chains there are, with the
// rest of the chip idle.  The draws {normal per dimension, log u} are pure functions of (seed, iteration, chain), so a
// wide kernel (metro_draw_fill) computes them for a stretch of iterations first -- parallel over iterations, which the
// chain is not -- and the chains then only read them.
//
// Table of one launch: per wave of 64 chains (nl = the chains the wave really has), [step][k][nl] doubles with k = 0..ndim-1
// the normals and k = ndim the log-uniform; waves one after the other (a full wave takes 64 * T * (ndim + 1) doubles).
// Kernel: workgroups of TWO waves -- wave 0 runs the chains, wave 1 is their loader: it copies the next tile of the table
// (contiguous, so coalesced whatever nl is) into the other half of a double buffer in LDS while wave 0 works through the
// current one; one barrier per tile.  The loader waits for its loads where it issues them -- nothing has to be kept in
// flight across a loop by the compiler's leave -- and wave 0's barrier waits for LDS only.
// Same stream, same arithmetic: bit-identical to metropolis_chains_body.
// ------------------------------------------------------------------------------------------------
constexpr int kMetroTileBytes = 32 * 1024;      // one half of the LDS double buffer (64 KiB per workgroup)

struct MetroDrawArgs {
    double*  out;
    int64_t  nchains;
    int64_t  it0;
    int32_t  nsteps;           // table length T (a multiple of the tile's steps)
    int32_t  ndim;
    uint32_t seed_lo, seed_hi;
};

__host__ __device__ __forceinline__ int metro_tile_steps(int nl, int ndim)
{
    const int per_step = (ndim + 1) * nl * 8;
    const int ts = (kMetroTileBytes / per_step) & ~1;                        // even: tiles start on 16-byte boundaries whatever nl is
    return ts < 1 ? 1 : ts;                                                  // (one step per tile: nl = 64, an even number of doubles)
}

template <class Dens, int ND>
__device__ __forceinline__ void metropolis_chains_tabled_body(const MetropolisArgs& a, const double* draws, int nsteps)
{
    extern __shared__ __attribute__((aligned(16))) double mlds[];
    const int lane = threadIdx.x & 63;
    const bool loader = threadIdx.x >= 64;
    const int64_t c0 = (int64_t)blockIdx.x * 64;
    const int nl = (int)(a.nchains - c0 < 64 ? a.nchains - c0 : 64);
    const int ndim = a.ndim;
    __builtin_assume(ndim > ND / 2 && ndim <= ND);                           // how the host picks ND (metropolis_nd)
    const int ts = metro_tile_steps(nl, ndim);
    const int tile_doubles = ts * (ndim + 1) * nl;
    const int ntiles = (nsteps + ts - 1) / ts;                               // the table is padded to whole tiles
    const double* wtab = draws + c0 * (int64_t)nsteps * (ndim + 1);
    constexpr int kHalf = kMetroTileBytes / 8;
    if (loader) {
        for (int t = 0; t <= ntiles; ++t) {
            if (t < ntiles) {
                const double* src = wtab + (int64_t)t * tile_doubles;
                double* dst = mlds + (t & 1) * kHalf;
                const int64_t left = (int64_t)nsteps * (ndim + 1) * nl - (int64_t)t * tile_doubles;      // (the last tile may be short)
                const int ncopy = left < tile_doubles ? (int)left : tile_doubles;
                // the whole tile in flight at once (<= 32 x 16 B per lane), then into LDS: one memory round trip per tile
                constexpr int kMax = kMetroTileBytes / (64 * 16);
                const double2* src2 = reinterpret_cast<const double2*>(src);
                double2* dst2 = reinterpret_cast<double2*>(dst);
                const int n2 = ncopy >> 1;
                double2 r[kMax];
#pragma unroll
                for (int j = 0; j < kMax; ++j) { const int i = lane + 64 * j; r[j] = src2[i < n2 ? i : 0]; }   // (unconditional: a load in a branch is waited for at its end)
                __builtin_amdgcn_sched_barrier(0);                           // (all the loads first: the scheduler would keep six in flight)
#pragma unroll
                for (int j = 0; j < kMax; ++j) dst2[lane + 64 * j] = r[j];       // (past the tile's end: slots of this half nobody reads)
                if (ncopy & 1) { if (lane == 0) dst[ncopy - 1] = src[ncopy - 1]; }
            }
            __syncthreads();                                                 // tile t is in LDS; the chains are done with tile t - 1
        }
        return;
    }
    const int64_t c = c0 + lane;
    const bool live = lane < nl;
    const int64_t cc = live ? c : c0;
    double x[ND], y[ND], sc[ND], s1[ND], s2[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d) {
        x[d]  = d < ndim ? a.pos[cc * ndim + d] : 0.0;
        sc[d] = d < ndim ? a.step[d] : 0.0;
        s1[d] = (a.csum != nullptr && d < ndim) ? a.csum[cc * ndim + d] : 0.0;
        s2[d] = (a.csum != nullptr && d < ndim) ? a.csumsq[cc * ndim + d] : 0.0;
    }
    double   p0  = a.logp[cc];
    uint32_t na  = a.naccept[cc];
    int64_t  cnt = a.cnt0, slot = a.slot0;
    int64_t  n   = a.it0 + 1 - a.nburnin;                                    // :96
    const int64_t nit = a.it1 - a.it0;
    lds_barrier();                                                           // tile 0 is in LDS
    // Every lane runs the loop (a wave's spare lanes shadow its first chain and store nothing): the iteration counters then
    // stay scalar.  The draws of step s + 1 are read from LDS before step s's dependent chain starts.
    const int64_t sample_stride = a.nchains * ndim;
    const bool has_chain = a.chain != nullptr, has_logp = a.chain_logp != nullptr, has_mom = a.csum != nullptr;
    double* chain_p = a.chain + (slot * a.nchains + cc) * ndim;              // where the next stored sample goes (if has_chain)
    double* logp_p = a.chain_logp + slot * a.nchains + cc;
    const int estep = (ndim + 1) * nl;
    for (int t = 0; t < ntiles; ++t) {
        const double* tile = mlds + (t & 1) * kHalf + (live ? lane : 0);
        const int s_end = (int)(nit - (int64_t)t * ts < ts ? nit - (int64_t)t * ts : ts);
        double nrm[ND], lu_n;
#pragma unroll
        for (int d = 0; d < ND; ++d) nrm[d] = d < ndim ? tile[d * nl] : 0.0;
        lu_n = tile[ndim * nl];
        for (int s = 0; s < s_end; ++s, ++n) {
            double nr[ND];
#pragma unroll
            for (int d = 0; d < ND; ++d) nr[d] = nrm[d];
            const double lu = lu_n;
            {
                const double* e = tile + (s + 1 < s_end ? s + 1 : s) * estep;
#pragma unroll
                for (int d = 0; d < ND; ++d) nrm[d] = d < ndim ? e[d * nl] : 0.0;
                lu_n = e[ndim * nl];
            }
            typename Dens::Seq q;
            Dens::seq_init(q);
#pragma unroll
            for (int d = 0; d < ND; ++d) {
                if (d < ndim) {
                    y[d] = fma(sc[d], nr[d], x[d]);                          // :98  theta1 = sample_ppdf(theta0)
                    Dens::seq_add(q, y[d], d, a.dp);
                }
            }
            const double p1 = Dens::seq_finish(q, ndim, a.dp);               // :99
            const bool acc = p1 - p0 > lu;                                   // :101, note the strict >
#pragma unroll
            for (int d = 0; d < ND; ++d) x[d] = acc ? y[d] : x[d];           // :102
            p0 = acc ? p1 : p0;                                              // :104
            na += (acc && n > 0) ? 1u : 0u;                                  // :105
            if (n > 0 && ++cnt == a.nthin) {                                 // :108, :112 (uniform)
                cnt = 0;
                if (slot < a.nsamples) {
                    if (live) {
                        if (has_chain) {
#pragma unroll
                            for (int d = 0; d < ND; ++d) if (d < ndim) chain_p[d] = x[d];        // :113
                        }
                        if (has_logp) *logp_p = p0;                                              // :115
                    }
                    chain_p += sample_stride;
                    logp_p += a.nchains;
                    if (has_mom) {
#pragma unroll
                        for (int d = 0; d < ND; ++d) { s1[d] += x[d]; s2[d] += x[d] * x[d]; }
                    }
                }
                ++slot;
            }
        }
        lds_barrier();                                                       // done with tile t; tile t + 1 is in LDS
    }
    if (live) {
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            if (d < ndim) {
                a.pos[c * ndim + d] = x[d];
                if (a.csum != nullptr) { a.csum[c * ndim + d] = s1[d]; a.csumsq[c * ndim + d] = s2[d]; }
            }
        }
        a.logp[c] = p0;
        a.naccept[c] = na;
    }
}

struct MetropolisTabledArgs {      // (one struct: how the runtime-compiled kernels are launched)
    MetropolisArgs a;
    const double*  draws;
    int32_t        nsteps, pad_;
};

template <class Dens, int ND>
__global__ __launch_bounds__(128) void metropolis_chains_tabled(const MetropolisArgs a, const double* draws, int nsteps)
{
    metropolis_chains_tabled_body<Dens, ND>(a, draws, nsteps);
}

// Any ndim: the chain stays in memory, dimension-major ([ndim][nchains]) so that the 64 chains of a wave
// read and write consecutive doubles; pos / csum (chain-major, the C ABI's layout) are converted by
// metropolis_transpose before the first and after the last launch.
template <class Dens>
__device__ __forceinline__ void metropolis_chains_any_body(const MetropolisArgs& a)
{
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= a.nchains) return;
    const int ndim = a.ndim;
    const int64_t nc = a.nchains;
    double* x = a.xt + c;
    double* y = a.yt + c;
    double   p0  = a.logp[c];
    uint32_t na  = a.naccept[c];
    int64_t  cnt = a.cnt0, slot = a.slot0;
    int64_t  n   = a.it0 + 1 - a.nburnin;
    const uint32_t k0 = a.seed_lo ^ 0x4d455452u, k1 = a.seed_hi;
    for (int64_t it = a.it0; it < a.it1; ++it, ++n) {
        const U4 w = philox4x32_10((uint32_t)it, (uint32_t)((uint64_t)it >> 32), (uint32_t)c, 0u, k0, k1);
        typename Dens::Seq q;
        Dens::seq_init(q);
        double n0, n1, n2 = 0.0, n3 = 0.0;
        normal_pair(w.x, w.y, n0, n1);
        for (int d = 0; d < ndim; ++d) {
            double nd;
            if (d < 2) nd = d == 0 ? n0 : n1;
            else {
                const int r = (d - 2) & 3;
                if (r == 0) {
                    const U4 v = philox4x32_10((uint32_t)it, (uint32_t)((uint64_t)it >> 32), (uint32_t)c,
                                               (uint32_t)(1 + ((d - 2) >> 2)), k0, k1);
                    normal_pair(v.x, v.y, n0, n1);
                    normal_pair(v.z, v.w, n2, n3);
                }
                nd = r == 0 ? n0 : r == 1 ? n1 : r == 2 ? n2 : n3;
            }
            const double yd = fma(a.step[d], nd, x[(int64_t)d * nc]);        // :98
            y[(int64_t)d * nc] = yd;
            Dens::seq_add(q, yd, d, a.dp);
        }
        const double p1 = Dens::seq_finish(q, ndim, a.dp);                   // :99
        const uint64_t kk = ((uint64_t)w.z << 20) | (uint64_t)(w.w >> 12);
        const double lu = log_pos_normal(((double)kk + 0.5) * 0x1.0p-52);
        const bool acc = p1 - p0 > lu;                                       // :101
        if (acc) {
            p0 = p1;
            na += n > 0 ? 1u : 0u;
        }
        bool store = false;
        if (n > 0 && ++cnt == a.nthin) {
            cnt = 0;
            store = slot < a.nsamples;
            if (!store) ++slot;
        }
        if constexpr (BlobTrait<Dens>::n > 0) {
            constexpr int NB = BlobTrait<Dens>::n;
            double* cur = a.blob + c * NB;
            const bool keep = store && a.chain_blob != nullptr;
            if (acc || keep) {
#pragma unroll 1
                for (int i = 0; i < NB; ++i) {
                    const double bv = acc ? q.blob[i] : cur[i];
                    if (acc) cur[i] = bv;                                    // :103 blob0 = blob1
                    if (keep) a.chain_blob[(slot * nc + c) * NB + i] = bv;   // :117 reduce_blob!(blobs, blob0)
                }
            }
        }
        if (acc || store) {
            for (int d = 0; d < ndim; ++d) {
                const double v = acc ? y[(int64_t)d * nc] : x[(int64_t)d * nc];
                if (acc) x[(int64_t)d * nc] = v;                             // :102
                if (store) {
                    if (a.chain != nullptr) a.chain[(slot * nc + c) * ndim + d] = v;             // :113
                    if (a.st1 != nullptr) { a.st1[(int64_t)d * nc + c] += v; a.st2[(int64_t)d * nc + c] += v * v; }
                }
            }
            if (store) {
                if (a.chain_logp != nullptr) a.chain_logp[slot * nc + c] = p0;                   // :115
                ++slot;
            }
        }
    }
    a.logp[c] = p0;
    a.naccept[c] = na;
}

// chain-major [nchains][ndim] <-> dimension-major [ndim][nchains] (once per run, either side of the launches)
struct TransposeArgs {
    const double* src;
    double*       dst;
    int64_t       nchains;
    int32_t       ndim;
    int32_t       to_dim_major;
};

template <class Dens, int ND>
__global__ __launch_bounds__(256) void metropolis_chains(const MetropolisArgs a)
{
    if constexpr (ND > 0) metropolis_chains_body<Dens, ND>(a);
    else metropolis_chains_any_body<Dens>(a);
}

// Host route (kmc_metropolis_config.host_logpdf / host_propose: ANY closure for `pdf` and / or `sample_ppdf`, reference
// src/samplers.jl:59-61): one iteration of all chains per round trip.  PROPOSE draws the Gaussian step on the device (same
// stream as the in-kernel chains) unless the proposals come from the host; ACCEPT takes the proposals' log-pdfs (from the
// host callback or from the device density), runs the accept test of :101 with the stream's uniform, and stores.
struct MetroHostArgs {
    double*        pos;        // [nchains][ndim]
    double*        logp;       // [nchains] p0
    uint32_t*      naccept;
    double*        prop;       // [nchains][ndim] theta1
    const double*  p1;         // [nchains] pdf(theta1)
    double*        chain;      // [nsamples][nchains][ndim] or nullptr
    double*        chain_logp;
    double*        csum;       // [nchains][ndim] or nullptr
    double*        csumsq;
    const double*  step;       // [ndim] (PROPOSE)
    unsigned char* acc_out;    // [nchains] or nullptr
    int64_t        nchains;
    int64_t        it;         // 0-based iteration; reference n = it + 1 - nburnin (:96)
    int64_t        n;
    int64_t        slot;       // sample slot when store != 0
    int32_t        store;      // this iteration stores the current state (:108-116)
    int32_t        ndim;
    uint32_t       seed_lo, seed_hi;
};

#ifdef KMC_DEFINE_METROPOLIS_KERNELS   // non-template kernels: defined once, in kmc_metropolis_api.hip
// the draw table of metropolis_chains_tabled: one thread per (step, chain), the stream of metropolis_chains_body
__global__ __launch_bounds__(256) void metro_draw_fill(const MetroDrawArgs a)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)a.nsteps * a.nchains) return;
    const int64_t s = idx / a.nchains;
    const int64_t c = idx - s * a.nchains;
    const int64_t it = a.it0 + s;
    const int ndim = a.ndim;
    const int64_t c0 = c & ~(int64_t)63;
    const int nl = (int)(a.nchains - c0 < 64 ? a.nchains - c0 : 64);
    double* e = a.out + c0 * (int64_t)a.nsteps * (ndim + 1) + s * (int64_t)(ndim + 1) * nl + (c - c0);
    const uint32_t k0 = a.seed_lo ^ 0x4d455452u, k1 = a.seed_hi;
    const U4 w = philox4x32_10((uint32_t)it, (uint32_t)((uint64_t)it >> 32), (uint32_t)c, 0u, k0, k1);
    if (ndim == 1) {
        e[0] = normal_first(w.x, w.y);
    } else {
        double n0, n1;
        normal_pair(w.x, w.y, n0, n1);
        e[0] = n0;
        e[nl] = n1;
        for (int b = 1; 4 * b - 2 < ndim; ++b) {
            const U4 v = philox4x32_10((uint32_t)it, (uint32_t)((uint64_t)it >> 32), (uint32_t)c, (uint32_t)b, k0, k1);
            double m0, m1, m2, m3;
            normal_pair(v.x, v.y, m0, m1);
            normal_pair(v.z, v.w, m2, m3);
            const int d0 = 4 * b - 2;
            e[(int64_t)d0 * nl] = m0;
            if (d0 + 1 < ndim) e[(int64_t)(d0 + 1) * nl] = m1;
            if (d0 + 2 < ndim) e[(int64_t)(d0 + 2) * nl] = m2;
            if (d0 + 3 < ndim) e[(int64_t)(d0 + 3) * nl] = m3;
        }
    }
    const uint64_t kk = ((uint64_t)w.z << 20) | (uint64_t)(w.w >> 12);
    e[(int64_t)ndim * nl] = log_pos_normal(((double)kk + 0.5) * 0x1.0p-52);
}
__global__ __launch_bounds__(256) void metro_host_propose(const MetroHostArgs a)
{
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= a.nchains) return;
    const uint32_t k0 = a.seed_lo ^ 0x4d455452u, k1 = a.seed_hi;
    const uint32_t it_lo = (uint32_t)a.it, it_hi = (uint32_t)((uint64_t)a.it >> 32);
    double n0 = 0.0, n1 = 0.0, n2 = 0.0, n3 = 0.0;
    for (int d = 0; d < a.ndim; ++d) {
        double nd;
        if (d < 2) {
            if (d == 0) { const U4 w = philox4x32_10(it_lo, it_hi, (uint32_t)c, 0u, k0, k1); normal_pair(w.x, w.y, n0, n1); }
            nd = d == 0 ? n0 : n1;
        } else {
            const int r = (d - 2) & 3;
            if (r == 0) {
                const U4 v = philox4x32_10(it_lo, it_hi, (uint32_t)c, (uint32_t)(1 + ((d - 2) >> 2)), k0, k1);
                normal_pair(v.x, v.y, n0, n1);
                normal_pair(v.z, v.w, n2, n3);
            }
            nd = r == 0 ? n0 : r == 1 ? n1 : r == 2 ? n2 : n3;
        }
        a.prop[c * a.ndim + d] = fma(a.step[d], nd, a.pos[c * a.ndim + d]);          // :98 theta1 = sample_ppdf(theta0)
    }
}

__global__ __launch_bounds__(256) void metro_host_accept(const MetroHostArgs a)
{
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= a.nchains) return;
    const U4 w = philox4x32_10((uint32_t)a.it, (uint32_t)((uint64_t)a.it >> 32), (uint32_t)c, 0u, a.seed_lo ^ 0x4d455452u, a.seed_hi);
    const uint64_t kk = ((uint64_t)w.z << 20) | (uint64_t)(w.w >> 12);
    const double lu = log_pos_normal(((double)kk + 0.5) * 0x1.0p-52);
    const double p1 = a.p1[c];
    double p0 = a.logp[c];
    const bool acc = p1 - p0 > lu;                                                   // :101, strict >
    if (acc) {
        p0 = p1;                                                                     // :104
        a.logp[c] = p1;
        if (a.n > 0) a.naccept[c] += 1u;                                             // :105, counters restart at n == 0 (:122-125)
    }
    if (a.acc_out != nullptr) a.acc_out[c] = acc ? 1 : 0;
    if (acc || a.store) {
        for (int d = 0; d < a.ndim; ++d) {
            const double v = acc ? a.prop[c * a.ndim + d] : a.pos[c * a.ndim + d];
            if (acc) a.pos[c * a.ndim + d] = v;                                      // :102
            if (a.store) {
                if (a.chain != nullptr) a.chain[(a.slot * a.nchains + c) * a.ndim + d] = v;      // :113
                if (a.csum != nullptr) { a.csum[c * a.ndim + d] += v; a.csumsq[c * a.ndim + d] += v * v; }
            }
        }
        if (a.store && a.chain_logp != nullptr) a.chain_logp[a.slot * a.nchains + c] = p0;       // :115
    }
}

__global__ __launch_bounds__(256) void metropolis_transpose(const TransposeArgs a)
{
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= a.nchains) return;
    for (int d = 0; d < a.ndim; ++d) {
        if (a.to_dim_major) a.dst[(int64_t)d * a.nchains + c] = a.src[c * a.ndim + d];
        else a.dst[c * a.ndim + d] = a.src[(int64_t)d * a.nchains + c];
    }
}
#endif

}  // namespace kmc
