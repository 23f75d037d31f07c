// kmc_islands.hpp -- ISLAND MODE (opt-in extension; the exact mode in kmc_kernels.hpp is the default).
//
// The north-star's "LDS-staged complementary-half walker block", done the only way that keeps a
// correct sampler: the ensemble is cut into islands of S walkers (64, 128 or 256), one workgroup each.
// An island's whole state (positions + log-pdfs, ~70 KB at ndim 32) is loaded into LDS once per
// epoch and the workgroup then runs `ngen` generations of the reference's half-split stretch move
// (src/samplers.jl:245-274) entirely out of LDS -- partners are drawn from the ISLAND's
// complementary half, `__syncthreads()` is the join of src/samplers.jl:273 -- before writing the
// island back.  Between epochs the walkers are re-dealt to islands by a state-independent
// permutation (slot s holds walker (A*s + C) mod N), so every walker keeps meeting new partners.
// Each island update is a valid emcee kernel for its 256 walkers and the deal ignores the state,
// hence the target distribution is untouched; what differs from the reference is the partner
// pool (island half instead of ensemble half, src/samplers.jl:250).  No HBM traffic and no kernel
// boundary inside an epoch.
//
// Thread mapping: 256 threads = 128 active walkers x 2 lanes; a walker's row is striped over its
// lane pair exactly like half_step_vec<L = 2> (lane j, chunk k holds elements 2(2k+j), 2(2k+j)+1), so
// the density code (frag_partial<2, K>) and the DPP pair reduction are shared with the exact mode.
// Random stream: the exact mode's Philox block keyed by step = 2*generation + half and by the SLOT
// index (island * 256 + local index).
#pragma once
#include "kmc_kernels.hpp"

namespace kmc {

constexpr int kIslandSizeDefault = 256;   // walkers per island (= threads per workgroup): 64, 128 or 256

// The sample test of src/samplers.jl:268 (n > 0 && n % nthin == 0 -> slot n / nthin - 1), carried from one generation to
// the next: the 64-bit division it takes literally is ~180 scalar instructions, a quarter of a generation of the resident
// kernels at the reference's own sizes.  One division per launch instead.
struct ThinClock {
    int64_t n, phase, q;
    int64_t ring, rs, rnext;                              // KMC_STREAM_CHAIN: the chain is a ring of `ring` slots; rs = (q - 1) mod ring at a hit
    __device__ __forceinline__ ThinClock(int64_t gen0, int64_t nburnin, int64_t nthin, int64_t ring_slots = 0)
    {
        n = gen0 - nburnin;                               // the loop variable (:245) of the generation before the launch's first
        if (n > 0) { q = n / nthin; phase = n - q * nthin; } else { q = 0; phase = 0; }
        ring = ring_slots; rs = 0;
        rnext = ring_slots > 0 ? q % ring_slots : 0;
    }
    // on to the next generation: true when it is a thinning hit (its slot is q - 1; the caller compares with nsamples)
    __device__ __forceinline__ bool tick(int64_t nthin)
    {
        ++n;
        if (n <= 0) return false;
        if (++phase == nthin) {
            phase = 0; ++q;
            rs = rnext;
            if (++rnext == ring) rnext = 0;               // (ring == 0: never equal, rnext just counts)
            return true;
        }
        return false;
    }
    // where the hit's sample goes in the chain buffer
    __device__ __forceinline__ int64_t wslot() const { return ring > 0 ? rs : q - 1; }
};

struct IslandArgs {
    double*       pos;        // [nwalkers][ld], walker-id order
    double*       logp;       // [nwalkers]
    uint32_t*     naccept;    // [nwalkers]
    int64_t       nwalkers;
    int64_t       permA, permC;     // slot -> walker id of this epoch
    int64_t       gen0;             // first generation of this launch
    int32_t       ngen;             // generations in this launch (<= epoch length)
    int32_t       ndim;
    int32_t       ld;               // global row stride (doubles)
    int32_t       pad_;
    int64_t       nburnin, nthin, nsamples;
    DrawConsts    dc;               // dc.nhalf = S / 2
    DensityParams dp;
    double*       msum;             // [nislands][4K] per-island partial sums or nullptr
    double*       msumsq;
};

__device__ __forceinline__ int64_t island_walker(const IslandArgs& a, int64_t slot)
{
    // (A * slot + C) mod N without overflow for N < 2^31 (A, C < N)
    return (int64_t)(((unsigned long long)a.permA * (unsigned long long)slot + (unsigned long long)a.permC) %
                     (unsigned long long)a.nwalkers);
}

template <class Dens, int S, int K, bool RAGGED>
__device__ __forceinline__ void island_epoch_body(const IslandArgs& a)
{
    static_assert(S == 64 || S == 128 || S == 256, "island = one workgroup of S threads");
    constexpr int HS = S / 2, L = 2, NW = S / 64;   // NW waves
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int ndim = RAGGED ? a.ndim : 4 * K;
    const int ld   = RAGGED ? a.ld : 4 * K;
    // LDS row: the two lanes' chunk sets kept apart, K+1 slots each -- chunk (k, j) sits at slot
    // j*(K+1) + k.  Rows then start 8(K+1) banks apart and the two lanes of a pair 4(K+1) banks apart,
    // which makes the 16-lane ds_read_b128 groups of the own-row reads conflict-free for K >= 2.
    constexpr int lld = 4 * (K + 1);                      // doubles per LDS row
    double* lpos  = lds;                                  // [S][lld]
    double* llogp = lds + S * lld;                        // [S]

    const int t = threadIdx.x;
    const int i = t >> 1, j = t & 1;                      // active index within a half, lane of the pair
    const int64_t slot0 = (int64_t)blockIdx.x * S;

    // ---- deal: slot (island, t) <- walker (A*slot + C) mod N ------------------------------------
    const int64_t wid = island_walker(a, slot0 + t);
    {
        const double2* src = reinterpret_cast<const double2*>(a.pos + wid * (int64_t)a.ld);
        double2* dst = reinterpret_cast<double2*>(lpos + t * lld);
        for (int c = 0; c < ld / 2; ++c) dst[(c & 1) * (K + 1) + (c >> 1)] = src[c];
        llogp[t] = a.logp[wid];
    }
    __syncthreads();

    bool cv[K];
#pragma unroll
    for (int k = 0; k < K; ++k) cv[k] = !RAGGED || 2 * (k * L + j) < ld;
    const double2 zero2 = make_double2(0.0, 0.0);
    double2 ms[K], mq[K];
#pragma unroll
    for (int k = 0; k < K; ++k) { ms[k] = zero2; mq[k] = zero2; }
    // Scalar layout (one walker per lane, once per GENERATION): lanes 0-31 of a wave own the wave's 32
    // first-half walkers, lanes 32-63 its 32 second-half walkers -- Philox, z, log z, log u, p0, the
    // accept test and the acceptance counter live here, so the two logarithms are computed once per
    // walker-step instead of once per lane of the pair.  Row layout: lane pair (i, j) as above.
    const int lane = t & 63, wave = t >> 6;
    const int hA   = lane >> 5;                                       // which half this scalar lane serves
    const int ownA = hA * HS + (wave << 5) + (lane & 31);             // its walker's local index
    uint32_t naccA = 0u;

    ThinClock clk(a.gen0, a.nburnin, a.nthin);
    for (int gg = 0; gg < a.ngen; ++gg) {
        const int64_t gen = a.gen0 + gg;
        const bool hit = clk.tick(a.nthin);               // clk.n: the reference's loop variable (:245)
        const bool count = clk.n > 0;
        const bool sample = hit && (clk.q - 1) < a.nsamples;                      // :268
        const Draw drA = draw_step(a.dc, 2ull * (uint64_t)gen + (uint64_t)hA, (uint64_t)(slot0 + ownA));   // :250, :252
        const double p0A = llogp[ownA];                   // only this walker's own update changes it
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            const int own_l = half * HS + i;                                      // :247
            const int src = ((half << 5) + (lane >> 1)) * 4;                      // scalar lane of this pair's walker
            const int partner = __builtin_amdgcn_ds_bpermute(src, (int)drA.partner);
            const double z = bperm_f64(src, drA.z);
            const int oth_l = (1 - half) * HS + partner;
            const double2* own = reinterpret_cast<const double2*>(lpos + own_l * lld);
            const double2* oth = reinterpret_cast<const double2*>(lpos + oth_l * lld);
            double2 xc[K], y[K];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                xc[k] = cv[k] ? own[j * (K + 1) + k] : zero2;
                const double2 xo = cv[k] ? oth[j * (K + 1) + k] : zero2;
                y[k].x = fma(z, xc[k].x - xo.x, xo.x);                            // :255
                y[k].y = fma(z, xc[k].y - xo.y, xo.y);
            }
            const double Ssum = group_sum<L>(Dens::template frag_partial<L, K>(y, j, ndim, a.dp));
            const double p1 = Dens::finish(Ssum, a.dp);                           // :257
            // row -> scalar: the pair of walker (lane & 31) sits in lanes 2*(lane & 31), +1
            const double p1A = bperm_f64(((lane & 31) * 2) * 4, p1);
            const bool accA = (hA == half) && accept_test(drA, p1A, p0A);         // :260
            const unsigned long long accmask = __ballot(accA);
            if (accA) {                                                           // :262, :265
                llogp[ownA] = p1A;
                if (count) naccA += 1u;
            }
            const bool acc = ((accmask >> ((half << 5) + (lane >> 1))) & 1ull) != 0;
            if (acc) {                                                            // :261
                double2* ownw = reinterpret_cast<double2*>(lpos + own_l * lld);
#pragma unroll
                for (int k = 0; k < K; ++k) if (cv[k]) ownw[j * (K + 1) + k] = y[k];
            }
            if (sample) {                                 // the walker's state after its update (:268-269)
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const double2 cur = sel2(acc, y[k], xc[k]);
                    ms[k].x += cur.x; ms[k].y += cur.y;
                    mq[k].x += cur.x * cur.x; mq[k].y += cur.y * cur.y;
                }
            }
            __syncthreads();                              // the join of :273
        }
    }

    // ---- hand the island back -------------------------------------------------------------------
    {
        double2* dst = reinterpret_cast<double2*>(a.pos + wid * (int64_t)a.ld);
        const double2* src = reinterpret_cast<const double2*>(lpos + t * lld);
        for (int c = 0; c < ld / 2; ++c) dst[c] = src[(c & 1) * (K + 1) + (c >> 1)];
        a.logp[wid] = llogp[t];
    }
    if (naccA) a.naccept[island_walker(a, slot0 + ownA)] += naccA;
    if (a.msum != nullptr) {
        // lanes with equal j hold the same dimensions: fold the wave's 32 pairs, then the 4 waves
        __syncthreads();                                  // LDS is free again
        double* red = lds;                                // [NW waves][K][2 lanes][4]
#pragma unroll
        for (int k = 0; k < K; ++k) {
            ms[k].x = wave_fold<L>(ms[k].x); ms[k].y = wave_fold<L>(ms[k].y);
            mq[k].x = wave_fold<L>(mq[k].x); mq[k].y = wave_fold<L>(mq[k].y);
            if (lane < 2) {
                double* r = red + ((wave * K + k) * 2 + lane) * 4;
                r[0] = ms[k].x; r[1] = ms[k].y; r[2] = mq[k].x; r[3] = mq[k].y;
            }
        }
        __syncthreads();
        if (t < 2 * K) {                                  // thread (k, j)
            double s0 = 0.0, s1 = 0.0, q0 = 0.0, q1 = 0.0;
            for (int w = 0; w < NW; ++w) {
                const double* r = red + ((w * K + (t >> 1)) * 2 + (t & 1)) * 4;
                s0 += r[0]; s1 += r[1]; q0 += r[2]; q1 += r[3];
            }
            double* gs = a.msum + ((int64_t)blockIdx.x * 2 * K + t) * 2;     // dims 2*(2k+j), +1
            double* gq = a.msumsq + ((int64_t)blockIdx.x * 2 * K + t) * 2;
            gs[0] += s0; gs[1] += s1;
            gq[0] += q0; gq[1] += q1;
        }
    }
}

template <class Dens, int S, int K, bool RAGGED>
__global__ __launch_bounds__(S) void island_epoch(const IslandArgs a)
{
    island_epoch_body<Dens, S, K, RAGGED>(a);
}


// ------------------------------------------------------------------------------------------------
// RESIDENT mode: the EXACT sampler for small ensembles (nwalkers <= 1024 and the ensemble fits 160 KiB of LDS).
//
// When the whole ensemble fits one workgroup's LDS, "the island's complementary half" IS the
// ensemble's complementary half: a single island with the identity deal and the RNG keyed by the
// walker index is the reference's algorithm unchanged -- bit-identical to half_step_vec/generic --
// but a launch now carries many generations (the reference's own sizes: 100 walkers, 10^5..10^7
// evaluations) instead of half of one, with `__syncthreads()` as the join of src/samplers.jl:273.
// Unlike the island kernel this one takes any even S <= TPB (256, 512 or 1024 threads, as LDS allows)
// and also stores the chain (:269-271).
// ------------------------------------------------------------------------------------------------
struct ResidentArgs {
    IslandArgs    is;           // pos/logp/naccept, generations, schedule, draws, density, per-block moments
    int32_t       S;            // nwalkers (even, <= threads per workgroup)
    int32_t       pad_;
    double*       chain;        // [nsamples][S][ld] or nullptr
    double*       chain_logp;   // [nsamples][S] or nullptr
    double*       blob;         // body densities with blobs (resident_lane_body): [S][NB] current blobs, or nullptr
    double*       chain_blob;   //   [nsamples][S][NB] or nullptr
    int64_t       ring_slots;   // KMC_STREAM_CHAIN: chain / chain_logp are rings of this many sample slots (0: one slot per sample)
    const double2* draws;       // the launch's draws, [ngen padded to kDrawBatch][S] x 32 B {z, (N-1) log z | log u, partner} from
                                //   draw_table_fill
};

// ------------------------------------------------------------------------------------------------
// Draw table of a resident launch.  The draws {partner, z, (N-1) log z, log u} of a walker-step are pure functions of
// (seed, step, walker) -- no state -- and in the one-walker-per-thread resident kernel they are most of the work: Philox
// and two logarithms are ~350 instructions a generation, the move itself a few dozen; timing-only builds without them run
// 0.26 instead of 0.41 us per half-step at 100 x 2 and 0.54 instead of 1.04 at 1000 x 4.  The resident kernel has one CU;
// the other 255 are idle.  So a wide kernel computes the draws of the launch's generations first (same function, same
// bits), and the resident kernel streams them, one generation ahead of their use.
// ------------------------------------------------------------------------------------------------
constexpr int kDrawBatch = 4;  // generations per load batch of the resident kernel (the table is padded to whole batches)
struct DrawTableArgs {
    DrawConsts dc;
    int64_t    gen0;
    int32_t    ngen, S;
    double2*   out;             // [ngen][S][2]
};
__device__ __forceinline__ Draw draw_table_load(const double2* tab, int64_t idx)
{
    const double2 e0 = tab[2 * idx], e1 = tab[2 * idx + 1];
    Draw d;
    d.z = e0.x; d.t1 = e0.y; d.lu = e1.x;
    // (both words carry the index: a destination register nobody reads would be handed out again at once, and whoever gets
    //  it waits for the load)
    d.partner = (uint32_t)__double2loint(e1.y) | (uint32_t)__double2hiint(e1.y);
    return d;
}
#ifdef KMC_DEFINE_LAUNCH_KERNELS   // kmc_launch.hip
__global__ __launch_bounds__(256) void draw_table_fill(const DrawTableArgs a)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)a.ngen * a.S) return;
    const int64_t g = idx / a.S;
    const int t = (int)(idx - g * a.S);
    const int half = t >= a.S / 2 ? 1 : 0;
    const Draw d = draw_step(a.dc, 2ull * (uint64_t)(a.gen0 + g) + (uint64_t)half, (uint64_t)t);   // as resident_lane_body draws
    a.out[2 * idx]     = make_double2(d.z, d.t1);
    a.out[2 * idx + 1] = make_double2(d.lu, __hiloint2double((int)d.partner, (int)d.partner));
}
#endif

template <class Dens, int K, bool RAGGED, int TPB = 256>
__device__ __forceinline__ void resident_body(const ResidentArgs& ra)
{
    static_assert(TPB == 256 || TPB == 512 || TPB == 1024, "one workgroup: TPB threads = TPB/2 lane pairs");
    const IslandArgs& a = ra.is;
    constexpr int L = 2;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int S = ra.S, HS = S / 2;
    const int ndim = RAGGED ? a.ndim : 4 * K;
    const int ld   = RAGGED ? a.ld : 4 * K;
    constexpr int lld = 4 * (K + 1);                      // same LDS row layout as the island kernel
    double* lpos  = lds;                                  // [S][lld]
    double* llogp = lds + S * lld;                        // [S]

    const int t = threadIdx.x;
    const int i = t >> 1, j = t & 1;
    const bool rowv = i < HS;                             // this lane pair has a walker in each half
    if (t < S) {
        const double2* src = reinterpret_cast<const double2*>(a.pos + (int64_t)t * a.ld);
        double2* dst = reinterpret_cast<double2*>(lpos + t * lld);
        for (int c = 0; c < ld / 2; ++c) dst[(c & 1) * (K + 1) + (c >> 1)] = src[c];
        llogp[t] = a.logp[t];
    }
    __syncthreads();

    bool cv[K];
#pragma unroll
    for (int k = 0; k < K; ++k) cv[k] = !RAGGED || 2 * (k * L + j) < ld;
    const double2 zero2 = make_double2(0.0, 0.0);
    double2 ms[K], mq[K];
#pragma unroll
    for (int k = 0; k < K; ++k) { ms[k] = zero2; mq[k] = zero2; }
    const int lane = t & 63, wave = t >> 6;
    const int hA   = lane >> 5;
    const int iA   = (wave << 5) + (lane & 31);           // index within the half served by this scalar lane
    const bool scv = iA < HS;
    const int ownA = hA * HS + (scv ? iA : 0);
    uint32_t naccA = 0u;
    // the launch's draws come from draw_table_fill (see there), one generation ahead of their use: a generation of these
    // longer rows outlasts the load
    Draw nxt{};
    if (a.ngen > 0) nxt = draw_table_load(ra.draws, ownA);

    ThinClock clk(a.gen0, a.nburnin, a.nthin, ra.ring_slots);
    for (int gg = 0; gg < a.ngen; ++gg) {
        const bool hit = clk.tick(a.nthin);               // clk.n: the reference's loop variable (:245)
        const bool count = clk.n > 0;
        const int64_t slot = clk.q - 1;
        const bool sample = hit && slot < a.nsamples;                                            // :268
        const int64_t wslot = clk.wslot();                // (the slot itself, or its place in the streamed chain's ring)
        const Draw drA = nxt;                             // row (step, walker ownA) of the table: keyed by the walker index
        nxt = draw_table_load(ra.draws, (int64_t)(gg + 1 < a.ngen ? gg + 1 : gg) * S + ownA);
        const double p0A = llogp[ownA];
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            const int own_l = half * HS + (rowv ? i : 0);                         // :247
            const int src = ((half << 5) + (lane >> 1)) * 4;
            const int partner = __builtin_amdgcn_ds_bpermute(src, (int)drA.partner);
            const double z = bperm_f64(src, drA.z);
            const int oth_l = (1 - half) * HS + (rowv ? partner : 0);
            const double2* own = reinterpret_cast<const double2*>(lpos + own_l * lld);
            const double2* oth = reinterpret_cast<const double2*>(lpos + oth_l * lld);
            double2 xc[K], y[K];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                xc[k] = cv[k] ? own[j * (K + 1) + k] : zero2;
                const double2 xo = cv[k] ? oth[j * (K + 1) + k] : zero2;
                y[k].x = fma(z, xc[k].x - xo.x, xo.x);                            // :255
                y[k].y = fma(z, xc[k].y - xo.y, xo.y);
            }
            const double Ssum = group_sum<L>(Dens::template frag_partial<L, K>(y, j, ndim, a.dp));
            const double p1 = Dens::finish(Ssum, a.dp);                           // :257
            const double p1A = bperm_f64(((lane & 31) * 2) * 4, p1);
            const bool accA = scv && (hA == half) && accept_test(drA, p1A, p0A);  // :260
            const unsigned long long accmask = __ballot(accA);
            if (accA) {                                                           // :262, :265
                llogp[ownA] = p1A;
                if (count) naccA += 1u;
            }
            const bool acc = ((accmask >> ((half << 5) + (lane >> 1))) & 1ull) != 0;
            if (acc) {                                                            // :261
                double2* ownw = reinterpret_cast<double2*>(lpos + own_l * lld);
#pragma unroll
                for (int k = 0; k < K; ++k) if (cv[k]) ownw[j * (K + 1) + k] = y[k];
            }
            if (sample && rowv) {
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const double2 cur = sel2(acc, y[k], xc[k]);
                    ms[k].x += cur.x; ms[k].y += cur.y;
                    mq[k].x += cur.x * cur.x; mq[k].y += cur.y * cur.y;
                }
            }
            lds_barrier();                                // the join of :273 (LDS only: the prefetched draws stay in flight)
        }
        if (sample && (ra.chain != nullptr || ra.chain_logp != nullptr)) {       // :268-271
            if (t < S) {
                if (ra.chain != nullptr) {
                    double2* dst = reinterpret_cast<double2*>(ra.chain + (wslot * S + t) * (int64_t)a.ld);
                    const double2* src2 = reinterpret_cast<const double2*>(lpos + t * lld);
                    for (int c = 0; c < ld / 2; ++c) dst[c] = src2[(c & 1) * (K + 1) + (c >> 1)];
                }
                if (ra.chain_logp != nullptr) ra.chain_logp[wslot * S + t] = llogp[t];
            }
            // rows are only read here; the next generation's first writes come after its own barrier-free
            // reads, by other threads -> order them
            lds_barrier();
        }
    }

    if (t < S) {
        double2* dst = reinterpret_cast<double2*>(a.pos + (int64_t)t * a.ld);
        const double2* src = reinterpret_cast<const double2*>(lpos + t * lld);
        for (int c = 0; c < ld / 2; ++c) dst[c] = src[(c & 1) * (K + 1) + (c >> 1)];
        a.logp[t] = llogp[t];
    }
    if (naccA) a.naccept[ownA] += naccA;
    if (a.msum != nullptr) {
        __syncthreads();
        double* red = lds;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            ms[k].x = wave_fold<L>(ms[k].x); ms[k].y = wave_fold<L>(ms[k].y);
            mq[k].x = wave_fold<L>(mq[k].x); mq[k].y = wave_fold<L>(mq[k].y);
            if (lane < 2) {
                double* r = red + ((wave * K + k) * 2 + lane) * 4;
                r[0] = ms[k].x; r[1] = ms[k].y; r[2] = mq[k].x; r[3] = mq[k].y;
            }
        }
        __syncthreads();
        if (t < 2 * K) {
            double s0 = 0.0, s1 = 0.0, q0 = 0.0, q1 = 0.0;
            for (int w = 0; w < TPB / 64; ++w) {
                const double* r = red + ((w * K + (t >> 1)) * 2 + (t & 1)) * 4;
                s0 += r[0]; s1 += r[1]; q0 += r[2]; q1 += r[3];
            }
            a.msum[2 * t] += s0; a.msum[2 * t + 1] += s1;
            a.msumsq[2 * t] += q0; a.msumsq[2 * t + 1] += q1;
        }
    }
}

template <class Dens, int K, bool RAGGED, int TPB>
__global__ __launch_bounds__(TPB) void resident_epoch(const ResidentArgs a)
{
    resident_body<Dens, K, RAGGED, TPB>(a);
}

// ------------------------------------------------------------------------------------------------
// RESIDENT mode, ONE WALKER PER THREAD (short rows: ndim <= 8 for the menu and term / pair densities, ndim <= 32 for body
// densities, which hold the whole proposal per lane anyway): the same exact sampler out of one workgroup's LDS -- thread t
// owns walker t (first half: t < S/2).  A generation: every thread draws its walker's step (both half-steps' draws at once:
// the stream is keyed by (step, walker)), then the first half's threads move against the second half's rows, barrier (the
// join of src/samplers.jl:273), and the other way round.  A thread's own row is written only by itself and, during its
// half-step, read only by itself (partners come from the other half), so an accepted move goes straight into LDS.  No
// cross-lane traffic at all (the two-lanes-per-walker kernel above pays three ds_bpermute round trips and a DPP reduction per
// half-step): measured 0.39 against 0.61 us per half-step at the README shape (100 walkers, 1-D), 0.42 / 0.59 at 2-D,
// 0.53 / 0.63 at 8-D, 1.39 / 1.12 at 32-D -- hence ndim <= 8 for the menu densities -- and with the draws taken from the
// table above instead of made here 0.22, 0.25, 0.42 and (body density, 32-D) 0.90.  Log-densities are summed in index order (the Seq interface),
// like half_step_generic and the oracle.  EXACT_ND: ndim == ND at compile time (runtime-compiled body densities).
// ------------------------------------------------------------------------------------------------
// T: the storage type of rows and chain (KMC_F32: float rows, double arithmetic, proposals rounded before their density).
template <class Dens, int ND, bool EXACT_ND, class T = double>
__device__ __forceinline__ void resident_lane_body(const ResidentArgs& ra)
{
    const IslandArgs& a = ra.is;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int S = ra.S, HS = S / 2;
    const int ndim = EXACT_ND ? ND : a.ndim;              // <= ND
    constexpr int LS = ND | 1;                            // LDS row stride in doubles (odd: rows start on different banks)
    double* lpos  = lds;                                  // [S][LS]
    double* llogp = lds + (size_t)S * LS;                 // [S]
    const int t = threadIdx.x;
    const bool live = t < S;
    if (live) {
#pragma unroll
        for (int d = 0; d < ND; ++d) lpos[t * LS + d] = d < ndim ? (double)reinterpret_cast<const T*>(a.pos)[(int64_t)t * a.ld + d] : 0.0;
        llogp[t] = a.logp[t];
    }
    __syncthreads();
    const int myhalf = t >= HS ? 1 : 0;
    uint32_t nacc = 0u;
    double s1[ND], s2[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d) { s1[d] = 0.0; s2[d] = 0.0; }
    // The draws come from draw_table_fill (ra.draws: the host driver's job), kDrawBatch generations per load and one batch
    // ahead of their use: the table was written by other CUs a moment ago and a load takes about a microsecond to come
    // back -- more than a generation takes here.  (One generation ahead: 0.40 us per half-step at 100 x 2 against 0.30
    // with loads that always hit.)  The batch's generations are unrolled, so each reads its draws from registers of its
    // own: no rotation of registers -- a copy of a register a load is still in flight to would wait for it.
    // (long rows: a generation outlasts a load by itself, and four copies of a 32-dimensional body spill)
    // (float rows: one generation ahead -- the option exists for large ensembles; here it only has to not fall off a cliff,
    //  and four copies of the generation for every row length and density are 10 s of build time)
    constexpr int B = sizeof(T) == 4 ? 1 : (ND <= 4 ? kDrawBatch : (ND <= 16 ? 2 : 1));    // (a batch in flight for >= 1.5 us)
    static_assert(kDrawBatch % B == 0, "the host pads the table to kDrawBatch generations");
    const int tl = live ? t : 0;
    const int nb = (a.ngen + B - 1) / B;                  // the table is padded to whole batches
    Draw cur[B], nxt[B];
#pragma unroll
    for (int u = 0; u < B; ++u) { cur[u] = Draw{}; nxt[u] = Draw{}; }
    if (nb > 0) {
#pragma unroll
        for (int u = 0; u < B; ++u) nxt[u] = draw_table_load(ra.draws, (int64_t)u * S + tl);
    }

    ThinClock clk(a.gen0, a.nburnin, a.nthin, ra.ring_slots);
#pragma unroll 1
    for (int b = 0; b < nb; ++b) {
#pragma unroll
      for (int u = 0; u < B; ++u) cur[u] = nxt[u];        // (waits for the batch loaded during the previous one)
      {
        const int bn = b + 1 < nb ? b + 1 : b;            // (the last one reloads itself)
#pragma unroll
        for (int u = 0; u < B; ++u) nxt[u] = draw_table_load(ra.draws, ((int64_t)bn * B + u) * S + tl);
      }
      // (unrolled: with a loop here the compiler drains every outstanding load before entering it)
#pragma unroll
      for (int sub = 0; sub < B; ++sub) {
        if (b * B + sub >= a.ngen) break;
        const bool hit = clk.tick(a.nthin);               // clk.n: the reference's loop variable (:245)
        const bool count = clk.n > 0;
        const int64_t slot = clk.q - 1;
        const bool sample = hit && slot < a.nsamples;                                            // :268
        const int64_t wslot = clk.wslot();                // (the slot itself, or its place in the streamed chain's ring)
        const Draw dr = cur[sub];                                                                // :250, :252
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            if (live && myhalf == half) {                                         // :247
                const double* oth = lpos + ((1 - half) * HS + (int)dr.partner) * LS;
                const double* own = lpos + t * LS;
                typename Dens::Seq q;
                Dens::seq_init(q);
                double y[ND];
#pragma unroll
                for (int d = 0; d < ND; ++d) {
                    if (d < ndim) {
                        y[d] = as_stored<T>(fma(dr.z, own[d] - oth[d], oth[d])); // :255
                        Dens::seq_add(q, y[d], d, a.dp);
                    }
                }
                const double p1 = Dens::seq_finish(q, ndim, a.dp);                // :257
                if (accept_test(dr, p1, llogp[t])) {                              // :260
#pragma unroll
                    for (int d = 0; d < ND; ++d) if (d < ndim) lpos[t * LS + d] = y[d];   // :261
                    llogp[t] = p1;                                                // :262
                    if (count) nacc += 1u;                                        // :265
                    if constexpr (BlobTrait<Dens>::n > 0) {
#pragma unroll 1
                        for (int i = 0; i < BlobTrait<Dens>::n; ++i) ra.blob[(int64_t)t * BlobTrait<Dens>::n + i] = q.blob[i];   // :264
                    }
                }
            }
            lds_barrier();                                // the join of :273 (rows live in LDS; the prefetched draws stay in flight)
        }
        if (sample && live) {                             // the walker's state after its update (:268-271); own row, own thread
            if (ra.chain != nullptr) {
                T* dst = reinterpret_cast<T*>(ra.chain) + (wslot * S + t) * (int64_t)a.ld;
#pragma unroll
                for (int d = 0; d < ND; ++d) if (d < ndim) dst[d] = (T)lpos[t * LS + d];
                if (a.ld > ndim) dst[ndim] = (T)0;                              // the pad column of an odd ndim
            }
            if (ra.chain_logp != nullptr) ra.chain_logp[wslot * S + t] = llogp[t];
            if constexpr (BlobTrait<Dens>::n > 0) {
                if (ra.chain_blob != nullptr) {
#pragma unroll 1
                    for (int i = 0; i < BlobTrait<Dens>::n; ++i)
                        ra.chain_blob[(slot * S + t) * BlobTrait<Dens>::n + i] = ra.blob[(int64_t)t * BlobTrait<Dens>::n + i];   // :270
                }
            }
            if (a.msum != nullptr) {
#pragma unroll
                for (int d = 0; d < ND; ++d) { const double v = lpos[t * LS + d]; s1[d] += v; s2[d] += v * v; }
            }
        }
      }   // generation of the batch
    }     // batch

    if (live) {
#pragma unroll
        for (int d = 0; d < ND; ++d) if (d < ndim) reinterpret_cast<T*>(a.pos)[(int64_t)t * a.ld + d] = (T)lpos[t * LS + d];
        a.logp[t] = llogp[t];
        if (nacc) a.naccept[t] += nacc;
    }
    if (a.msum != nullptr) {                              // per-thread sums -> per-dimension sums, in thread order (deterministic)
        for (int pass = 0; pass < 2; ++pass) {
            __syncthreads();                              // LDS is free again (rows written back / the previous pass consumed)
            if (live) {
#pragma unroll
                for (int d = 0; d < ND; ++d) lds[(size_t)d * S + t] = pass == 0 ? s1[d] : s2[d];
            }
            __syncthreads();
            if (t < ndim) {
                double acc = 0.0;
                for (int u = 0; u < S; ++u) acc += lds[(size_t)t * S + u];
                if (pass == 0) a.msum[t] += acc; else a.msumsq[t] += acc;
            }
        }
    }
}

template <class Dens, int ND, class T = double>
__global__ __launch_bounds__(1024) void resident_lane(const ResidentArgs a)
{
    resident_lane_body<Dens, ND, true, T>(a);
}

// ------------------------------------------------------------------------------------------------
// ... and TWO walkers per thread, for ensembles of 1026 .. 2048 walkers (as LDS allows): thread t owns walker t of the first half
// and walker S/2 + t of the second, so every thread moves a walker in both half-steps and 1024 threads carry 2048 walkers.  Same
// sampler, same draws (the table holds an entry per walker; a thread reads its two).  Without it an ensemble of 1100 walkers ran
// 4.7x slower than one of 1000 (launch per half-step, 2.9 us, against 0.62 resident).
// ------------------------------------------------------------------------------------------------
template <class Dens, int ND>
__device__ __forceinline__ void resident_lane2_body(const ResidentArgs& ra)
{
    static_assert(BlobTrait<Dens>::n == 0, "blobs: the one-walker-per-thread kernel");
    const IslandArgs& a = ra.is;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int S = ra.S, HS = S / 2;
    constexpr int ndim = ND;
    constexpr int LS = ND | 1;
    double* lpos  = lds;                                  // [S][LS]
    double* llogp = lds + (size_t)S * LS;                 // [S]
    const int t = threadIdx.x;
    const bool live = t < HS;
    for (int r = t; r < S; r += (int)blockDim.x) {
#pragma unroll
        for (int d = 0; d < ND; ++d) lpos[r * LS + d] = a.pos[(int64_t)r * a.ld + d];
        llogp[r] = a.logp[r];
    }
    __syncthreads();
    uint32_t nacc0 = 0u, nacc1 = 0u;
    double s1[ND], s2[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d) { s1[d] = 0.0; s2[d] = 0.0; }
    constexpr int B = ND <= 4 ? 2 : 1;                    // generations per batch of draws (two entries per thread and generation)
    static_assert(kDrawBatch % B == 0, "the host pads the table to kDrawBatch generations");
    const int tl = live ? t : 0;
    const int nb = (a.ngen + B - 1) / B;
    Draw cur0[B], cur1[B], nxt0[B], nxt1[B];
#pragma unroll
    for (int u = 0; u < B; ++u) { cur0[u] = Draw{}; cur1[u] = Draw{}; nxt0[u] = Draw{}; nxt1[u] = Draw{}; }
    if (nb > 0) {
#pragma unroll
        for (int u = 0; u < B; ++u) {
            nxt0[u] = draw_table_load(ra.draws, (int64_t)u * S + tl);
            nxt1[u] = draw_table_load(ra.draws, (int64_t)u * S + HS + tl);
        }
    }
    ThinClock clk(a.gen0, a.nburnin, a.nthin, ra.ring_slots);
#pragma unroll 1
    for (int b = 0; b < nb; ++b) {
#pragma unroll
      for (int u = 0; u < B; ++u) { cur0[u] = nxt0[u]; cur1[u] = nxt1[u]; }
      {
        const int bn = b + 1 < nb ? b + 1 : b;
#pragma unroll
        for (int u = 0; u < B; ++u) {
            nxt0[u] = draw_table_load(ra.draws, ((int64_t)bn * B + u) * S + tl);
            nxt1[u] = draw_table_load(ra.draws, ((int64_t)bn * B + u) * S + HS + tl);
        }
      }
#pragma unroll
      for (int sub = 0; sub < B; ++sub) {
        if (b * B + sub >= a.ngen) break;
        const bool hit = clk.tick(a.nthin);               // clk.n: the reference's loop variable (:245)
        const bool count = clk.n > 0;
        const int64_t slot = clk.q - 1;
        const bool sample = hit && slot < a.nsamples;                                            // :268
        const int64_t wslot = clk.wslot();
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            Draw dr = cur0[sub];                                                                 // :250, :252
            if (half) dr = cur1[sub];
            if (live) {                                                           // :247
                const int w = half * HS + t;
                const double* oth = lpos + ((1 - half) * HS + (int)dr.partner) * LS;
                const double* own = lpos + w * LS;
                typename Dens::Seq q;
                Dens::seq_init(q);
                double y[ND];
#pragma unroll
                for (int d = 0; d < ND; ++d) {
                    y[d] = fma(dr.z, own[d] - oth[d], oth[d]);                    // :255
                    Dens::seq_add(q, y[d], d, a.dp);
                }
                const double p1 = Dens::seq_finish(q, ndim, a.dp);                // :257
                if (accept_test(dr, p1, llogp[w])) {                              // :260
#pragma unroll
                    for (int d = 0; d < ND; ++d) lpos[w * LS + d] = y[d];         // :261
                    llogp[w] = p1;                                                // :262
                    if (count) { if (half) nacc1 += 1u; else nacc0 += 1u; }       // :265
                }
            }
            lds_barrier();                                // the join of :273
        }
        if (sample && live) {                             // both walkers' states after their updates (:268-271)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int w = k * HS + t;
                if (ra.chain != nullptr) {
                    double* dst = ra.chain + (wslot * S + w) * (int64_t)a.ld;
#pragma unroll
                    for (int d = 0; d < ND; ++d) dst[d] = lpos[w * LS + d];
                    if (a.ld > ndim) dst[ndim] = 0.0;                             // the pad column of an odd ndim
                }
                if (ra.chain_logp != nullptr) ra.chain_logp[wslot * S + w] = llogp[w];
                if (a.msum != nullptr) {
#pragma unroll
                    for (int d = 0; d < ND; ++d) { const double v = lpos[w * LS + d]; s1[d] += v; s2[d] += v * v; }
                }
            }
        }
      }
    }

    __syncthreads();
    for (int r = t; r < S; r += (int)blockDim.x) {
#pragma unroll
        for (int d = 0; d < ND; ++d) a.pos[(int64_t)r * a.ld + d] = lpos[r * LS + d];
        a.logp[r] = llogp[r];
    }
    if (live) {
        if (nacc0) a.naccept[t] += nacc0;
        if (nacc1) a.naccept[HS + t] += nacc1;
    }
    if (a.msum != nullptr) {                              // per-thread sums -> per-dimension sums, in thread order (deterministic)
        for (int pass = 0; pass < 2; ++pass) {
            __syncthreads();
            if (live) {
#pragma unroll
                for (int d = 0; d < ND; ++d) lds[(size_t)d * HS + t] = pass == 0 ? s1[d] : s2[d];
            }
            __syncthreads();
            if (t < ndim) {
                double acc = 0.0;
                for (int u = 0; u < HS; ++u) acc += lds[(size_t)t * HS + u];
                if (pass == 0) a.msum[t] += acc; else a.msumsq[t] += acc;
            }
        }
    }
}

template <class Dens, int ND>
__global__ __launch_bounds__(1024) void resident_lane2(const ResidentArgs a)
{
    resident_lane2_body<Dens, ND>(a);
}

}  // namespace kmc
