// kmc_generation.hpp -- ONE launch per generation for mid-size ensembles with short rows (exact; round 4).
//
// The multi-launch kernels (kmc_kernels.hpp) pay the dependent-launch boundary -- 1.2-1.35 us -- twice per generation, and for
// short rows that boundary IS the half-step: 2.5 us at 4 096 x 4 and at 16 384 x 4, whatever the size.  A join inside the launch
// costs more than the boundary (profiles/r04_join_probe.txt).  This kernel has no join at all.
//
// The draws of a walker-step are state-free (Philox keyed by (step, walker)), so a walker k of the SECOND half
// (reference src/samplers.jl:246-247, batch 2) does not have to wait for its partner j of the first half to be updated: it
// recomputes j's first-half-step move itself -- j's row and log-pdf as they were before the generation, j's own partner row (a
// second-half row, which nobody changes in the first half-step), j's draws -- through the SAME instructions the owner of j runs
// (one copy of the move's code, executed twice by second-half lanes: the recomputed row is the owner's row bit for bit by
// construction), and then makes its own move (:255-266) against the result.  Every thread reads only the INPUT copy of the state
// and writes its walker's row to the OUTPUT copy; the two copies swap per generation (the host keeps the state in the sampler's
// canonical arrays between kmc_sampler_run calls).  Cost: second-half walkers make two moves instead of one (1.5 x the arithmetic,
// 2.5 x the row reads -- nothing for rows of <= 64 bytes); gain: one boundary per generation instead of two.  Same chains,
// counters and log-pdfs as the two-launch kernels and the oracle (tests/test_gpu_generation.py); measured
// (scripts/probes/fused_probe.hip, profiles/r04_fused_probe.txt): 4 096 x 4 2.30 -> 1.44 us per half-step, 16 384 x 4 2.46 -> 1.67.
//
// One walker per lane, rows of ND <= 8 doubles in registers, the density through its sequential interface (element order =
// the oracle's order).  Blocks [0, nb) carry the second half -- the longer chain is dispatched first -- blocks [nb, 2 nb) the first.
#pragma once
#include "kmc_kernels.hpp"

namespace kmc {

constexpr int kGenerationTPB = 256;         // most threads per workgroup (the host launches one wave per workgroup while the waves are few: a mid-size ensemble
                                            //   then spreads over as many CUs as it has waves)

struct GenerationArgs {
    const double*     pin;          // [nwalkers][ld]   state before the generation
    double*           pout;         // [nwalkers][ld]   state after it (the other copy)
    const double*     lin;          // [nwalkers]
    double*           lout;
    uint32_t*         naccept;      // [nwalkers]
    const SchedEntry* sched;        // this generation's entry of the device table (graph replay), or nullptr
    SchedEntry        sched_inline; //   ... eager launches
    DrawConsts        dc;           // dc.nhalf = nwalkers / 2
    DensityParams     dp;
    uint32_t          h;            // walkers per half
    uint32_t          nb;           // workgroups per half
    int32_t           ld;           // row stride (doubles; ndim rounded up to even)
    int32_t           ndim;
    double*           chain;        // [nsamples][nwalkers][ld] or nullptr              (:269)
    double*           chain_logp;   // [nsamples][nwalkers] or nullptr                  (:271)
    double*           msum;         // [nwalkers][ld] per-walker sums of the stored samples (laid out like the rows; read out in walker order), or nullptr;
                                    //   generation_group with K == 2 and L = 8 / 16 / 32: per-WAVE accumulators [waves][NVL][64] in half_step_vec's transposed
                                    //   layout (kmc_kernels.hpp: FoldT) -- msumsq unused
    double*           msumsq;
    // lane-striped form (generation_group) only:
    uint32_t*         klast;        // [nwalkers] samples a walker's CURRENT value has already been credited for: its moments are sojourn-weighted like the
                                    //   two-launch kernels' (a value is credited, times the samples it stood for, when it is replaced; the read-out credits the rest)
    uint32_t*         glast;        // [nwalkers] 1 + the generation of the walker's last accepted move (0: none): a row is written to the output copy only
                                    //   when that copy does not hold it already (see generation_group_body)
};

// What the head of a wave's chain needs, as the kernel's LEADING scalar parameters: the build preloads the first 14 dwords of the
// arguments into SGPRs (-amdgpu-kernarg-preload-count), so the schedule entry's load is issued at wave entry instead of behind the
// load of the argument struct that used to carry its address -- two dependent scalar round trips in front of the first Philox block were
// 0.45 us of a 3.5 us launch (profiles/r04_generation_timeline.txt).  An eager launch (sched == nullptr) has its generation here too.
struct GenerationFront {
    const SchedEntry* sched;        // = GenerationArgs::sched
    const double*     pin;
    const double*     lin;
    double*           pout;
    uint32_t          seed_lo, seed_hi;
    uint32_t          h, nb;        // nb: workgroups per half in the low 30 bits, threads per workgroup / 64 - 1 above them (blockDim.x is a hidden kernel argument: read from
                                    // there it costs every wave a scalar round trip before its first instruction of substance -- 0.3-0.5 us of a launch of 3)
    int32_t           ld;           // row stride in the low 16 bits, ndim above them (a row of these kernels is at most 512 elements): both in front of the first loads
    uint32_t          gen;          // eager launches: the generation (sched == nullptr)
};
static_assert(sizeof(GenerationFront) == 56, "14 preloaded dwords");
#define KMC_GEN_FRONT_PARAMS const kmc::SchedEntry* g_sched, const double* g_pin, const double* g_lin, double* g_pout, uint32_t g_seed_lo, uint32_t g_seed_hi, \
                             uint32_t g_h, uint32_t g_nb, int32_t g_ld, uint32_t g_gen
#define KMC_GEN_FRONT_PACK kmc::GenerationFront{g_sched, g_pin, g_lin, g_pout, g_seed_lo, g_seed_hi, g_h, g_nb, g_ld, g_gen}
#define KMC_GEN_FRONT_TYPES const kmc::SchedEntry*, const double*, const double*, double*, uint32_t, uint32_t, uint32_t, uint32_t, int32_t, uint32_t
// kernarg image of (KMC_GEN_FRONT_PARAMS, const GenerationArgs): module launches of runtime-compiled kernels pass it as one block
struct GenerationLaunch {
    GenerationFront f;
    GenerationArgs  a;
};

// the schedule entry and the draw constants of a launch, from the preloaded head where the chain starts and from the struct for the rest
// The head of a wave's chain touches nothing but preloaded parameters (as in half_step_vec, kmc_kernels.hpp).  Table graph: one scalar round trip for the schedule entry,
// waited for HERE (the argument struct's fields ride the same round trip); eager form / updated graph (f.sched == nullptr): the generation is a preloaded parameter and
// Philox starts at wave entry -- the rest of the entry is in the argument struct and is first looked at when the rows are on their way (generation_schedule_late).
__device__ __forceinline__ SchedEntry generation_schedule_early(const GenerationFront& f)
{
    SchedEntry t{0, 0, 0u, 0u, {0u, 0u}};
    if (f.sched != nullptr) {
        typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
        u32x8 r;
        asm volatile("s_load_dwordx8 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(f.sched));
        t.gen     = (int64_t)(((uint64_t)r[1] << 32) | r[0]);
        t.slot    = (int64_t)(((uint64_t)r[3] << 32) | r[2]);
        t.flags   = r[4];
        t.nbefore = r[5];
    } else {
        t.gen = (int64_t)f.gen;
    }
    return t;
}
__device__ __forceinline__ SchedEntry generation_schedule_late(const GenerationFront& f, const GenerationArgs& a, const SchedEntry& early)
{
    // (the launch kind again, opaque to the optimiser: merged with the branch in generation_schedule_early it would pull the struct's first use -- and the wait for it -- in front of Philox)
    int eager_late = f.sched == nullptr ? 1 : 0;
    asm volatile("" : "+v"(eager_late));
    eager_late = __builtin_amdgcn_readfirstlane(eager_late);
    SchedEntry e = early;
    if (eager_late != 0) { e = a.sched_inline; e.gen = early.gen; }
    return e;
}
__device__ __forceinline__ DrawConsts generation_draw_consts_early(const GenerationFront& f)        // what Philox and the partner index need
{
    DrawConsts dc{};
    dc.seed_lo = f.seed_lo; dc.seed_hi = f.seed_hi; dc.nhalf = f.h;
    return dc;
}
__device__ __forceinline__ DrawConsts generation_draw_consts(const GenerationFront& f, const GenerationArgs& a)
{
    DrawConsts dc = a.dc;
    dc.seed_lo = f.seed_lo; dc.seed_hi = f.seed_hi; dc.nhalf = f.h;
    return dc;
}

template <int ND>
__device__ __forceinline__ void gen_load_row(const double* p, double (&x)[ND])
{
    // rows start on 16-byte boundaries (ld is even); an odd row length ends with one 8-byte element
#pragma unroll
    for (int c = 0; c < ND / 2; ++c) { const double2 t = reinterpret_cast<const double2*>(p)[c]; x[2 * c] = t.x; x[2 * c + 1] = t.y; }
    if constexpr (ND % 2 == 1) x[ND - 1] = p[ND - 1];
}
template <int ND, bool PAD = false>
__device__ __forceinline__ void gen_store_row(double* p, const double (&x)[ND])
{
#pragma unroll
    for (int c = 0; c < ND / 2; ++c) reinterpret_cast<double2*>(p)[c] = make_double2(x[2 * c], x[2 * c + 1]);
    if constexpr (ND % 2 == 1) {
        // the pad column of an odd ndim: 0 already in both copies of the state (never written); written with a chain row
        if constexpr (PAD) reinterpret_cast<double2*>(p)[ND / 2] = make_double2(x[ND - 1], 0.0);
        else p[ND - 1] = x[ND - 1];
    }
}

template <class Dens, int ND>
__device__ __forceinline__ void generation_lane_body(const GenerationFront& f, const GenerationArgs& a)
{
    static_assert(BlobTrait<Dens>::n == 0, "blobs: the multi-launch kernels");
    KMC_STAMP(0);                                                        // (-DKMC_PROBE builds only: scripts/probe_generation.py) wave entry
    const uint32_t nb = f.nb & 0x3fffffffu, tpb = 64u * ((f.nb >> 30) + 1u);
    const bool second = blockIdx.x < nb;
    const uint32_t i = (second ? blockIdx.x : blockIdx.x - nb) * tpb + threadIdx.x;
    if (i >= f.h) return;
    const SchedEntry sch_e = generation_schedule_early(f);
    const DrawConsts dcf = generation_draw_consts_early(f);
    const uint64_t step0 = 2ull * (uint64_t)sch_e.gen;                   // the first half-step of this generation (:246, batch 1)
#if KMC_PROBE_PINS
    asm volatile("" :: "s"(step0));
#endif
    KMC_STAMP(1);                                                        // the schedule entry has arrived
    const uint32_t me = (second ? f.h : 0u) + i;
    const size_t ld = (size_t)(f.ld & 0xffff);
    // level 1 = my own move; level 0 (second half only) = my partner's move in the first half-step
    const U4 mybits = draw_bits(dcf, step0 + (second ? 1u : 0u), me);
    const uint32_t mypartner = (second ? 0u : f.h) + draw_partner(dcf, mybits);     // :250
#if KMC_PROBE_PINS
    asm volatile("" :: "v"(mypartner));
#endif
    KMC_STAMP(2);                                                        // my Philox block is done
    double own[ND], oth[ND], myown[ND];
    double p0, myp0 = 0.0;
    U4 bits;
    if (second) {
        const uint32_t w = mypartner;                                    // a first-half walker: its move of step0
        bits = draw_bits(dcf, step0, w);
        const uint32_t jp = f.h + draw_partner(dcf, bits);             // its partner: a second-half row, unchanged by the first half-step
#if KMC_PROBE_PINS
        asm volatile("" :: "v"(jp));
#endif
        KMC_STAMP(3);                                                    // (second half) my partner's Philox block is done
        gen_load_row<ND>(f.pin + (size_t)jp * ld, oth);
        gen_load_row<ND>(f.pin + (size_t)w * ld, own);
        p0 = f.lin[w];
        gen_load_row<ND>(f.pin + (size_t)me * ld, myown);
        myp0 = f.lin[me];
    } else {
        bits = mybits;
        gen_load_row<ND>(f.pin + (size_t)mypartner * ld, oth);
        gen_load_row<ND>(f.pin + (size_t)me * ld, own);
        p0 = f.lin[me];
    }
    const SchedEntry sch = generation_schedule_late(f, a, sch_e);        // from here on the argument struct
    const DrawConsts dc = generation_draw_consts(f, a);
    const bool count = (sch.flags & kCount) != 0u, sample = (sch.flags & kSample) != 0u;
    double m1[ND], m2[ND];
    const bool moments = sample && a.msum != nullptr;
    if (moments) {                                                       // (issued here, used at the end)
        gen_load_row<ND>(a.msum + (size_t)me * ld, m1);
        gen_load_row<ND>(a.msumsq + (size_t)me * ld, m2);
    }
    // both moves' draws now, while the rows are on their way (z, (N-1) log z, log u: two logarithms each -- off the second move's chain)
    const Draw dr_mine = draw_finish(dc, mybits);                      // :252
    Draw dr_first = dr_mine;
    if (second) dr_first = draw_finish(dc, bits);                      // (uniform per workgroup; without the branch the four logarithms serialise: measured)
#if KMC_PROBE_PINS
    asm volatile("" :: "v"(dr_first.lu), "v"(dr_mine.lu));
#endif
    KMC_STAMP(4);                                                        // loads issued, logarithms done
    bool acc = false;
    double p1 = 0.0;
    double y[ND];
#pragma unroll 1
    for (int level = second ? 0 : 1; level < 2; ++level) {               // ONE copy of the move: my partner's and my own are the same instructions
        Draw dr = dr_mine;
        if (level == 0) dr = dr_first;
        if (level == 1) {
#if KMC_PROBE_PINS
            asm volatile("" :: "v"(oth[0]), "v"(own[0]));
#endif
            KMC_STAMP(5);                                                // rows arrived (first half) / my partner's move done (second half)
        }
        typename Dens::Seq q;
        Dens::seq_init(q);
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            y[d] = fma(dr.z, own[d] - oth[d], oth[d]);                   // :255
            Dens::seq_add(q, y[d], d, a.dp);
        }
        p1 = Dens::seq_finish(q, ND, a.dp);                              // :257
        acc = accept_test(dr, p1, p0);                                   // :260
        if (level == 0) {                                                // my partner as it stands after the first half-step; now my move
#pragma unroll
            for (int d = 0; d < ND; ++d) { oth[d] = acc ? y[d] : own[d]; own[d] = myown[d]; }
            p0 = myp0;
        }
    }
#pragma unroll
    for (int d = 0; d < ND; ++d) y[d] = acc ? y[d] : own[d];             // :261
    const double pnew = acc ? p1 : p0;                                   // :262
#if KMC_PROBE_PINS
    asm volatile("" :: "v"(pnew));
#endif
    KMC_STAMP(6);                                                        // my move is done
    gen_store_row<ND>(f.pout + (size_t)me * ld, y);
    a.lout[me] = pnew;
    // :265 (counted after burn-in only, :285-288).  A no-return atomic: nothing at the wave's end waits for the counter's old value (the
    // owner is the only one who adds; loading it early, with the rows, measured SLOWER -- its address arrives with the argument struct)
    if (acc && count) (void)__hip_atomic_fetch_add(&a.naccept[me], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (sample) {                                                        // the walker's state after its update, accepted or not (:268-271)
        const size_t row = (size_t)sch.slot * (2u * (size_t)f.h) + me;
        if (a.chain != nullptr) gen_store_row<ND, true>(a.chain + row * ld, y);
        if (a.chain_logp != nullptr) a.chain_logp[row] = pnew;
        if (moments) {
#pragma unroll
            for (int d = 0; d < ND; ++d) { m1[d] += y[d]; m2[d] += y[d] * y[d]; }
            gen_store_row<ND>(a.msum + (size_t)me * ld, m1);
            gen_store_row<ND>(a.msumsq + (size_t)me * ld, m2);
        }
    }
    KMC_STAMP(7);                                                        // the last store is issued
#ifdef KMC_PROBE
    {
        unsigned long long st[8];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        KMC_STAMP_READ(st[0], 80, 81); KMC_STAMP_READ(st[1], 82, 83); KMC_STAMP_READ(st[2], 84, 85); KMC_STAMP_READ(st[3], 86, 87);
        KMC_STAMP_READ(st[4], 88, 89); KMC_STAMP_READ(st[5], 90, 91); KMC_STAMP_READ(st[6], 92, 93); KMC_STAMP_READ(st[7], 94, 95);
        if (!second) st[3] = st[2];                                      // (first half: no second Philox block)
        if (threadIdx.x == 0 && blockIdx.x < 8192) for (int q = 0; q < 8; q += (KMC_PROBE == 2 ? 7 : 1)) g_probe[sch.gen & 1][blockIdx.x][q] = st[q];   // [generation parity][workgroup (its first wave)]
    }
#endif
}

template <class Dens, int ND>
__global__ __launch_bounds__(kGenerationTPB) void generation_lane(KMC_GEN_FRONT_PARAMS, const GenerationArgs a)
{
    generation_lane_body<Dens, ND>(KMC_GEN_FRONT_PACK, a);
}

// ------------------------------------------------------------------------------------------------
// The same generation for longer rows: a walker's row striped over the L lanes of a group, K chunks of two doubles per lane (lane j,
// chunk k holds elements 2 (k L + j), 2 (k L + j) + 1 -- the layout, the lane-striped density code and the group reduction of
// half_step_vec, so the log-pdfs are its bits).  Every lane of a group computes its walker's draws itself (redundantly: at these
// sizes most SIMDs are idle, and it keeps the chain free of cross-lane traffic); the accept decision is therefore the same in all of
// them.  Workgroups of blockDim.x / L walkers; blocks [0, nb) carry the second half.  Used where the ensemble's state is small enough
// that reading 2.5 x the rows and writing every row per generation costs less than the boundary saved (kmc_sampler.hip).
// ------------------------------------------------------------------------------------------------
template <class Dens, int L, int K>
__device__ __forceinline__ void generation_group_body(const GenerationFront& f, const GenerationArgs& a)
{
    static_assert(BlobTrait<Dens>::n == 0 && RowEvalTrait<Dens>::n == 0, "lane-striped densities only");
    KMC_STAMP(0);                                                        // (-DKMC_PROBE builds only: scripts/probe_timeline.py) wave entry
    const uint32_t nb = f.nb & 0x3fffffffu, tpb = 64u * ((f.nb >> 30) + 1u);
    const uint32_t gpb = tpb / L;                                        // walkers per workgroup
    const bool second = blockIdx.x < nb;
    const uint32_t i0 = (second ? blockIdx.x : blockIdx.x - nb) * gpb + threadIdx.x / L;
    const bool valid = i0 < f.h;
    const uint32_t i = valid ? i0 : f.h - 1u;                            // (idle groups of the last workgroup move the last walker and store nothing:
    const int j = (int)(threadIdx.x & (L - 1));                          //  the cross-lane sums want whole waves)
    const SchedEntry sch_e = generation_schedule_early(f);
    const DrawConsts dcf = generation_draw_consts_early(f);
    const uint64_t step0 = 2ull * (uint64_t)sch_e.gen;
    const uint32_t me = (second ? f.h : 0u) + i;
    const int ld = f.ld & 0xffff, ndim = f.ld >> 16;
    const double2 zero2 = make_double2(0.0, 0.0);
    // No masks on a row's last chunk (as in half_step_vec, kmc_kernels.hpp): a lane whose chunk lies past the row's end works on the row's LAST chunk instead -- it loads
    // what that chunk's real lane loads, computes the same bits and stores them to the same address; the densities select by element index against ndim.
    int ck[K];
#pragma unroll
    for (int k = 0; k < K; ++k) ck[k] = 2 * (k * L + j) < ld ? k * L + j : (ld >> 1) - 1;
    auto row_off = [&](uint32_t w) -> size_t { return (size_t)((uint64_t)w * (uint64_t)(uint32_t)ld); };
    auto row = [&](const double* base, uint32_t w, double2 (&x)[K]) {
        const double2* r = reinterpret_cast<const double2*>(base + row_off(w));
#pragma unroll
        for (int k = 0; k < K; ++k) x[k] = r[ck[k]];
    };
    const U4 mybits = draw_bits(dcf, step0 + (second ? 1u : 0u), me);
    const uint32_t mypartner = (second ? 0u : f.h) + draw_partner(dcf, mybits);     // :250
#if KMC_PROBE_PINS
    asm volatile("" :: "v"(mypartner));
#endif
    KMC_STAMP(1);                                                        // my Philox block is done
    double2 own[K], oth[K], myown[K];
    double p0, myp0 = 0.0;
    U4 bits;
    if (second) {
        const uint32_t w = mypartner;                                    // a first-half walker: its move of step0
        bits = draw_bits(dcf, step0, w);
        const uint32_t jp = f.h + draw_partner(dcf, bits);
        row(f.pin, jp, oth);
        row(f.pin, w, own);
        p0 = f.lin[w];
        row(f.pin, me, myown);
        myp0 = f.lin[me];
    } else {
        bits = mybits;
        row(f.pin, mypartner, oth);
        row(f.pin, me, own);
        p0 = f.lin[me];
#pragma unroll
        for (int k = 0; k < K; ++k) myown[k] = zero2;
    }
    KMC_STAMP(2);                                                        // rows requested
    const SchedEntry sch = generation_schedule_late(f, a, sch_e);        // from here on the argument struct
    const DrawConsts dc = generation_draw_consts(f, a);
    const bool count = (sch.flags & kCount) != 0u, sample = (sch.flags & kSample) != 0u;
    const bool moments = count && a.msum != nullptr;
    const uint32_t gl = a.glast[me];
    uint32_t kl = 0u;
    if (moments) kl = a.klast[me];
    // Moment accumulators.  K == 2, L = 8 / 16 / 32: one set per WAVE in half_step_vec's transposed layout (the wave's 64 / L walkers fold their credits with
    // the same reduce-scatter; 8 L / 64 doubles per lane) -- a per-walker pair of sums read with every row was 47 % of this kernel's read bytes at 16 384 x 64
    // (profiles/r05_generation_summary.json: 35.8 MB per generation, 16.8 of them the sums).  Other geometries: per-walker sums laid out like the rows.
    constexpr bool kFold = FoldT<L, K>::on;
    const int lane = (int)(threadIdx.x & 63u);
    const int64_t wave = (int64_t)blockIdx.x * (tpb >> 6) + (threadIdx.x >> 6);
    double2 m1[K], m2[K];
    double acct[4] = {0.0, 0.0, 0.0, 0.0};
    if constexpr (kFold) {
        if (moments) {
#pragma unroll
            for (int r = 0; r < FoldT<L, K>::NVL; ++r) acct[r] = a.msum[(wave * FoldT<L, K>::NVL + r) * 64 + lane];
        }
    } else {
        if (moments) { row(a.msum, me, m1); row(a.msumsq, me, m2); }                 // (issued with the rows, used -- and rewritten -- only if my move is accepted)
    }
    // :252 -- both moves' draws now (see generation_lane_body).  Their four logarithms are most of a lane's chain here (~0.2 us each: dependent
    // fp64 operations of a wave that has its SIMD to itself), and the lanes of a quad belong to one walker when L >= 4: each computes ONE of
    // them -- same function, same argument, hence the same bits as draw_finish -- and the quad shares the results.
    Draw dr_mine, dr_first;
    if constexpr (L >= 4) {
        dr_mine.partner = draw_partner(dc, mybits); dr_first.partner = draw_partner(dc, bits);
        const double t_m = fma(((double)mybits.y + 0.5) * 0x1.0p-32, dc.c1, dc.c0), t_f = fma(((double)bits.y + 0.5) * 0x1.0p-32, dc.c1, dc.c0);
        dr_mine.z = t_m * t_m; dr_first.z = t_f * t_f;
        const double ua_m = ((double)(((uint64_t)mybits.z << 20) | (uint64_t)(mybits.w >> 12)) + 0.5) * 0x1.0p-52;
        const double ua_f = ((double)(((uint64_t)bits.z << 20) | (uint64_t)(bits.w >> 12)) + 0.5) * 0x1.0p-52;
        const int q4 = j & 3;
        const double arg = q4 == 0 ? dr_mine.z : q4 == 1 ? ua_m : q4 == 2 ? dr_first.z : ua_f;
        const double lg = log_pos_normal(arg);
        dr_mine.t1 = dc.nm1 * dpp_f64<0x00>(lg);                         // quad_perm [0,0,0,0]
        dr_mine.lu = dpp_f64<0x55>(lg);                                  // quad_perm [1,1,1,1]
        dr_first.t1 = dc.nm1 * dpp_f64<0xAA>(lg);                        // quad_perm [2,2,2,2]
        dr_first.lu = dpp_f64<0xFF>(lg);                                 // quad_perm [3,3,3,3]
    } else {
        dr_mine = draw_finish(dc, mybits);
        dr_first = dr_mine;
        if (second) dr_first = draw_finish(dc, bits);
    }
    bool acc = false;
    double p1 = 0.0;
    double2 y[K];
#if KMC_PROBE_PINS
    asm volatile("" :: "v"(dr_mine.lu), "v"(dr_first.lu));
#endif
    KMC_STAMP(3);                                                        // draws finished (logarithms), accumulators requested
#pragma unroll 1
    for (int level = second ? 0 : 1; level < 2; ++level) {               // ONE copy of the move (see generation_lane_body)
        Draw dr = dr_mine;
        if (level == 0) dr = dr_first;
#pragma unroll
        for (int k = 0; k < K; ++k) {                                    // :255
            y[k].x = fma(dr.z, own[k].x - oth[k].x, oth[k].x);
            y[k].y = fma(dr.z, own[k].y - oth[k].y, oth[k].y);
        }
        if constexpr (MultiSumTrait<Dens>::n > 0) {                      // a function body feeding several sums over the elements
            double S[MultiSumTrait<Dens>::n];
            Dens::template frag_partial_n<L, K>(y, j, ndim, a.dp, S);
#pragma unroll
            for (int q = 0; q < MultiSumTrait<Dens>::n; ++q) S[q] = group_sum<L>(S[q]);
            p1 = Dens::finish_n(S, a.dp);                                // :257
        } else {
            p1 = Dens::finish(group_sum<L>(Dens::template frag_partial<L, K>(y, j, ndim, a.dp)), a.dp);   // :257
        }
        acc = accept_test(dr, p1, p0);                                   // :260 (the same bits, hence the same decision, in every lane of the group)
        if (level == 0) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                oth[k].x = acc ? y[k].x : own[k].x; oth[k].y = acc ? y[k].y : own[k].y;
                own[k] = myown[k];
            }
            p0 = myp0;
        }
    }
    // Streaming moments, sojourn-weighted (the two-launch kernels' rule, kmc_kernels.hpp): the value this move replaces is credited now, times the samples it
    // stood for; nothing is written for a walker that stays.  Round 4 rewrote both sums of every walker with every sample (7.5 x the state per generation in
    // all; 8 192 x 64 3.46 -> 2.59 us per half-step without that).  Tried instead of reading accumulators up front: no-return fp64 atomic adds on accept --
    // slower at every shape, the L2 retires about one 8-byte add per channel per 3-4 cycles (profiles/r05_generation_mid.txt).  `own` is my row before my move.
    const double wgt = (valid && acc && moments) ? (double)(sch.nbefore - kl) : 0.0;
#if KMC_PROBE_PINS
    asm volatile("" :: "v"(wgt));
#endif
    KMC_STAMP(4);                                                        // rows arrived, both moves done
    if constexpr (kFold) {
        if (moments && __ballot(wgt != 0.0) != 0ull) {                   // (wave-uniform: every lane takes part in the fold, idle groups with zeros)
            double2 ms[K], mq[K];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                ms[k] = make_double2(own[k].x * wgt, own[k].y * wgt);
                mq[k] = make_double2((own[k].x * own[k].x) * wgt, (own[k].y * own[k].y) * wgt);
            }
            double v[8];
            fold_scatter<L, K>(lane, ms, mq, v);
            double* slot = a.msum + (wave * FoldT<L, K>::NVL) * 64 + lane;
#pragma unroll
            for (int r = 0; r < FoldT<L, K>::NVL; ++r) store_wt(&slot[r * 64], acct[r] + v[r]);
        }
    }
    KMC_STAMP(5);                                                        // moments folded, accumulator stores issued
    if (!valid) return;
    if constexpr (!kFold) {
        if (wgt != 0.0) {
            double* s1 = a.msum + row_off(me);
            double* s2 = a.msumsq + row_off(me);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int e = 2 * ck[k];
                *reinterpret_cast<double2*>(&s1[e]) = make_double2(m1[k].x + own[k].x * wgt, m1[k].y + own[k].y * wgt);
                *reinterpret_cast<double2*>(&s2[e]) = make_double2(m2[k].x + (own[k].x * own[k].x) * wgt, m2[k].y + (own[k].y * own[k].y) * wgt);
            }
        }
    }
    if (acc && moments && j == 0) store_wt(&a.klast[me], sch.nbefore);
#pragma unroll
    for (int k = 0; k < K; ++k) { y[k].x = acc ? y[k].x : own[k].x; y[k].y = acc ? y[k].y : own[k].y; }   // :261
    const double pnew = acc ? p1 : p0;                                   // :262
    // The output copy already holds my row unless I moved in this generation or in the one before (it was last written two generations ago or
    // earlier, and is valid as long as nothing was accepted since: by induction from two equal copies at the start of a run, kmc_launch.hip).
    // A stale glast -- a restart -- can only ask for a write too many.
    if (acc || gl == (uint32_t)sch.gen) {
        double2* out = reinterpret_cast<double2*>(f.pout + row_off(me));
#pragma unroll
        for (int k = 0; k < K; ++k) store_wt(&out[ck[k]], y[k]);
        if (j == 0) store_wt(&a.lout[me], pnew);
    }
    if (acc && j == 0) {
        store_wt(&a.glast[me], (uint32_t)sch.gen + 1u);
        if (count) (void)__hip_atomic_fetch_add(&a.naccept[me], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // :265 (see generation_lane_body)
    }
    if (sample) {                                                        // :268-271
        const size_t srow = (size_t)sch.slot * (2u * (size_t)f.h) + me;
        if (a.chain != nullptr) {
            double2* dst = reinterpret_cast<double2*>(a.chain + srow * (size_t)ld);
#pragma unroll
            for (int k = 0; k < K; ++k) store_wt(&dst[ck[k]], y[k]);
        }
        if (a.chain_logp != nullptr && j == 0) store_wt(&a.chain_logp[srow], pnew);
    }
    KMC_STAMP(7);                                                        // the last store is issued
#ifdef KMC_PROBE
    {
        unsigned long long st[8] = {};
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        KMC_STAMP_READ(st[0], 80, 81); KMC_STAMP_READ(st[1], 82, 83); KMC_STAMP_READ(st[2], 84, 85); KMC_STAMP_READ(st[3], 86, 87);
        KMC_STAMP_READ(st[4], 88, 89); KMC_STAMP_READ(st[5], 90, 91); KMC_STAMP_READ(st[7], 94, 95);
        st[6] = st[5];
        if (threadIdx.x == 0 && blockIdx.x < 8192) for (int q = 0; q < 8; q += (KMC_PROBE == 2 ? 7 : 1)) g_probe[sch.gen & 1][blockIdx.x][q] = st[q];   // [generation parity][workgroup (its first wave)]
    }
#endif
}

template <class Dens, int L, int K>
__global__ __launch_bounds__(256) void generation_group(KMC_GEN_FRONT_PARAMS, const GenerationArgs a)
{
    generation_group_body<Dens, L, K>(KMC_GEN_FRONT_PACK, a);
}

}  // namespace kmc
