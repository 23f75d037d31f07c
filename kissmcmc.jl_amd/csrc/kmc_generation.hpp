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

constexpr int kGenerationTPB = 64;          // one wave per workgroup: a mid-size ensemble spreads over as many CUs as it has waves

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
    int32_t           pad_;
    double*           chain;        // [nsamples][nwalkers][ld] or nullptr              (:269)
    double*           chain_logp;   // [nsamples][nwalkers] or nullptr                  (:271)
    double*           msum;         // [ND][nwalkers] per-walker sums of the stored samples (read out in walker order), or nullptr
    double*           msumsq;
};

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
__device__ __forceinline__ void generation_lane_body(const GenerationArgs& a)
{
    static_assert(BlobTrait<Dens>::n == 0, "blobs: the multi-launch kernels");
    const bool second = blockIdx.x < a.nb;
    const uint32_t i = (second ? blockIdx.x : blockIdx.x - a.nb) * (uint32_t)kGenerationTPB + threadIdx.x;
    if (i >= a.h) return;
    const SchedEntry sch = a.sched != nullptr ? *a.sched : a.sched_inline;
    const uint64_t step0 = 2ull * (uint64_t)sch.gen;                     // the first half-step of this generation (:246, batch 1)
    const uint32_t me = (second ? a.h : 0u) + i;
    const size_t ld = (size_t)a.ld;
    // level 1 = my own move; level 0 (second half only) = my partner's move in the first half-step
    const U4 mybits = draw_bits(a.dc, step0 + (second ? 1u : 0u), me);
    const uint32_t mypartner = (second ? 0u : a.h) + draw_partner(a.dc, mybits);      // :250
    double own[ND], oth[ND], myown[ND];
    double p0, myp0 = 0.0;
    U4 bits;
    if (second) {
        const uint32_t w = mypartner;                                    // a first-half walker: its move of step0
        bits = draw_bits(a.dc, step0, w);
        const uint32_t jp = a.h + draw_partner(a.dc, bits);              // its partner: a second-half row, unchanged by the first half-step
        gen_load_row<ND>(a.pin + (size_t)jp * ld, oth);
        gen_load_row<ND>(a.pin + (size_t)w * ld, own);
        p0 = a.lin[w];
        gen_load_row<ND>(a.pin + (size_t)me * ld, myown);
        myp0 = a.lin[me];
    } else {
        bits = mybits;
        gen_load_row<ND>(a.pin + (size_t)mypartner * ld, oth);
        gen_load_row<ND>(a.pin + (size_t)me * ld, own);
        p0 = a.lin[me];
    }
    const bool count = (sch.flags & kCount) != 0u, sample = (sch.flags & kSample) != 0u;
    double m1[ND], m2[ND];
    const bool moments = sample && a.msum != nullptr;
    if (moments) {                                                       // (issued here, used at the end)
#pragma unroll
        for (int d = 0; d < ND; ++d) { m1[d] = a.msum[(size_t)d * (2u * a.h) + me]; m2[d] = a.msumsq[(size_t)d * (2u * a.h) + me]; }
    }
    bool acc = false;
    double p1 = 0.0;
    double y[ND];
#pragma unroll 1
    for (int level = second ? 0 : 1; level < 2; ++level) {               // ONE copy of the move: my partner's and my own are the same instructions
        const Draw dr = draw_finish(a.dc, bits);                         // :252
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
            bits = mybits;
        }
    }
#pragma unroll
    for (int d = 0; d < ND; ++d) y[d] = acc ? y[d] : own[d];             // :261
    const double pnew = acc ? p1 : p0;                                   // :262
    gen_store_row<ND>(a.pout + (size_t)me * ld, y);
    a.lout[me] = pnew;
    if (acc && count) a.naccept[me] += 1u;                               // :265 (counted after burn-in only, :285-288)
    if (sample) {                                                        // the walker's state after its update, accepted or not (:268-271)
        const size_t row = (size_t)sch.slot * (2u * (size_t)a.h) + me;
        if (a.chain != nullptr) gen_store_row<ND, true>(a.chain + row * ld, y);
        if (a.chain_logp != nullptr) a.chain_logp[row] = pnew;
        if (moments) {
#pragma unroll
            for (int d = 0; d < ND; ++d) {
                a.msum[(size_t)d * (2u * a.h) + me] = m1[d] + y[d];
                a.msumsq[(size_t)d * (2u * a.h) + me] = m2[d] + y[d] * y[d];
            }
        }
    }
}

template <class Dens, int ND>
__global__ __launch_bounds__(kGenerationTPB) void generation_lane(const GenerationArgs a)
{
    generation_lane_body<Dens, ND>(a);
}

}  // namespace kmc
