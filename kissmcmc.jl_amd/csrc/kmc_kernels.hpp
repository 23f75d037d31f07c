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
//       over the L lanes of a group (16 B per lane per chunk), so the own row and the randomly
//       drawn partner row are both read as fully coalesced 16-B-per-lane segments, and the
//       log-pdf is a DPP reduction over the group.  A wave owns W = (64/L)*ITER consecutive
//       walkers and works in two layouts:
//         "scalar" layout  lane (g,j), j < ITER, owns walker slot j*G+g: Philox, partner, z,
//                          log z, log u, p0, the accept test -- once per walker, not per lane;
//         "row" layout     group g, iteration it, owns slot it*G+g: loads, stretch move,
//                          density partial sums, stores.
//       Scalars travel scalar->row by ds_bpermute (partner, z), row->scalar for free (the lane
//       with j == it keeps the reduced log-pdf), and the accept bits come back as a ballot mask.
//       The proposal stays in registers until the accept decision and is stored only on accept.
//       What the kernel is built around (each measured): nothing in front of Philox waits for
//       memory (the first 14 dwords of the arguments are preloaded into SGPRs; an eager-form launch carries its
//       step there too); the argument struct is one scalar round trip; partner-row loads are in flight before the
//       two logarithms; every store is write-through, so the end-of-kernel write-back of the half-step boundary
//       finds nothing dirty; streaming moments are a reduce-scatter over the wave.
//   half_step_generic<Density>          any ndim; one walker per lane, scalar loops.  Used for ndim > 1024, for
//       host-evaluated densities and with KMC_PLAN=generic.
#pragma once
#include "kmc_device.hpp"

namespace kmc {

constexpr int kTPB = 256;   // threads per workgroup of the half-step kernels with long rows (waves are independent)
// Workgroup size of the vector kernels by lane-group width, from repeated A/B runs of forced sizes: L <= 8 (ndim <= 32,
// C2) 128 threads (-3 % against 256), L = 16 (C3) 64 threads (-2.5 %; 128 is 8 % SLOWER there with moments on), longer
// rows (C5) 256.
__host__ __device__ constexpr int vec_tpb(int L) { return L <= 8 ? 128 : L == 16 ? 64 : kTPB; }

// What the reference's loop variable n (src/samplers.jl:245) implies for one generation.
struct SchedEntry {
    int64_t  gen;     // generation index g in [0, G);  n = g + 1 - nburnin
    int64_t  slot;    // stored-sample index k when (flags & kSample)
    uint32_t flags;
    uint32_t nbefore; // number of samples taken by generations < gen (sojourn-weighted moments)
    uint32_t pad_[2]; // 32 bytes: one s_load_dwordx8 per entry
};
static_assert(sizeof(SchedEntry) == 32, "schedule_of loads an entry as eight dwords");
enum : uint32_t { kCount = 1u, kSample = 2u };

// ring_slots > 0 (KMC_STREAM_CHAIN): the device keeps only a ring of that many sample slots; sample k goes to slot k % ring_slots.
__host__ __device__ inline SchedEntry make_sched(int64_t gen, int64_t nburnin, int64_t nthin, int64_t nsamples, int64_t ring_slots = 0)
{
    SchedEntry e{gen, 0, 0u, 0u, {0u, 0u}};
    const int64_t n = gen + 1 - nburnin;
    if (n > 1) {
        const int64_t nb = (n - 1) / nthin;
        e.nbefore = (uint32_t)(nb < nsamples ? nb : nsamples);
    }
    if (n > 0) {
        e.flags |= kCount;                              // :265, counters restart at n == 0 (:285-288)
        if (n % nthin == 0) {                           // :268
            const int64_t k = n / nthin - 1;
            if (k < nsamples) { e.flags |= kSample; e.slot = ring_slots > 0 ? k % ring_slots : k; }
        }
    }
    return e;
}

struct HalfStepArgs {
    double*           pos;
    double*           logp;
    uint32_t*         naccept;
    const SchedEntry* sched_table;  // device table (always valid)
    SchedEntry        sched_inline; // used when sched_index < 0 (eager launches)
    int32_t           sched_index;  // entry of sched_table (graph replay), or -1
    int32_t           half;         // 0: update [0,h) against [h,2h); 1: swapped       (:247)
    int64_t           gw0;          // GLOBAL walker index of this launch's active walker 0 (keys the RNG)
    int64_t           own_row0;     // its row in pos / index in logp, naccept
    int64_t           oth_row0;     // row of the complementary half's walker 0 (in each peer's pos for P2P)
    int32_t           n_active;     // number of active walkers of this shard
    int32_t           ndim;
    int32_t           ld;           // row stride of pos / chain in ELEMENTS (double, or float with KMC_F32): ndim rounded up to even
    int32_t           hloc_shift;   // log2(hloc) when hloc is a power of two, else -1
    // peer-to-peer sharding (P2P kernels only): partner p of the complementary half lives on rank
    // p / hloc at row oth_row0 + p % hloc of that rank's pos
    uint32_t          hloc;
    int32_t           nranks;       // bits 0-7: ranks; bits 8-15: this rank; bit 16: KMC_P2P_PUSH (pack_ranks / p2p_nranks, p2p_me, p2p_push: no field of its own --
                                    //   the struct every half-step kernel takes stays as short as round 4 left it)
    double*           peer_pos[8];
    const unsigned long long* flags; // flags[r] = number of half-steps rank r has completed
    unsigned long long* err;         // set non-zero when a wait times out
    DrawConsts        dc;
    DensityParams     dp;
    double*           chain;        // [nsamples][chain_rows][ndim] or nullptr          (:269)
    double*           chain_logp;   // [nsamples][chain_rows] or nullptr                (:271)
    int64_t           chain_rows;   // rows per sample slot
    int64_t           chain_row0;   // row of this launch's first active walker in a slot
    double*           msum;         // per-thread moment accumulators or nullptr
    double*           msumsq;
    int64_t           macc_stride;  // threads in the accumulator grid
    uint32_t*         klast;        // vec kernels: per walker, samples already credited to the moments
    // draw ring (vec kernels): the per-walker-step draws {partner, z, (N-1) log z, log u} are pure functions of
    // (seed, step, walker), and in the scalar layout only ITER of a group's L lanes carry a walker -- so a wave
    // computes them for this step AND the walkers' next Q-1 steps at once (all lanes busy, same instruction
    // count) and parks the future ones here: [4 slots][rows] entries of 32 B {t1, lu | z, partner + step tag}.
    // The next Q-1 launches of these walkers load one entry instead of running Philox and two logarithms.
    double2*          ring;         // or nullptr
    int64_t           ring_rows;    // rows per slot
    int32_t           ring_slot;    // slot of this generation (generation & 3)
    int32_t           prop_ld;      // (host-evaluated densities) row stride of prop_out in doubles
    // host-evaluated densities (HostEval) only
    double*           prop_out;     // PROPOSE pass: proposals [n_active][prop_ld]; nothing else is touched.  Device memory, or -- small
                                    //   batches -- the caller-facing page-locked host array itself (dense rows), written over the link
    const double*     p1_in;        // ACCEPT pass: log-pdf of proposal i as evaluated by the host
    unsigned char*    acc_out;      // ACCEPT pass, optional: 1 where proposal i replaced its walker (:261), else 0
    // moment ring (long rows: L == 64, ndim > 128, where the accumulators are NOT prefetched): a wave with an accepted move
    // POSTS the replaced row and its weight into its next ring slot instead of reading, adding to and rewriting its
    // accumulator slots -- in a bandwidth-saturated launch a dependent round trip issued at the end of a wave queues
    // behind everybody's row loads (C5: the 4-8 % of waves that accept ended ~3 us after the rest).  moments_sweep
    // folds the posted entries into msum / msumsq between graph chunks, in order.  Entries posted / folded so far are
    // counted per wave (mcnt / mswept, read at wave entry); a ring without room falls back to the read-modify-write.
    double2*          mring;        // [waves][mring_depth][K][64]: the replaced row as the lanes hold it, or nullptr
    double*           mring_w;      // [waves][mring_depth]: its weight (samples it stood for)
    uint32_t*         mcnt;         // [waves]
    const uint32_t*   mswept;       // [waves]
    int32_t           mring_depth;
    // blobs of a body density (BodyBlobDensity, NB doubles per evaluation; one walker per lane kernels only)
    double*           blob;         // [rows][NB]: the blob of every walker's current position (blob0s, src/samplers.jl:210, :264)
    double*           chain_blob;   // [nsamples][chain_rows][NB] (reduce_blob!, :270) or nullptr
};

// The peer-to-peer exchange has TWO variants, both ordered by a separate signal kernel and neither resting on any cache state: PULL of the drawn partner
// rows from their owner, and PUSH (KMC_P2P_PUSH, round 5) of every accepted row into a local copy of the owner's shard on every peer -- in both the
// partner rows are read with system-scope loads (sc0 sc1: never served from the reader's L2, whether the line belongs to a peer's memory or to local
// memory a peer writes over the fabric), and every row store is write-through.  Push moves acc * h_loc rows per link and half-step, pull h_loc / P: push
// wins while the acceptance is below 1 / P (C4's 0.234: P = 2 25 against 54 us per link, P = 4 25 against 27, P = 8 25 against 13.6 -- DESIGN.md
// section 7).  (Rounds 1-4 carried four more variants -- lazy pull into local copies driven by per-row stamps, and the progress signal folded into the
// half-step kernel -- behind -DKMC_P2P_EXPERIMENTAL: bit-exact with every "peer" on one GPU, never run on two, and by section 7's arithmetic unable to
// move the bound; deleted in round 5, `git show b051b2c:kissmcmc.jl_amd/csrc/kmc_kernels.hpp`.)
__host__ __device__ inline int32_t pack_ranks(int nranks, int me, bool push) { return (int32_t)(nranks | (me << 8) | (push ? 1 << 16 : 0)); }
__device__ __forceinline__ int  p2p_nranks(const HalfStepArgs& a) { return a.nranks & 0xff; }
__device__ __forceinline__ int  p2p_me(const HalfStepArgs& a) { return (a.nranks >> 8) & 0xff; }
__device__ __forceinline__ bool p2p_push(const HalfStepArgs& a) { return ((a.nranks >> 16) & 1) != 0; }

// The fields a wave needs before it can issue its first loads travel as LEADING SCALAR kernel parameters, ahead
// of the argument struct: built with -mllvm -amdgpu-kernarg-preload-count=14, gfx950 delivers them in SGPRs at
// wave launch (kernarg preload), so the own-row loads, the schedule entry and the Philox block do not wait for
// the kernarg fetch (one scalar-memory round trip right after the kernel boundary).  Without the flag they are
// ordinary kernel arguments -- same code, same results.
struct HalfStepFront {
    double*           pos;        // = HalfStepArgs::pos
    const SchedEntry* sched;      // this launch's schedule entry (sched_table + sched_index), or nullptr: eager launch, the
                                  //   entry travels in the args (sched_inline) and the step below
    const double2*    ring_now;   // draw ring, slot of THIS generation: entry of row r at ring_now[2 r] (nullptr: no ring)
    double*           logp;       // = HalfStepArgs::logp, the head of the per-walker block {logp[nrows], naccept[nrows],
                                  //   klast[nrows]} (one allocation): a walker's log-pdf and counters are requested at wave
                                  //   entry, with its row, instead of one round trip after the argument struct has arrived
    uint32_t          gw0;        // = HalfStepArgs::gw0 (walker indices fit 31 bits); own_row0 = P2P ? half * n_active : gw0
    uint32_t          nact_half;  // n_active | half << 31; the complementary half starts at row (1 - half) * (P2P ? n_active : nhalf)
    uint32_t          seed_lo, seed_hi, nhalf;
    uint32_t          step;       // sched == nullptr (eager launch): 2 * generation + half, known at launch time (< 2^32)
    __host__ __device__ int32_t n_active() const { return (int32_t)(nact_half & 0x7fffffffu); }
    __host__ __device__ int32_t half() const { return (int32_t)(nact_half >> 31); }
};                                // 14 dwords: all of it is preloaded (16 user SGPRs - 2 for the kernarg pointer)
#define KMC_FRONT_PARAMS double* f_pos, const kmc::SchedEntry* f_sched, const double2* f_ring_now, double* f_logp, uint32_t f_gw0, \
                         uint32_t f_nact_half, uint32_t f_seed_lo, uint32_t f_seed_hi, uint32_t f_nhalf, uint32_t f_step
#define KMC_FRONT_PACK kmc::HalfStepFront{f_pos, f_sched, f_ring_now, f_logp, f_gw0, f_nact_half, f_seed_lo, f_seed_hi, f_nhalf, f_step}
#define KMC_FRONT_TYPES double*, const kmc::SchedEntry*, const double2*, double*, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t

// Graph replay: one scalar load of the whole 32-byte entry (s_load_dwordx8; the scalar cache is invalidated at
// kernel start and the table is only written by advance_schedule between launches) -- one round trip, issued
// together with the rest of the argument struct.  Eager launch (f.sched == nullptr): the entry is in the args and
// the step among the preloaded parameters, so Philox starts at wave entry without waiting for memory at all.
__device__ __forceinline__ SchedEntry schedule_entry(const HalfStepFront& f)
{
    typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
    u32x8 r;
    asm volatile("s_load_dwordx8 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(f.sched));
    SchedEntry t;
    t.gen     = (int64_t)(((uint64_t)r[1] << 32) | r[0]);
    t.slot    = (int64_t)(((uint64_t)r[3] << 32) | r[2]);
    t.flags   = r[4];
    t.nbefore = r[5];
    return t;
}
__device__ __forceinline__ SchedEntry schedule_of(const HalfStepFront& f, const HalfStepArgs& a)
{
    if (f.sched == nullptr) return a.sched_inline;
    return schedule_entry(f);
}

// Stores of the half-step kernels are WRITE-THROUGH (sc0 sc1): the line goes to memory while the kernel is still
// running instead of sitting dirty in the XCD's L2 until the end-of-kernel write-back, which is part of the
// dependent-kernel boundary every half-step pays (C2: 4.27 -> 3.92 us per half-step; the next launch's readers sit
// on other XCDs and must get the data from memory anyway; ordinary stores were the A/B build of round 1).
// The s_nop covers the store-data hazard the compiler cannot see through the asm.
__device__ __forceinline__ void store_wt(double2* p, const double2& v)
{
    typedef double v2d __attribute__((ext_vector_type(2)));
    const v2d t = {v.x, v.y};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(p), "v"(t) : "memory");
}
__device__ __forceinline__ void store_wt(double* p, double v)
{
    asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void store_wt(uint32_t* p, uint32_t v)
{
    asm volatile("global_store_dword %0, %1, off sc0 sc1\n\ts_nop 0" :: "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void store_wt(unsigned char* p, unsigned char v)
{
    const uint32_t t = v;
    asm volatile("global_store_byte %0, %1, off sc0 sc1\n\ts_nop 0" :: "v"(p), "v"(t) : "memory");
}
// Row load: every row is read once per launch (a non-temporal variant was measured in round 1: no gain).
__device__ __forceinline__ double2 load_row16(const double2* p)
{
    return *p;
}

// Storage type T of the walker rows and the chain: double (KMC_F64) or float (KMC_F32: half the row bytes; the
// arithmetic stays double -- a chunk is widened on load, and a proposal is rounded to single BEFORE its log-density
// is evaluated, so a stored row and its stored log-pdf always belong together).
template <class T> struct RowOf;
template <> struct RowOf<double> { using V2 = double2; };
template <> struct RowOf<float>  { using V2 = float2; };
__device__ __forceinline__ void store_wt(float2* p, const float2& v)
{
    typedef float v2f __attribute__((ext_vector_type(2)));
    const v2f t = {v.x, v.y};
    asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(p), "v"(t) : "memory");
}
__device__ __forceinline__ double2 load_row(const double2* p) { return load_row16(p); }
__device__ __forceinline__ double2 load_row(const float2* p) { const float2 v = *p; return make_double2((double)v.x, (double)v.y); }
__device__ __forceinline__ void store_row(double2* p, const double2& v) { store_wt(p, v); }
__device__ __forceinline__ void store_row(float2* p, const double2& v) { store_wt(p, make_float2((float)v.x, (float)v.y)); }
template <class T> __device__ __forceinline__ double as_stored(double v)
{
    if constexpr (sizeof(T) == 4) return (double)(float)v; else return v;
}

__device__ __forceinline__ double2 sel2(bool c, const double2& a, const double2& b)
{
    return make_double2(c ? a.x : b.x, c ? a.y : b.y);
}

// ------------------------------------------------------------------------------------------
// Vector kernel.
// ------------------------------------------------------------------------------------------
// Fold the wave's G groups (same dimensions, different walkers) and add the result to that wave's accumulator
// slots.  Two forms:
//   * transposed (K == 2, L = 8/16/32 -- ndim 17..128): a reduce-scatter over the groups.  At each butterfly level a
//     lane keeps half of its values and hands the other half to its partner (v_permlane{16,32}_swap do exactly this
//     exchange in one instruction per word), so the 8 per-lane sums {sum x, sum x^2} x {4 elements} shrink to
//     8 L / 64 values per lane, every lane ends up owning a different piece, and the accumulator update is one
//     coalesced 8-byte read-modify-write per lane: a third of the cross-lane instructions of the plain butterfly.
//     Lane l keeps, as its r-th value, original index  L = 8: 4 b3 + 2 b4 + b5;  L = 16: 4 b4 + 2 b5 + r;
//     L = 32: 4 b5 + r  (b_i = bit i of l); index = which * 4 + k * 2 + xy, element 2 (k L + l mod L) + xy.
//     Slot of (wave w, r, lane l): msum[(w * NVL + r) * 64 + l]; msumsq is unused.
//   * plain: every value folded into group 0, whose lanes update [K][threads] double2 slots.
template <int L, int K>
struct FoldT {
    static constexpr bool on = (K == 2) && (L == 8 || L == 16 || L == 32);
    static constexpr int NVL = on ? (8 * L) / 64 : 1;
};

__device__ __forceinline__ double swap16_sum(double a, double b)
{   // even 16-lane rows: a + a of the odd partner row; odd rows: b + b of the even partner row
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ double swap32_sum(double a, double b)
{   // lanes < 32: a + a of lane + 32; lanes >= 32: b + b of lane - 32
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}

// the transposed fold itself: afterwards v[0 .. NVL) are this lane's pieces of the wave's sums (see above)
template <int L, int K>
__device__ __forceinline__ void fold_scatter(int lane, const double2 (&ms)[K], const double2 (&mq)[K], double (&v)[8])
{
    static_assert(FoldT<L, K>::on, "K == 2, L = 8 / 16 / 32");
    {
        v[0] = ms[0].x; v[1] = ms[0].y; v[2] = ms[1].x; v[3] = ms[1].y; v[4] = mq[0].x; v[5] = mq[0].y; v[6] = mq[1].x; v[7] = mq[1].y;
        int n = 8;
        if constexpr (L == 8) {                          // lane ^ 8, inside the 16-lane row
            const bool hi = (lane & 8) != 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                // (both candidates as opaque register values first: an older compiler -- the comgr a PyTorch wheel bundles, which is
                //  what hiprtc resolves to inside a torch process -- otherwise folds "hi ? v[i + 4] : v[i]" into a dynamically
                //  indexed read of v[] and emits an 8-way select chain per value: +150 instructions, -20 % on runtime-compiled kernels)
                double lo_v = v[i], hi_v = v[i + 4];
                asm volatile("" : "+v"(lo_v), "+v"(hi_v));
                const double keep = hi ? hi_v : lo_v, send = hi ? lo_v : hi_v;
                v[i] = keep + dpp_f64<0x128>(send);      // row_ror:8
            }
            n = 4;
        }
        if constexpr (L <= 16) {
#pragma unroll
            for (int i = 0; i < 4; ++i) if (i < n / 2) v[i] = swap16_sum(v[i], v[i + n / 2]);
            n /= 2;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) if (i < n / 2) v[i] = swap32_sum(v[i], v[i + n / 2]);
    }
}

template <int L, int K, bool HAVE_OLD>
__device__ __forceinline__ void accumulate_wave(double* msum, double* msumsq, int64_t stride, int tid, int g,
                                                double2 (&ms)[K], double2 (&mq)[K],
                                                const double2 (&olds)[K], const double2 (&oldq)[K], const double (&oldt)[4])
{
    if constexpr (FoldT<L, K>::on) {
        constexpr int NVL = FoldT<L, K>::NVL;
        const int lane = tid & 63;
        double v[8];
        fold_scatter<L, K>(lane, ms, mq, v);
        double* slot = msum + ((int64_t)(tid >> 6) * NVL) * 64 + lane;
#pragma unroll
        for (int r = 0; r < NVL; ++r) store_wt(&slot[r * 64], (HAVE_OLD ? oldt[r] : slot[r * 64]) + v[r]);
        return;
    }
    if constexpr (L < 64) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            ms[k].x = wave_fold<L>(ms[k].x); ms[k].y = wave_fold<L>(ms[k].y);
            mq[k].x = wave_fold<L>(mq[k].x); mq[k].y = wave_fold<L>(mq[k].y);
        }
    }
    if (g == 0) {
        double2* s = reinterpret_cast<double2*>(msum);
        double2* q = reinterpret_cast<double2*>(msumsq);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int64_t idx = (int64_t)k * stride + tid;
            const double2 sv = HAVE_OLD ? olds[k] : s[idx], qv = HAVE_OLD ? oldq[k] : q[idx];
            if constexpr (HAVE_OLD) {
                store_wt(&s[idx], make_double2(sv.x + ms[k].x, sv.y + ms[k].y));
                store_wt(&q[idx], make_double2(qv.x + mq[k].x, qv.y + mq[k].y));
            } else {      // long rows: 16 KB per accepted wave -- plain stores (write-through measured slower here)
                s[idx] = make_double2(sv.x + ms[k].x, sv.y + ms[k].y);
                q[idx] = make_double2(qv.x + mq[k].x, qv.y + mq[k].y);
            }
        }
    }
}

// Bounded wait until every rank has completed `need` half-steps (P2P only).  flags[] is this
// rank's fine-grained progress array, written by the peers' signal kernels over xGMI.
__device__ __forceinline__ void wait_for_peers(const HalfStepArgs& a, unsigned long long need, int lane)
{
    const int nranks = p2p_nranks(a);
    bool ok = lane >= nranks ||
              __hip_atomic_load(a.flags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) >= need;
    unsigned spins = 0;
    // once a wait has timed out the run is invalid: do not pay the timeout again at every step
    if (__hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull) ok = true;
    while (!__all(ok)) {
        __builtin_amdgcn_s_sleep(2);
        ok = lane >= nranks ||
             __hip_atomic_load(a.flags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) >= need;
        if (++spins > 30000000u) {                // ~10 s: a peer died -- flag it and fall through
            if (lane == 0) __hip_atomic_store(a.err, need + 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
    }
    // The fence only pins the compiler's load order.  Coherence of the PEER ROWS does not rest on cache state: the pull
    // kernels read them with system-scope loads (load_row_sys: sc0 sc1 -- a line of another agent's memory is fetched
    // from its owner, never served from this XCD's L2), because whether the kernel-start invalidate of a launch inside a
    // hipGraph chain covers remote memory depends on the acquire scope the runtime puts into the AQL packet, and that
    // cannot be observed with every "peer" sharing one GPU's L2.  (A system-scope acquire fence here instead --
    // buffer_inv sc0 sc1 in every workgroup -- was measured: 24 against 11 us per half-step at 2 x 65 536 walkers.)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// A 16-byte piece of a peer's row, read at SYSTEM scope (two relaxed 8-byte atomic loads: global_load_dwordx2 sc0 sc1).
__device__ __forceinline__ double2 load_row_sys(const double2* p)
{
    const unsigned long long* q = reinterpret_cast<const unsigned long long*>(p);
    const unsigned long long lo = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned long long hi = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return make_double2(__longlong_as_double((long long)lo), __longlong_as_double((long long)hi));
}

#ifdef KMC_PROBE   // diagnostic build only: per-wave 100 MHz timestamps of the last launch of each half
// Eight stamps per wave WITHOUT any wait at the stamp: s_memrealtime writes its result asynchronously into a fixed SGPR
// pair above everything the kernels allocate (the C2 kernel uses 64 SGPRs; the clobber makes the pair part of the
// kernel's allocation), and all eight are collected after one s_waitcnt at the very end -- a stamp that waited
// (lgkmcnt(0)) would also wait for the argument struct's scalar loads and move the very thing it measures.
static __device__ unsigned long long g_probe[2][8192][8];
// -DKMC_PROBE=2, the LIGHT probe: only wave entry and "last store issued" are stamped and nothing pins the schedule at the other six points --
// body / boundary of a launch at (nearly) the production kernel's own period, the duration source of the profile records (scripts/summarize_r04.py).
#define KMC_STAMP_AT(lo, hi) asm volatile("s_memrealtime s[" #lo ":" #hi "]" ::: "s" #lo, "s" #hi, "memory")
#define KMC_STAMP(i) KMC_STAMP_##i
#define KMC_STAMP_0 KMC_STAMP_AT(80, 81)
#define KMC_STAMP_7 KMC_STAMP_AT(94, 95)
#if KMC_PROBE == 2
#define KMC_PROBE_PINS 0
#define KMC_STAMP_1 do { } while (0)
#define KMC_STAMP_2 do { } while (0)
#define KMC_STAMP_3 do { } while (0)
#define KMC_STAMP_4 do { } while (0)
#define KMC_STAMP_5 do { } while (0)
#define KMC_STAMP_6 do { } while (0)
#define KMC_STAMP_READ(dst, lo, hi) do { if ((lo) == 80 || (lo) == 94) asm volatile("s_mov_b64 %0, s[" #lo ":" #hi "]" : "=s"(dst)); else (dst) = 0ull; } while (0)
#else
#define KMC_PROBE_PINS 1
#define KMC_STAMP_1 KMC_STAMP_AT(82, 83)
#define KMC_STAMP_2 KMC_STAMP_AT(84, 85)
#define KMC_STAMP_3 KMC_STAMP_AT(86, 87)
#define KMC_STAMP_4 KMC_STAMP_AT(88, 89)
#define KMC_STAMP_5 KMC_STAMP_AT(90, 91)
#define KMC_STAMP_6 KMC_STAMP_AT(92, 93)
#define KMC_STAMP_READ(dst, lo, hi) asm volatile("s_mov_b64 %0, s[" #lo ":" #hi "]" : "=s"(dst))
#endif
#else
#define KMC_PROBE_PINS 0
#define KMC_STAMP(i) do { } while (0)
#endif

// RAGGED = false: ndim == 2*L*K exactly (row stride and every mask fold at compile time);
// RAGGED = true : ndim < 2*L*K, runtime ndim and row stride (preloaded, see the body), tail chunks folded onto the row's last chunk.
template <class Dens, int L, int K, int ITER, bool P2P, bool RAGGED, class T = double>
__device__ __forceinline__ void half_step_vec_body(const HalfStepFront& f, const HalfStepArgs& a)
{
    static_assert(L >= 1 && L <= 64 && (L & (L - 1)) == 0, "L must be a power of two <= 64");
    static_assert(!P2P || sizeof(T) == 8, "the peer-to-peer kernels keep double rows");
    using V2 = typename RowOf<T>::V2;                   // one chunk = two consecutive elements of a row
    T* const posT = reinterpret_cast<T*>(f.pos);
    static_assert(ITER >= 1 && ITER <= L, "a group's scalar lanes must cover its iterations");
    constexpr int G = 64 / L;          // groups = walkers in flight per wave
    constexpr int W = G * ITER;        // walkers per wave
    // Ragged rows: the row stride stands in front of the very first loads and ndim in front of the log-density.  Read from the argument struct they cost a scalar round
    // trip each where the compiler happens to put the wait (measured: in front of the own rows, and again -- sharing a counter with ds_bpermute -- in front of the partner
    // rows) that the exact-size kernels do not pay: everything in front of THEIR loads is a preloaded parameter.  So ndim travels among the preloaded parameters too, in the
    // 16 bits above a device address (kmc_launch.hip: front_of packs it for exactly these kernels); ld = ndim rounded up to even (kmc_sampler_create).
    // (pointer arithmetic, not an integer cast back: the loads stay global_load -- a flat_load would count on lgkmcnt and stall the scalar pipeline's waits)
    const uint64_t ndim_tag = RAGGED ? reinterpret_cast<uint64_t>(f.logp) >> 48 : 0ull;
    double* const logp_p = RAGGED ? reinterpret_cast<double*>(reinterpret_cast<char*>(f.logp) - (ndim_tag << 48)) : f.logp;
    const int ndim = RAGGED ? (int)ndim_tag : 2 * L * K;
    const int64_t ld = RAGGED ? (int64_t)((ndim + 1) & ~1) : (int64_t)(2 * L * K);
    const int tid   = blockIdx.x * vec_tpb(L) + threadIdx.x;
    const int lane  = threadIdx.x & 63;
    const int j     = lane & (L - 1);
    const int g     = lane / L;
    const int gbase = lane & ~(L - 1);                  // first lane of this group
    const int w0    = (tid >> 6) * W;                   // first active index of this wave
    const int nact  = f.n_active();
    const int half  = f.half();
    const int64_t own_row0 = P2P ? (int64_t)half * nact : (int64_t)f.gw0;    // row of active walker 0 in pos / logp / naccept
    bool cv[K];                                         // chunk k of this lane lies inside the row
#pragma unroll
    for (int k = 0; k < K; ++k) cv[k] = !RAGGED || 2 * (k * L + j) < (int)ld;
    // Ragged rows are loaded AND stored without masks: a lane whose chunk lies past the row's end works on the row's LAST chunk instead (same cache line; no exec-mask
    // region around every load and store, nothing to zero).  It then computes what the chunk's real lane computes, from the same inputs, and stores the same bits to the
    // same address; what it holds never counts -- the densities select by element index against ndim, the moment read-out stops at ndim.
    // Row offsets: rows < 2^31 and ld < 2^31, one 32 x 32 -> 64 multiply instead of the 64-bit product.
    int ck[K];
#pragma unroll
    for (int k = 0; k < K; ++k) ck[k] = cv[k] ? k * L + j : (int)(ld >> 1) - 1;
    auto row_off = [&](int64_t row) -> int64_t { return RAGGED ? (int64_t)((uint64_t)(uint32_t)row * (uint64_t)(uint32_t)ld) : row * ld; };
    const double2 zero2 = make_double2(0.0, 0.0);
    KMC_STAMP(0);                                       // wave entry

    // ---- scalar layout: one walker per lane.  Lane (g, j) carries walker slot js = j % ITER of its group and,
    //      when draws are computed, the walker's q-th next step, q = j / ITER < Q; the lanes with q == 0
    //      (j < ITER) feed this launch -------------------------------------------------------------------
    constexpr int Q = (L / ITER) >= 4 ? 4 : (L / ITER);                 // steps drawn per heavy launch
    // worth it when a wave carries few walkers (long rows): at <= 8 walkers per wave the per-walker scalar work
    // dominates the wave's instruction count; with more (C2: 16) the extra load in the chain costs what it saves
    // (with the eager-form launches, where Philox starts at wave entry, it is worth +-2 %; it still pays +5 % under
    //  the table graph, e.g. for the P2P shards.  L = 64: slightly negative, off)
    // (round 2, one more bounded try for L = 8 -- C2 -- in both launch modes: 4.83 against 4.27 us per half-step under the table
    //  graph, 4.70 against 3.99 with the step preloaded: the ring entry is one more dependent load in front of the partner
    //  row, and at 16 walkers per wave the Philox it replaces was already hidden.  Dropped; profiles/NOTES.md.)
    constexpr bool kRing = Q >= 2 && L >= 16 && L <= 32;
    const int64_t oth_row0 = (int64_t)(1 - half) * (int64_t)(P2P ? (uint32_t)nact : f.nhalf);
    const int  jq     = j / ITER, js = j - jq * ITER;
    const bool useA   = jq == 0;
    const int  iA     = w0 + (jq < Q ? js : 0) * G + g;
    const bool validA = useA && (iA < nact);
    const int      iAc = iA < nact ? iA : nact - 1;
    const int64_t  rowA = own_row0 + iAc;                                // row in pos / index in logp, naccept
    const bool ring_on = kRing && f.ring_now != nullptr;
    double2 e0 = zero2, e1 = zero2;                                     // this launch's parked draws, if any
    if constexpr (kRing) {
        if (ring_on && useA) { e0 = f.ring_now[2 * rowA]; e1 = f.ring_now[2 * rowA + 1]; }
    }

    // ---- row layout: own rows of every iteration (independent of the random draws) ----------
    bool    validB[ITER];
    double2 xc[ITER][K], xo[ITER][K];
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const int i = w0 + it * G + g;
        validB[it] = i < nact;
        const V2* own = reinterpret_cast<const V2*>(posT + row_off(own_row0 + (validB[it] ? i : nact - 1)));
#pragma unroll
        for (int k = 0; k < K; ++k) xc[it][k] = load_row(&own[ck[k]]);
    }

    // ---- the walker's log-pdf and counters: same block, addressed from the preloaded parameters alone -------------
    const int64_t nrows_blk = 2 * (int64_t)(P2P ? (uint32_t)nact : f.nhalf);
    uint32_t* const naccept_p = reinterpret_cast<uint32_t*>(logp_p + nrows_blk);
    uint32_t* const klast_p = naccept_p + nrows_blk;
    const double   p0 = logp_p[rowA];
    const uint32_t na = naccept_p[rowA];
    const uint32_t kl = klast_p[rowA];

    // ---- the step, then Philox: nothing here touches the argument struct.  Eager launch: the step is a preloaded
    //      parameter, so the partner index is known without any memory access; graph replay: one scalar round trip
    //      for the schedule entry (the struct's fields ride the same round trip, see below) -----------------------
    const bool eager = f.sched == nullptr;
    SchedEntry sch_t{0, 0, 0u, 0u, {0u, 0u}};
    if (!eager) sch_t = schedule_entry(f);
    const uint64_t step = eager ? (uint64_t)f.step : 2ull * (uint64_t)sch_t.gen + (uint64_t)half;
    DrawConsts dcf{};                                                   // what Philox and the partner index need
    dcf.seed_lo = f.seed_lo; dcf.seed_hi = f.seed_hi; dcf.nhalf = f.nhalf;
    // parked draws are valid iff they carry this step's tag (wave-uniform decision)
    const uint32_t e1y_lo = (uint32_t)__double2loint(e1.y), e1y_hi = (uint32_t)__double2hiint(e1.y);
    const bool fresh = ring_on && __all(!useA || e1y_hi == (uint32_t)step);
    U4 bits{0u, 0u, 0u, 0u};
    uint32_t partnerA = e1y_lo;                                         // :250
    if (!fresh) {
        bits = draw_bits(dcf, step + 2ull * (uint64_t)jq, (uint64_t)f.gw0 + (uint64_t)iAc);   // RNG keyed by the GLOBAL walker index
        partnerA = draw_partner(dcf, bits);
    }
#if KMC_PROBE_PINS
    asm volatile("" :: "v"(partnerA));
#endif
    KMC_STAMP(1);                                       // Philox done: the partner index is known

    // ---- scalar -> row: the partner of slot it*G+g lives in lane gbase+it.  The partner-row loads of the first
    //      half of the iterations go out, then the first logarithm, then the other half, then the second
    //      logarithm: measured best at C2 (all loads first: +0.15 us per half-step; loads after both logs: same),
    //      and pinned with scheduling barriers because the compiler's own placement moves with unrelated edits ----
    unsigned long long addrA = 0ull;                                    // P2P: the partner row's address
    const int nranks = P2P ? p2p_nranks(a) : 1, me_rank = P2P ? p2p_me(a) : 0;
    const bool push = P2P && p2p_push(a);
    const int64_t shard_stride = 2 * (int64_t)a.hloc * ld;              // push: pos is (1 + nranks) blocks of a shard's rows -- block 0 this rank's, block 1 + q a copy of rank q's
    auto load_partner_rows = [&](int it) {
        const V2* oth;
        if constexpr (!P2P) {
            const uint32_t partner = (uint32_t)__builtin_amdgcn_ds_bpermute((gbase + it) * 4, (int)partnerA);
            oth = reinterpret_cast<const V2*>(posT + row_off(oth_row0 + partner));
        } else {
            const int src = (gbase + it) * 4;
            const unsigned lo = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)(unsigned)addrA);
            const unsigned hi = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)(unsigned)(addrA >> 32));
            const unsigned long long ad = ((unsigned long long)hi << 32) | lo;
            oth = reinterpret_cast<const V2*>(ad);
        }
        if constexpr (P2P) {
            // pull: the row lives in its owner's memory; push: in a local copy that the owner writes over the fabric -- either way read at
            // system scope, never from this XCD's L2
            if (nranks > 1) {
#pragma unroll
                for (int k = 0; k < K; ++k) xo[it][k] = load_row_sys(reinterpret_cast<const double2*>(&oth[ck[k]]));
                return;
            }
        }
#pragma unroll
        for (int k = 0; k < K; ++k) xo[it][k] = load_row(&oth[ck[k]]);
    };
    if constexpr (P2P) {
        // owner rank and row of the partner, resolved once per walker; the row address travels
        const uint32_t q = a.hloc_shift >= 0 ? partnerA >> a.hloc_shift : partnerA / a.hloc;
        const uint32_t r = partnerA - q * a.hloc;
        const double* base = a.peer_pos[0];
#pragma unroll
        for (int t = 1; t < 8; ++t) base = (q == (uint32_t)t) ? a.peer_pos[t] : base;
        if (push) base = (q == (uint32_t)me_rank) ? a.pos : a.pos + (int64_t)(1u + q) * shard_stride;   // local copy of rank q's shard
        addrA = (unsigned long long)(base + (oth_row0 + r) * ld);
        if (nranks > 1) {
            // every rank must have finished half-step `step - 1`: one polling wave per workgroup (the
            // flags sit in uncached fine-grained memory), the other waves wait at the barrier
            if ((threadIdx.x >> 6) == 0) wait_for_peers(a, step, lane);
            __syncthreads();
        }
    }
    constexpr int kFirst = ITER >= 2 ? ITER / 2 : ITER;                 // iterations whose loads precede the first logarithm
#pragma unroll
    for (int it = 0; it < kFirst; ++it) load_partner_rows(it);
    __builtin_amdgcn_sched_barrier(0);
    KMC_STAMP(2);                                       // the first partner-row loads are issued

    // ---- from here on the argument struct: one scalar round trip for all of it (have every field the kernel
    //      uses later requested by now, otherwise the compiler fetches some lazily: a round trip each) ----------
    asm volatile("" :: "s"(a.chain), "s"(a.chain_logp), "s"(a.chain_rows), "s"(a.chain_row0),
                 "s"(a.msum), "s"(a.msumsq), "s"(a.macc_stride), "s"(a.sched_inline.gen), "s"(a.sched_inline.slot),
                 "s"(a.sched_inline.flags), "s"(a.sched_inline.nbefore));
    // (the launch kind again, opaque to the optimiser: merged with the branch above it would pull the struct's first
    //  use -- and the wait for it -- in front of Philox)
    int eager_late = eager ? 1 : 0;
    asm volatile("" : "+v"(eager_late));
    eager_late = __builtin_amdgcn_readfirstlane(eager_late);
    SchedEntry sch = sch_t;
    if (eager_late != 0) sch = a.sched_inline;
    const bool count  = (sch.flags & kCount) != 0;
    const bool sample = (sch.flags & kSample) != 0;
    DrawConsts dc = a.dc;                                               // seed and nhalf from the front parameters
    dc.seed_lo = f.seed_lo; dc.seed_hi = f.seed_hi; dc.nhalf = f.nhalf;
    // Streaming moments are sojourn-weighted: a walker's value is credited, times the number of
    // samples it stood for, when it is replaced (and by flush_moments_vec at read-out).  Only waves
    // with an accepted move touch their accumulators -- at low acceptance (large ndim) almost none.
    const bool do_mom = count && a.msum != nullptr;
    // small rows: nearly every wave has an accepted move, so fetch its accumulator slots now and
    // keep that latency off the kernel's tail; large rows: fetch only when needed
    constexpr bool kPrefetchAcc = K <= 2 && L != 64;                   // L == 64: the moment ring instead (below)
    constexpr bool kMomRing = !kPrefetchAcc && !FoldT<L, K>::on && L == 64;          // HalfStepArgs::mring
    if constexpr (kMomRing) asm volatile("" :: "s"(a.mring), "s"(a.mring_w), "s"(a.mcnt), "s"(a.mswept), "s"(a.mring_depth));
    double2 accs[K], accq[K];
    double  acct[4] = {0.0, 0.0, 0.0, 0.0};
    // Large ensembles (ITER >= 4: the planner's choice from 16 384 waves on -- states that live in HBM) with several waves per workgroup: the
    // waves' sums are added up through LDS and ONE wave per workgroup reads and rewrites accumulator slots (its own; the others' stay as the
    // read-out and the flush kernel find them), in a fixed order -- a quarter (L = 32) or half (L = 8) of the accumulator bytes, which were 10 %
    // of a launch's HBM traffic at 524 288 x 128 (33.5 MB in + 33.5 MB out of 656 MB; profiles/traffic_hbm_512kx128.json).
    constexpr bool kWgFold = FoldT<L, K>::on && vec_tpb(L) > 64 && ITER >= 4;
    const bool acc_owner = !kWgFold || (threadIdx.x >> 6) == 0;
    if constexpr (FoldT<L, K>::on) {
        if (do_mom && acc_owner) {
#pragma unroll
            for (int r = 0; r < FoldT<L, K>::NVL; ++r) acct[r] = a.msum[((int64_t)(tid >> 6) * FoldT<L, K>::NVL + r) * 64 + lane];
        }
    } else if constexpr (kPrefetchAcc) {
        if (do_mom && g == 0) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int64_t idx = (int64_t)k * a.macc_stride + tid;
                accs[k] = reinterpret_cast<const double2*>(a.msum)[idx];
                accq[k] = reinterpret_cast<const double2*>(a.msumsq)[idx];
            }
        }
    }
    uint32_t ring_posted = 0u, ring_swept = 0u;
    if constexpr (kMomRing) {
        if (do_mom && a.mring != nullptr) { ring_posted = a.mcnt[tid >> 6]; ring_swept = a.mswept[tid >> 6]; }
    }
#if KMC_PROBE_PINS
    asm volatile("" :: "s"(count ? 1 : 0), "s"(a.dc.c0));
#endif
    KMC_STAMP(3);                                       // the argument struct has arrived (schedule entry, constants)
    Draw dr;
    dr.partner = partnerA; dr.z = e1.x; dr.t1 = e0.x; dr.lu = e0.y;
    double ua = 0.5;
    if (!fresh) {                                                       // the arithmetic of draw_finish, in two parts
        const double uz = ((double)bits.y + 0.5) * 0x1.0p-32;
        const double t  = fma(uz, dc.c1, dc.c0);
        dr.z = t * t;                                                   // :252
        const uint64_t kk = ((uint64_t)bits.z << 20) | (uint64_t)(bits.w >> 12);
        ua = ((double)kk + 0.5) * 0x1.0p-52;
        dr.t1 = dc.nm1 * log_pos_normal(dr.z);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int it = kFirst; it < ITER; ++it) load_partner_rows(it);
    __builtin_amdgcn_sched_barrier(0);
    if (!fresh) {
        dr.lu = log_pos_normal(ua);                                     // :260
        if constexpr (kRing) {
            if (ring_on && jq >= 1 && jq < Q && iA < nact) {            // park the walkers' next steps
                double2* slot = a.ring + ((int64_t)((a.ring_slot + jq) & 3) * a.ring_rows + rowA) * 2;
                store_wt(&slot[0], make_double2(dr.t1, dr.lu));
                store_wt(&slot[1], make_double2(dr.z, __hiloint2double((int)(uint32_t)(step + 2ull * (uint64_t)jq), (int)dr.partner)));
            }
        }
    }
#if KMC_PROBE_PINS
    asm volatile("" :: "v"(dr.lu), "v"(dr.t1));
#endif
    KMC_STAMP(4);                                       // both logarithms done, every partner-row load issued
    double zB[ITER];
#pragma unroll
    for (int it = 0; it < ITER; ++it) zB[it] = bperm_f64((gbase + it) * 4, dr.z);

    // ---- stretch move + log-pdf; xo becomes the proposal ------------------------------------
    double myp1 = 0.0;
    constexpr int kRowND = RowEvalTrait<Dens>::n;                       // > 0: a function body over the whole proposal (see below)
    double blob1[BlobTrait<Dens>::n > 0 ? BlobTrait<Dens>::n : 1];      // ... and the blob it returned (blob1 of src/samplers.jl:257), scalar layout
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int k = 0; k < K; ++k) {                                   // :255
            xo[it][k].x = as_stored<T>(fma(zB[it], xc[it][k].x - xo[it][k].x, xo[it][k].x));
            xo[it][k].y = as_stored<T>(fma(zB[it], xc[it][k].y - xo[it][k].y, xo[it][k].y));
        }
        if constexpr (MultiSumTrait<Dens>::n > 0) {                    // a function body feeding several sums over the elements
            double S[MultiSumTrait<Dens>::n];
            Dens::template frag_partial_n<L, K>(xo[it], j, ndim, a.dp, S);
#pragma unroll
            for (int q = 0; q < MultiSumTrait<Dens>::n; ++q) S[q] = group_sum<L>(S[q]);
            const double p1 = Dens::finish_n(S, a.dp);                   // :257
            myp1 = (j == it) ? p1 : myp1;
        } else if constexpr (kRowND == 0) {
            const double S  = group_sum<L>(Dens::template frag_partial<L, K>(xo[it], j, ndim, a.dp));
            const double p1 = Dens::finish(S, a.dp);                     // :257
            myp1 = (j == it) ? p1 : myp1;                               // row -> scalar, no traffic
        }
    }
    if constexpr (kRowND > 0) {
        // A caller's function body over the whole proposal (BodyDensity): rows are loaded, moved and stored lane-striped like
        // everybody's, and only the evaluation is per walker -- the wave's W proposals go through a per-wave LDS tile (row stride
        // 2 L K + 2 doubles: 16-byte aligned chunks) and the scalar-layout lane of each walker, the one that holds its draws and
        // runs its accept test, calls the body once on its row there (src/samplers.jl:257), elements in index order: the same
        // value, bit for bit, as the one-walker-per-lane kernels give.
        extern __shared__ __attribute__((aligned(16))) double vec_rows[];
        constexpr int TS = 2 * L * K + 2;
        double* tile = vec_rows + (size_t)(threadIdx.x >> 6) * (size_t)(W * TS);
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int k = 0; k < K; ++k)
                *reinterpret_cast<double2*>(&tile[(it * G + g) * TS + 2 * (k * L + j)]) = xo[it][k];     // (a tile row is 2 L K + 2 wide: chunks past a ragged row's end land behind it, unread)
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if constexpr (BlobTrait<Dens>::n > 0) {
            if (useA) {
#pragma unroll 1
                for (int i = 0; i < BlobTrait<Dens>::n; ++i) blob1[i] = 0.0;
                myp1 = Dens::eval_row(&tile[(js * G + g) * TS], ndim, a.dp, blob1);
            }
        } else {
            if (useA) myp1 = Dens::eval_row(&tile[(js * G + g) * TS], ndim, a.dp);  // (inlined: the body's x[i] become LDS reads of this lane's row)
        }
    }

#if KMC_PROBE_PINS
    asm volatile("" :: "v"(myp1));
#endif
    KMC_STAMP(5);                                       // both rows have arrived, the proposal's log-pdf is reduced
    // ---- accept test in the scalar layout ---------------------------------------------------
    const bool acc = validA && accept_test(dr, myp1, p0);               // :260
    const unsigned long long accmask = __ballot(acc);
    if (acc) {
        store_wt(&logp_p[rowA], myp1);                                  // :262
        if (count) store_wt(&naccept_p[rowA], na + 1u);                 // :265
        if (do_mom) store_wt(&klast_p[rowA], sch.nbefore);
    }
    if constexpr (BlobTrait<Dens>::n > 0) {
        // the blob of the walker's CURRENT position follows it (blob0s[nc] = blob1 on accept, :264) and is stored with every sample (:270)
        constexpr int NB = BlobTrait<Dens>::n;
        const bool keep = validA && sample && a.chain_blob != nullptr;
        if (acc || keep) {
            double* cur = a.blob + rowA * NB;
            double* dst = a.chain_blob + (sch.slot * a.chain_rows + a.chain_row0 + iA) * NB;
#pragma unroll 1
            for (int i = 0; i < NB; ++i) {
                const double b = acc ? blob1[i] : cur[i];
                if (acc) cur[i] = b;
                if (keep) dst[i] = b;
            }
        }
    }
    const uint32_t wA = (acc && do_mom) ? sch.nbefore - kl : 0u;        // samples the replaced value stood for
    const bool any_w = __ballot(wA != 0u) != 0ull;
    if (sample && a.chain_logp != nullptr && validA)                    // :271
        store_wt(&a.chain_logp[sch.slot * a.chain_rows + a.chain_row0 + iA], acc ? myp1 : p0);

    KMC_STAMP(6);                                       // accept test done, per-walker scalars stored
    // ---- row layout again: store accepted proposals, samples, moments -----------------------
    double2 ms[K], mq[K];
#pragma unroll
    for (int k = 0; k < K; ++k) { ms[k] = make_double2(0.0, 0.0); mq[k] = make_double2(0.0, 0.0); }
    // room for one entry per walker of the wave? (wave-uniform; kMomRing geometries have one group per wave)
    const bool use_ring = kMomRing && a.mring != nullptr && ring_posted - ring_swept + (uint32_t)ITER <= (uint32_t)a.mring_depth;
    uint32_t ring_new = 0u;
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const bool accB = ((accmask >> (gbase + it)) & 1ull) != 0;
        if (accB) {                                                     // :261
            V2* own = reinterpret_cast<V2*>(posT + row_off(own_row0 + w0 + it * G + g));
#pragma unroll
            for (int k = 0; k < K; ++k) store_row(&own[ck[k]], xo[it][k]);
            if constexpr (P2P) {
                if (push) {                                    // ... and into this rank's copy on every peer (write-through, over the fabric)
                    const int64_t off = (int64_t)(1 + me_rank) * shard_stride + (own_row0 + w0 + it * G + g) * ld;
                    for (int r = 0; r < nranks; ++r) {
                        if (r == me_rank) continue;
                        double2* rem = reinterpret_cast<double2*>(a.peer_pos[r] + off);
#pragma unroll
                        for (int k = 0; k < K; ++k) if (cv[k]) store_wt(&rem[k * L + j], xo[it][k]);      // (over the fabric: a folded tail chunk would be sent twice)
                    }
                }
            }
        }
        if (any_w) {
            const double wB = (double)(uint32_t)__builtin_amdgcn_ds_bpermute((gbase + it) * 4, (int)wA);
            if (use_ring) {
                if (wB != 0.0) {                                          // wave-uniform (L == 64: one group)
                    const int64_t e = (int64_t)(tid >> 6) * a.mring_depth + (int64_t)((ring_posted + ring_new) % (uint32_t)a.mring_depth);
                    double2* slot = a.mring + e * K * 64 + lane;
#pragma unroll
                    for (int k = 0; k < K; ++k) slot[k * 64] = xc[it][k];
                    if (lane == 0) a.mring_w[e] = wB;
                    ring_new += 1u;
                }
            } else {
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    ms[k].x += xc[it][k].x * wB; ms[k].y += xc[it][k].y * wB;
                    mq[k].x += (xc[it][k].x * xc[it][k].x) * wB; mq[k].y += (xc[it][k].y * xc[it][k].y) * wB;
                }
            }
        }
        if (sample && a.chain != nullptr && validB[it]) {               // :268-269
            V2* dst = reinterpret_cast<V2*>(reinterpret_cast<T*>(a.chain) + (sch.slot * a.chain_rows + a.chain_row0 + w0 + it * G + g) * ld);
#pragma unroll
            for (int k = 0; k < K; ++k) store_row(&dst[ck[k]], sel2(accB, xo[it][k], xc[it][k]));
        }
    }
    if constexpr (kWgFold) {
        if (do_mom) {                                                   // (uniform over the launch: every wave of the workgroup arrives)
            constexpr int NVL = FoldT<L, K>::NVL, NWV = vec_tpb(L) / 64;
            __shared__ double wg_fold[NWV - 1][NVL][64];
            const int wv = (int)(threadIdx.x >> 6);
            double v[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            if (any_w) fold_scatter<L, K>(lane, ms, mq, v);
            if (wv != 0) {
#pragma unroll
                for (int r = 0; r < NVL; ++r) wg_fold[wv - 1][r][lane] = v[r];
            }
            lds_barrier();
            if (wv == 0) {
                double* slot = a.msum + ((int64_t)(tid >> 6) * NVL) * 64 + lane;
#pragma unroll
                for (int r = 0; r < NVL; ++r) {
                    double t = v[r];
#pragma unroll
                    for (int w = 0; w < NWV - 1; ++w) t += wg_fold[w][r][lane];
                    store_wt(&slot[r * 64], acct[r] + t);
                }
            }
        }
    } else if (any_w) {
        if (use_ring) {
            if (lane == 0) a.mcnt[tid >> 6] = ring_posted + ring_new;
        } else {
            if constexpr (kPrefetchAcc) accumulate_wave<L, K, true>(a.msum, a.msumsq, a.macc_stride, tid, g, ms, mq, accs, accq, acct);
            else accumulate_wave<L, K, false>(a.msum, a.msumsq, a.macc_stride, tid, g, ms, mq, accs, accq, acct);
        }
    }
    KMC_STAMP(7);                                       // the last store is issued
#ifdef KMC_PROBE
    {
        unsigned long long st[8];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        KMC_STAMP_READ(st[0], 80, 81); KMC_STAMP_READ(st[1], 82, 83); KMC_STAMP_READ(st[2], 84, 85); KMC_STAMP_READ(st[3], 86, 87);
        KMC_STAMP_READ(st[4], 88, 89); KMC_STAMP_READ(st[5], 90, 91); KMC_STAMP_READ(st[6], 92, 93); KMC_STAMP_READ(st[7], 94, 95);
        if (lane == 0 && (tid >> 6) < 8192) for (int q = 0; q < 8; q += (KMC_PROBE == 2 ? 7 : 1)) g_probe[half][tid >> 6][q] = st[q];
    }
#endif
}

template <class Dens, int L, int K, int ITER, bool P2P, bool RAGGED, class T = double>
__global__ __launch_bounds__(vec_tpb(L)) void half_step_vec(KMC_FRONT_PARAMS, const HalfStepArgs a)
{
    half_step_vec_body<Dens, L, K, ITER, P2P, RAGGED, T>(KMC_FRONT_PACK, a);
}

// Moment read-out: credit every walker's current value with the samples it has stood for since it
// was last credited (S = samples taken so far), same lane mapping as half_step_vec.
struct FlushArgs {
    const double* pos;
    uint32_t*     klast;
    double*       msum;
    double*       msumsq;
    int64_t       macc_stride;
    int64_t       row0;      // first row of this launch in pos / klast
    int32_t       n_active;
    uint32_t      nsamp;     // S
    int32_t       ld;        // row stride in doubles
};

template <int L, int K, int ITER, class T = double>
__global__ __launch_bounds__(vec_tpb(L)) void flush_moments_vec(const FlushArgs a)
{
    using V2 = typename RowOf<T>::V2;
    constexpr int G = 64 / L;
    constexpr int W = G * ITER;
    const int64_t ld = a.ld;
    const int tid  = blockIdx.x * vec_tpb(L) + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int j    = lane & (L - 1);
    const int g    = lane / L;
    const int w0   = (tid >> 6) * W;
    double2 ms[K], mq[K];
#pragma unroll
    for (int k = 0; k < K; ++k) { ms[k] = make_double2(0.0, 0.0); mq[k] = make_double2(0.0, 0.0); }
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const int i = w0 + it * G + g;
        if (i < a.n_active) {
            const int64_t row = a.row0 + i;
            const double w = (double)(a.nsamp - a.klast[row]);
            const V2* x = reinterpret_cast<const V2*>(reinterpret_cast<const T*>(a.pos) + row * ld);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const double2 v = 2 * (k * L + j) < (int)ld ? load_row(&x[k * L + j]) : make_double2(0.0, 0.0);
                ms[k].x += v.x * w; ms[k].y += v.y * w;
                mq[k].x += (v.x * v.x) * w; mq[k].y += (v.y * v.y) * w;
            }
        }
    }
    __syncthreads();                       // every lane has read klast before anyone rewrites it
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const int i = w0 + it * G + g;
        if (i < a.n_active && j == 0) a.klast[a.row0 + i] = a.nsamp;
    }
    const double none[4] = {0.0, 0.0, 0.0, 0.0};
    accumulate_wave<L, K, false>(a.msum, a.msumsq, a.macc_stride, tid, g, ms, mq, ms, mq, none);
}

// ------------------------------------------------------------------------------------------
// Generic kernel: one walker per lane, any ndim.
// ------------------------------------------------------------------------------------------
template <class Dens, bool P2P, class T = double>
__device__ __forceinline__ void half_step_generic_body(const HalfStepFront& f, const HalfStepArgs& a)
{
    static_assert(!P2P || sizeof(T) == 8, "the peer-to-peer kernels keep double rows");
    const int tid = blockIdx.x * 256 + threadIdx.x;
    const SchedEntry sch = schedule_of(f, a);
    const uint64_t step = 2ull * (uint64_t)sch.gen + (uint64_t)a.half;      // (eager: sched_inline.gen)
    if constexpr (P2P) { if (p2p_nranks(a) > 1) wait_for_peers(a, step, (int)(threadIdx.x & 63)); }   // whole waves, before any exit
    if (tid >= a.n_active) return;
    const int ndim = a.ndim;
    const bool count  = (sch.flags & kCount) != 0;
    const bool sample = (sch.flags & kSample) != 0;
    const int64_t gw = a.own_row0 + tid;                                // row in pos / index in logp, naccept
    const Draw dr = draw_step(a.dc, step, (uint64_t)(a.gw0 + tid));
    const int64_t ld = a.ld;
    T* own = reinterpret_cast<T*>(a.pos) + gw * ld;
    const T* oth;
    if constexpr (!P2P) {
        oth = reinterpret_cast<const T*>(a.pos) + (a.oth_row0 + dr.partner) * ld;
    } else {
        const uint32_t q = a.hloc_shift >= 0 ? dr.partner >> a.hloc_shift : dr.partner / a.hloc;
        const uint32_t r = dr.partner - q * a.hloc;
        const double* base = a.peer_pos[0];
#pragma unroll
        for (int t = 1; t < 8; ++t) base = (q == (uint32_t)t) ? a.peer_pos[t] : base;
        oth = reinterpret_cast<const T*>(base) + (a.oth_row0 + r) * ld;
    }
    const double p0 = a.logp[gw];
    // partner element d: P2P rows live in their owner's memory and are read at system scope (see load_row_sys)
    auto oth_at = [&](int d) -> double {
        if constexpr (P2P) {
            if (p2p_nranks(a) > 1)
                return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(oth) + d,
                                                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
        }
        return (double)oth[d];
    };

    constexpr bool kHost = HostEvalTrait<Dens>::value;
    if constexpr (kHost) {
        if (a.prop_out != nullptr) {                                    // PROPOSE pass
            for (int d = 0; d < ndim; ++d) { const double o = oth_at(d); a.prop_out[(int64_t)tid * a.prop_ld + d] = fma(dr.z, (double)own[d] - o, o); }
            return;
        }
    }
    typename Dens::Seq q;
    Dens::seq_init(q);
    for (int d = 0; d < ndim; ++d) {
        const double o = oth_at(d);
        const double y = as_stored<T>(fma(dr.z, (double)own[d] - o, o));   // :255
        Dens::seq_add(q, y, d, a.dp);
    }
    double p1 = Dens::seq_finish(q, ndim, a.dp);                         // :257
    if constexpr (kHost) p1 = a.p1_in[tid];
    const bool acc = accept_test(dr, p1, p0);                           // :260
    if constexpr (kHost) { if (a.acc_out != nullptr) a.acc_out[tid] = acc ? 1 : 0; }

    const bool do_mom = sample && a.msum != nullptr;
    const bool do_chain = sample && a.chain != nullptr;
    const int64_t row = sch.slot * a.chain_rows + a.chain_row0 + tid;
    if (acc || do_mom || do_chain) {
        for (int d = 0; d < ndim; ++d) {
            const double xcd = (double)own[d];
            const double o = acc ? oth_at(d) : 0.0;
            const double cur = acc ? as_stored<T>(fma(dr.z, xcd - o, o)) : xcd;
            if (acc) own[d] = (T)cur;                                   // :261
            if (do_chain) reinterpret_cast<T*>(a.chain)[row * ld + d] = (T)cur;   // :269
            if (do_mom) {
                const int64_t idx = (int64_t)d * a.macc_stride + tid;
                a.msum[idx] += cur;
                a.msumsq[idx] += cur * cur;
            }
        }
    }
    if (acc) {
        a.logp[gw] = p1;                                                // :262
        if (count) a.naccept[gw] += 1u;                                 // :265
    }
    if (sample && a.chain_logp != nullptr) a.chain_logp[row] = acc ? p1 : p0;   // :271
    if constexpr (BlobTrait<Dens>::n > 0) {
        constexpr int NB = BlobTrait<Dens>::n;
        double* cur = a.blob + gw * NB;
        const bool keep = sample && a.chain_blob != nullptr;
        if (acc || keep) {
#pragma unroll 1
            for (int i = 0; i < NB; ++i) {
                const double b = acc ? q.blob[i] : cur[i];
                if (acc) cur[i] = b;                                    // :264
                if (keep) a.chain_blob[row * NB + i] = b;               // :270
            }
        }
    }
}

template <class Dens, bool P2P, class T = double>
__global__ __launch_bounds__(256) void half_step_generic(KMC_FRONT_PARAMS, const HalfStepArgs a)
{
    half_step_generic_body<Dens, P2P, T>(KMC_FRONT_PACK, a);
}

// ------------------------------------------------------------------------------------------
// Staged kernel for body densities (BodyDensity: the whole proposal per lane; ndim = ND known when the runtime compiler
// instantiates it).  One walker per lane like the generic kernel, but no lane ever walks its own row in memory (64 lanes
// x one 8-byte element of 64 different rows = 64 cache lines per load instruction): a wave's 64 own rows and its 64
// partner rows are fetched COOPERATIVELY -- 16 B per lane, consecutive lanes on consecutive chunks of a row, the own
// block one contiguous stream -- through a per-wave LDS tile (64 rows x TD doubles, padded) and handed to the lanes as
// private arrays, which the compiler keeps in registers for short rows.  Accepted rows go back the same way.  Same
// draws, same arithmetic and element order as half_step_generic (results identical); double rows, one GPU.
// ------------------------------------------------------------------------------------------
constexpr int kBodyVecMaxDim = 1024;                                     // a body evaluated per walker inside the vector kernel (every row length the vector kernel has)
// LDS of the vector kernel when it evaluates a body per walker: one tile of W rows x (2 L K + 2) doubles per wave
__host__ __device__ constexpr size_t body_vec_lds_bytes(int L, int K, int iter) { return (size_t)(vec_tpb(L) / 64) * (size_t)((64 / L) * iter) * (size_t)(2 * L * K + 2) * sizeof(double); }
constexpr int kStagedTPB = 128;                                          // two waves per workgroup, one LDS tile each
constexpr int kStagedMaxDim = 256;                                       // up to 64: two rows + the proposal per lane in registers; beyond: in scratch (spilled)
__host__ __device__ constexpr int staged_tile_doubles(int nd) { return nd + (nd & 1) < 32 ? nd + (nd & 1) : 32; }
// the staged kernel's moment accumulators are rows [wave][ld] (true) or per-lane columns [d][tid] like half_step_generic's (false)
__host__ __device__ constexpr bool staged_tile_moments(int nd) { return 64 % (staged_tile_doubles(nd) / 2) == 0; }
__host__ __device__ constexpr size_t staged_lds_bytes(int nd) { return (size_t)(kStagedTPB / 64) * 64 * (size_t)(staged_tile_doubles(nd) + 1) * sizeof(double); }

template <class Dens, int ND>
__device__ __forceinline__ void half_step_staged_body(const HalfStepFront& f, const HalfStepArgs& a)
{
    constexpr int LD = ND + (ND & 1);                 // row stride in doubles
    constexpr int TD = staged_tile_doubles(ND);       // tile width (doubles); TD / 2 16-byte chunks per row piece
    constexpr int CPR = TD / 2;                       // chunks per row piece
    constexpr int NT = (LD + TD - 1) / TD;            // tiles per row
    extern __shared__ double staged_lds[];
    const int lane = threadIdx.x & 63;
    double* tile = staged_lds + (size_t)(threadIdx.x >> 6) * 64 * (TD + 1);
    const int tid = blockIdx.x * kStagedTPB + threadIdx.x;
    const int w0 = (tid >> 6) * 64;                   // first active index of this wave
    const SchedEntry sch = schedule_of(f, a);
    const uint64_t step = 2ull * (uint64_t)sch.gen + (uint64_t)a.half;
    const int nact = a.n_active;
    if (w0 >= nact) return;                           // (whole waves only)
    const bool valid = tid < nact;
    const int  ic = valid ? tid : nact - 1;
    const bool count  = (sch.flags & kCount) != 0;
    const bool sample = (sch.flags & kSample) != 0;
    const int64_t gw = a.own_row0 + ic;
    const int rows_here = nact - w0 < 64 ? nact - w0 : 64;

    // Row r of a tile <- piece [c0, c0 + TD) of a global row; lane q handles chunk q % CPR of tile row q / CPR (a chunk beyond
    // the row's end re-reads the last one and lands in tile columns nobody reads).  ALL the loads of the wave go out before
    // anything waits: the own rows at once (they do not depend on the draw), the partner rows as soon as Philox has given the
    // partner indices, the two logarithms of the draw while they fly -- then the pieces are transposed through LDS.  (Measured
    // at 65 536 x 32: worth 2 % against loading and transposing piece by piece -- a wave here is ~2000 instructions for its 64
    // walkers with one wave per SIMD, so it is instruction issue, not the memory round trips, that bounds this kernel.)
    // Beyond 64 dimensions the pieces go one at a time (own and partner piece of a tile together): the staging registers of a whole
    // row pair would spill, and the rows themselves already live in scratch there (the compiler's spill of xc / xo -- lane-
    // interleaved private memory, still far better than the generic kernel's element-wise loads of 64 different rows).
    constexpr bool kAllAtOnce = ND <= 64;
    // (addresses: a wave-uniform base + a 32-bit lane offset wherever the row is one of the wave's own 64 -- the 64-bit address
    //  arithmetic of every load and store was a quarter of this kernel's instructions)
    const int w0u = __builtin_amdgcn_readfirstlane(w0);
    double* const own_base = a.pos + (a.own_row0 + (int64_t)w0u) * LD;
    auto issue_tile = [&](auto rowptr, int t, double2 (&v)[CPR]) {
#pragma unroll
        for (int i = 0; i < CPR; ++i) {
            const int q = i * 64 + lane, r = q / CPR, ch = q - r * CPR;
            const int col = t * TD + 2 * ch < LD ? t * TD + 2 * ch : LD - 2;
            v[i] = *reinterpret_cast<const double2*>(rowptr(r) + (unsigned)col);
        }
    };
    double xc[LD], xo[LD];
    auto land_tile = [&](const double2 (&v)[CPR], int t, double (&dst)[LD]) {
        const int c0 = t * TD;
#pragma unroll
        for (int i = 0; i < CPR; ++i) {
            const int q = i * 64 + lane, r = q / CPR, ch = q - r * CPR;
            tile[r * (TD + 1) + 2 * ch] = v[i].x;
            tile[r * (TD + 1) + 2 * ch + 1] = v[i].y;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int e = 0; e < TD; ++e)
            if (c0 + e < LD) dst[c0 + e] = tile[lane * (TD + 1) + e];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    auto own_row = [&](int r) { return own_base + (unsigned)(r < rows_here ? r : rows_here - 1) * (unsigned)LD; };
    Draw dr;
    double p0;
    if constexpr (kAllAtOnce) {
        double2 vown[NT][CPR], voth[NT][CPR];
#pragma unroll
        for (int t = 0; t < NT; ++t) issue_tile(own_row, t, vown[t]);
        p0 = a.logp[gw];
        const U4 bits = draw_bits(a.dc, step, (uint64_t)(a.gw0 + ic));
        const uint32_t partner = draw_partner(a.dc, bits);              // :250
        auto oth_row = [&](int r) { return a.pos + (a.oth_row0 + (int64_t)(uint32_t)__shfl((int)partner, r)) * (int64_t)LD; };
#pragma unroll
        for (int t = 0; t < NT; ++t) issue_tile(oth_row, t, voth[t]);
        dr = draw_finish(a.dc, bits);                                   // :252, and the accept test's logarithms
#pragma unroll
        for (int t = 0; t < NT; ++t) land_tile(vown[t], t, xc);
#pragma unroll
        for (int t = 0; t < NT; ++t) land_tile(voth[t], t, xo);
    } else {
        p0 = a.logp[gw];
        const U4 bits = draw_bits(a.dc, step, (uint64_t)(a.gw0 + ic));
        const uint32_t partner = draw_partner(a.dc, bits);              // :250
        auto oth_row = [&](int r) { return a.pos + (a.oth_row0 + (int64_t)(uint32_t)__shfl((int)partner, r)) * (int64_t)LD; };
        dr = draw_finish(a.dc, bits);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            double2 vown[CPR], voth[CPR];
            issue_tile(own_row, t, vown);
            issue_tile(oth_row, t, voth);
            land_tile(vown, t, xc);
            land_tile(voth, t, xo);
        }
    }

    typename Dens::Seq q;
    Dens::seq_init(q);
#pragma unroll
    for (int d = 0; d < ND; ++d) Dens::seq_add(q, fma(dr.z, xc[d] - xo[d], xo[d]), d, a.dp);   // :255
    const double p1 = Dens::seq_finish(q, ND, a.dp);                    // :257
    const bool acc = valid && accept_test(dr, p1, p0);                  // :260

    // Streaming moments.  When a lane keeps the same chunk of the row through the cooperative write-out below (CPR divides 64),
    // the sums are taken there: a lane adds the chunk's two elements of its 64 / CPR rows, the lanes that share a chunk fold, and
    // the wave updates ONE contiguous accumulator row [wave][LD] (staged_tile_moments(): the read-out knows) -- instead of every
    // lane reading and writing all ND of its own, which doubled the kernel's traffic and was a third of its instructions.
    constexpr bool kTileMoments = staged_tile_moments(ND);
    const bool mom_wave = sample && a.msum != nullptr;                  // (uniform)
    const bool do_mom = !kTileMoments && valid && mom_wave;
    const bool do_chain = sample && a.chain != nullptr;
    const int64_t crow = sch.slot * a.chain_rows + a.chain_row0;        // chain row of this launch's active walker 0
    double* const chain_base = a.chain + (crow + (int64_t)w0u) * LD;
    if (do_mom) {                                                       // per-lane accumulators, eight dimensions' loads in flight
#pragma unroll
        for (int d0 = 0; d0 < ND; d0 += 8) {
            double s1[8], s2[8];
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (d0 + k < ND) { s1[k] = a.msum[(int64_t)(d0 + k) * a.macc_stride + tid]; s2[k] = a.msumsq[(int64_t)(d0 + k) * a.macc_stride + tid]; }
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (d0 + k < ND) {
                    const int d = d0 + k;
                    const double cur = acc ? fma(dr.z, xc[d] - xo[d], xo[d]) : xc[d];
                    a.msum[(int64_t)d * a.macc_stride + tid] = s1[k] + cur;
                    a.msumsq[(int64_t)d * a.macc_stride + tid] = s2[k] + cur * cur;
                }
        }
    }
    // rows out: through the tile again, so that the stores are 16 B per lane on consecutive chunks (:261, :269)
    const unsigned long long accmask = __ballot(acc);
    if (accmask != 0ull || do_chain || (kTileMoments && mom_wave)) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int c0 = t * TD;
#pragma unroll
            for (int e = 0; e < TD; ++e)
                if (c0 + e < LD)                                        // (the pad column of an odd ndim stays zero)
                    tile[lane * (TD + 1) + e] = c0 + e >= ND ? 0.0 : (acc ? fma(dr.z, xc[c0 + e] - xo[c0 + e], xo[c0 + e]) : xc[c0 + e]);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            double2 ms = make_double2(0.0, 0.0), mq = make_double2(0.0, 0.0);
#pragma unroll
            for (int q0 = 0; q0 < 64 * CPR; q0 += 64) {
                const int qq = q0 + lane, r = qq / CPR, ch = qq - r * CPR;
                if (c0 + 2 * ch < LD && r < rows_here) {
                    const double2 v = make_double2(tile[r * (TD + 1) + 2 * ch], tile[r * (TD + 1) + 2 * ch + 1]);
                    const unsigned off = (unsigned)r * (unsigned)LD + (unsigned)(c0 + 2 * ch);
                    if ((accmask >> r) & 1ull) *reinterpret_cast<double2*>(own_base + off) = v;
                    if (do_chain) *reinterpret_cast<double2*>(chain_base + off) = v;
                    if constexpr (kTileMoments) { ms.x += v.x; ms.y += v.y; mq.x += v.x * v.x; mq.y += v.y * v.y; }
                }
            }
            if constexpr (kTileMoments) {
                if (mom_wave) {                                         // lane l holds chunk l % CPR in every pass above
#pragma unroll
                    for (int off = CPR; off < 64; off <<= 1) {
                        ms.x += __shfl_xor(ms.x, off); ms.y += __shfl_xor(ms.y, off);
                        mq.x += __shfl_xor(mq.x, off); mq.y += __shfl_xor(mq.y, off);
                    }
                    if (lane < CPR && c0 + 2 * lane < LD) {             // (the pad column of an odd ndim sums zeros)
                        double2* s1 = reinterpret_cast<double2*>(a.msum + (int64_t)(tid >> 6) * LD + c0) + lane;
                        double2* s2 = reinterpret_cast<double2*>(a.msumsq + (int64_t)(tid >> 6) * LD + c0) + lane;
                        const double2 o1 = *s1, o2 = *s2;
                        *s1 = make_double2(o1.x + ms.x, o1.y + ms.y);
                        *s2 = make_double2(o2.x + mq.x, o2.y + mq.y);
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (acc) {
        a.logp[gw] = p1;                                                // :262
        if (count) a.naccept[gw] += 1u;                                 // :265
    }
    if (valid && sample && a.chain_logp != nullptr) a.chain_logp[crow + tid] = acc ? p1 : p0;   // :271
    if constexpr (BlobTrait<Dens>::n > 0) {
        constexpr int NB = BlobTrait<Dens>::n;
        double* cur = a.blob + gw * NB;
        const bool keep = valid && sample && a.chain_blob != nullptr;
        if (acc || keep) {
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const double b = acc ? q.blob[i] : cur[i];
                if (acc) cur[i] = b;                                    // :264
                if (keep) a.chain_blob[(crow + tid) * NB + i] = b;      // :270
            }
        }
    }
}

// Initial log-pdfs, src/samplers.jl:209.
struct LogpdfArgs {
    const double* pos;
    double*       logp;
    int64_t       nrows;
    int32_t       ndim;
    int32_t       ld;
    DensityParams dp;
    double*       blob;       // body densities with blobs: [nrows][NB] (or nullptr: log-pdfs only)
};

template <class Dens>
__device__ __forceinline__ void logpdf_rows_body(const LogpdfArgs& a)
{
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= a.nrows) return;
    typename Dens::Seq q;
    Dens::seq_init(q);
    for (int d = 0; d < a.ndim; ++d) Dens::seq_add(q, a.pos[r * a.ld + d], d, a.dp);
    a.logp[r] = Dens::seq_finish(q, a.ndim, a.dp);
    if constexpr (BlobTrait<Dens>::n > 0) {
        if (a.blob != nullptr) {
#pragma unroll 1
            for (int i = 0; i < BlobTrait<Dens>::n; ++i) a.blob[r * BlobTrait<Dens>::n + i] = q.blob[i];   // :209-210
        }
    }
}

template <class Dens>
__global__ __launch_bounds__(256) void logpdf_rows(const LogpdfArgs a)
{
    logpdf_rows_body<Dens>(a);
}

// P2P: after a half-step kernel has drained (kernel boundary = its rows are in memory), tell every
// rank -- including this one -- that this rank has completed half-step `step` (flag = step + 1).
struct SignalArgs {
    unsigned long long* peer_flags[8];   // peer_flags[r] = rank r's flags array
    int32_t             nranks;
    int32_t             me;
    const SchedEntry*   sched_table;
    SchedEntry          sched_inline;
    int32_t             sched_index;
    int32_t             half;
};
// Device-side make_theta0s (reference src/samplers.jl:311-349, the intended behaviour): walker w gets
// theta0 + N(0, diag(r^2)), redrawn while its log-pdf is -inf -- up to `ntries` draws per ball size,
// the ball shrinking by the reference's cumulative factors 1, 1/2, 1/8, 1/64, ... per halving step
// (:326), reset for every walker.  Normals: Box-Muller on Philox4x32-10 keyed by
// (seed ^ "BALL", try index, walker, pair of dimensions).  fail[0] counts walkers left without an
// admissible point.
struct InitBallArgs {
    double*       pos;        // [nrows][ld]
    double*       logp;       // [nrows]
    const double* theta0;     // [ndim]
    const double* radius;     // [ndim]
    int64_t       nrows;
    int64_t       row_walker0;   // global walker index of row 0 (P2P shards pass their slices separately)
    int32_t       ndim, ld;
    int32_t       halving_steps, ntries;
    uint32_t      seed_lo, seed_hi;
    DensityParams dp;
    unsigned long long* fail;
    double*       blob;       // body densities with blobs: [nrows][NB] of the admitted points, or nullptr
};

template <class Dens>
__device__ __forceinline__ void init_ball_body(const InitBallArgs& a)
{
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= a.nrows) return;
    const uint64_t walker = (uint64_t)(a.row_walker0 + r);
    double* x = a.pos + r * a.ld;
    double shrink = 1.0;
    uint32_t attempt = 0;
    for (int k = 1; k <= a.halving_steps; ++k) {
        shrink *= ldexp(1.0, -(k - 1));                                        // :326  1/2^(k-1), exact for any k
        for (int t = 0; t < a.ntries; ++t, ++attempt) {
            typename Dens::Seq q;
            Dens::seq_init(q);
            for (int d = 0; d < a.ndim; d += 2) {
                const U4 w = philox4x32_10(attempt, (uint32_t)(d >> 1), (uint32_t)walker, (uint32_t)(walker >> 32),
                                           a.seed_lo ^ 0x42414c4cu, a.seed_hi);
                const double u1 = ((double)(((uint64_t)w.x << 20) | (w.y >> 12)) + 0.5) * 0x1.0p-52;
                const double u2 = ((double)w.z + 0.5) * 0x1.0p-32;
                const double rad = sqrt(-2.0 * log(u1));
                double sn, cs;
                sincos(6.283185307179586476925286766559 * u2, &sn, &cs);
                const double n0 = rad * cs, n1 = rad * sn;
                x[d] = a.theta0[d] + n0 * (a.radius[d] * shrink);             // :328-332
                Dens::seq_add(q, x[d], d, a.dp);
                if (d + 1 < a.ndim) {
                    x[d + 1] = a.theta0[d + 1] + n1 * (a.radius[d + 1] * shrink);
                    Dens::seq_add(q, x[d + 1], d + 1, a.dp);
                }
            }
            const double p = Dens::seq_finish(q, a.ndim, a.dp);
            if (p > -INFINITY && p == p) {                                    // :338
                a.logp[r] = p;
                if constexpr (BlobTrait<Dens>::n > 0) {
                    if (a.blob != nullptr) {
#pragma unroll 1
                        for (int i = 0; i < BlobTrait<Dens>::n; ++i) a.blob[r * BlobTrait<Dens>::n + i] = q.blob[i];
                    }
                }
                return;
            }
        }
    }
    a.logp[r] = -INFINITY;
    atomicAdd(a.fail, 1ull);
}

template <class Dens>
__global__ __launch_bounds__(256) void init_ball(const InitBallArgs a)
{
    init_ball_body<Dens>(a);
}

// Folds the moment ring's posted entries into the accumulators, oldest first (one 64-lane workgroup per wave of the
// half-step grid; same slot mapping as accumulate_wave's plain form: slot k of global thread t at [k * stride + t]).
struct SweepArgs {
    const double2* ring;
    const double*  ring_w;
    const uint32_t* cnt;
    uint32_t*      swept;
    double*        msum;
    double*        msumsq;
    int64_t        macc_stride;
    int32_t        K, depth;
};

// Non-template kernels of the host driver: each group is defined once, in the translation unit that launches it.
#ifdef KMC_DEFINE_LAUNCH_KERNELS   // kmc_launch.hip
// one 64-lane workgroup per (wave of the half-step grid, chunk k): sum += x w, sumsq += x^2 w over the wave's posted rows
__global__ __launch_bounds__(64) void moments_sweep(const SweepArgs a)
{
    const int64_t wave = blockIdx.x / a.K;
    const int k = (int)(blockIdx.x - wave * a.K);
    const int lane = threadIdx.x;
    const uint32_t c = a.cnt[wave], sw = a.swept[wave];
    if (c == sw) return;
    double2* s = reinterpret_cast<double2*>(a.msum) + (int64_t)k * a.macc_stride + wave * 64 + lane;
    double2* q = reinterpret_cast<double2*>(a.msumsq) + (int64_t)k * a.macc_stride + wave * 64 + lane;
    double2 sv = *s, qv = *q;
    for (uint32_t e = sw; e != c; ++e) {
        const int64_t slot = wave * a.depth + (int64_t)(e % (uint32_t)a.depth);
        const double w = a.ring_w[slot];
        const double2 x = a.ring[(slot * a.K + k) * 64 + lane];
        sv.x += x.x * w; sv.y += x.y * w;
        qv.x += (x.x * x.x) * w; qv.y += (x.y * x.y) * w;
    }
    *s = sv;
    *q = qv;
}
// after moments_sweep: everything posted has been folded
__global__ __launch_bounds__(256) void moments_swept(const uint32_t* cnt, uint32_t* swept, int64_t nwaves)
{
    const int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (w < nwaves) swept[w] = cnt[w];
}

__global__ void p2p_signal(const SignalArgs a)
{
    const SchedEntry sch = a.sched_index >= 0 ? a.sched_table[a.sched_index] : a.sched_inline;
    const unsigned long long done = 2ull * (unsigned long long)sch.gen + (unsigned long long)a.half + 1ull;
    __threadfence_system();
    if ((int)threadIdx.x < a.nranks)
        __hip_atomic_store(a.peer_flags[threadIdx.x] + a.me, done, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

#endif  // KMC_DEFINE_LAUNCH_KERNELS (advance_schedule: below)

#ifdef KMC_DEFINE_STATE_KERNELS    // kmc_state.hip
// Walker re-deal between sub-ensembles (kmc_config.deal_count, kmc_sampler_deal_pack / _unpack): a walker travels as
// one row of ndim + 2 doubles {its position, its log-pdf, (walker id << 32 | naccept)}.  PACK writes the row of local
// slot j to position t = (A j + C) mod S of the send buffer (a state-independent affine shuffle of this sub-ensemble's
// S slots; chunk t / (S / P) goes to sub-ensemble t / (S / P)); UNPACK takes row i of the receive buffer into slot i.
struct DealArgs {
    double*   pos;        // [S][ld]
    double*   logp;       // [S]
    uint32_t* naccept;    // [S]
    uint32_t* ids;        // [S] global walker index the slot currently holds
    double*   buf;        // [S][ndim + 2]
    int64_t   S;
    int64_t   A, C;       // pack only
    int32_t   ndim, ld;
};
__global__ __launch_bounds__(256) void deal_pack(const DealArgs a)
{
    const int64_t w = (int64_t)a.ndim + 2;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= a.S * w) return;
    const int64_t j = idx / w;
    const int e = (int)(idx - j * w);
    const int64_t t = (int64_t)(((uint64_t)a.A * (uint64_t)j + (uint64_t)a.C) % (uint64_t)a.S);
    double v;
    if (e < a.ndim) v = a.pos[j * a.ld + e];
    else if (e == a.ndim) v = a.logp[j];
    else v = __hiloint2double((int)a.ids[j], (int)a.naccept[j]);
    a.buf[t * w + e] = v;
}
__global__ __launch_bounds__(256) void deal_unpack(const DealArgs a)
{
    const int64_t w = (int64_t)a.ndim + 2;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= a.S * w) return;
    const int64_t j = idx / w;
    const int e = (int)(idx - j * w);
    const double v = a.buf[idx];
    if (e < a.ndim) a.pos[j * a.ld + e] = v;
    else if (e == a.ndim) a.logp[j] = v;
    else { a.naccept[j] = (uint32_t)__double2loint(v); a.ids[j] = (uint32_t)__double2hiint(v); }
}
__global__ __launch_bounds__(256) void deal_init_ids(uint32_t* ids, int64_t S, uint32_t first)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < S) ids[i] = first + (uint32_t)i;
}

#endif  // KMC_DEFINE_STATE_KERNELS

// (the copy / read-out kernels of kmc_copy.hip: kmc_copy_kernels.hpp -- not on the sampling path, and kept out of this file so that an edit
//  there does not invalidate the profile records of the kernels above, which are keyed to this file's hash)

#ifdef KMC_DEFINE_LAUNCH_KERNELS
// Graph replay support: the device-side generation counter and the schedule table of the next
// `n` generations (one thread each).  *gen += by happens before the table is rebuilt.
__global__ void advance_schedule(int64_t* gen, SchedEntry* table, int n, int64_t by,
                                 int64_t nburnin, int64_t nthin, int64_t nsamples, int64_t ring_slots)
{
    const int64_t base = *gen + by;
    __syncthreads();
    if ((int)threadIdx.x < n) table[threadIdx.x] = make_sched(base + threadIdx.x, nburnin, nthin, nsamples, ring_slots);
    if (threadIdx.x == 0) *gen = base;
}
#endif  // KMC_DEFINE_LAUNCH_KERNELS

}  // namespace kmc
