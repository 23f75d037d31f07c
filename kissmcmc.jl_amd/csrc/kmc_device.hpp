// kmc_device.hpp -- device-side building blocks of the stretch-move half-step (gfx950).
//
//   Philox4x32-10 counter RNG + the three per-walker-step draws   (reference src/samplers.jl:250,252,260)
//   stretch-factor inverse CDF                                     (reference src/samplers.jl:227)
//   the log-density menu standing in for the user closure          (reference src/samplers.jl:257)
//
// Arithmetic contract (mirrored by the CPU oracle so that accept decisions and positions agree
// bit for bit): compiled with -ffp-contract=off; every fused multiply-add is an explicit fma().
#pragma once
#ifndef __HIPCC_RTC__            // hiprtc (runtime-compiled user densities) brings its own HIP prelude
#include <hip/hip_runtime.h>
#include <stdint.h>
#else
typedef int                int32_t;
typedef unsigned int       uint32_t;
typedef long long          int64_t;
typedef unsigned long long uint64_t;
#ifndef INFINITY
#define INFINITY (__builtin_inf())
#endif
#endif

namespace kmc {

// ------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11).  One call = 128 random bits for one walker-step.
// ------------------------------------------------------------------------------------------
struct U4 { uint32_t x, y, z, w; };

__device__ __forceinline__ U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                            uint32_t k0, uint32_t k1)
{
    constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)M0 * c0;   // one v_mad_u64_u32 each
        const uint64_t p1 = (uint64_t)M1 * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c0 = n0; c1 = (uint32_t)p1; c2 = n2; c3 = (uint32_t)p0;
        k0 += W0; k1 += W1;
    }
    return U4{c0, c1, c2, c3};
}

// The random inputs of one walker-step.
struct Draw {
    uint32_t partner;  // index into the complementary half, uniform with replacement  (:250)
    double   z;        // stretch factor ~ g                                           (:252)
    double   t1;       // (N-1) * log z                                                (:260)
    double   lu;       // log(rand())                                                  (:260)
};

struct DrawConsts {
    uint32_t seed_lo, seed_hi;
    uint32_t nhalf;
    double   c0, c1;   // sqrt(1/a), sqrt(a) - sqrt(1/a)   (:227, hoisted)
    double   nm1;      // N - 1
};

// Natural logarithm for the two logarithms of the accept test (reference src/samplers.jl:260): arguments are
// positive, finite and normal (z in [1/a, a]; u in [2^-53, 1)), so none of libm's special-case handling is needed.
// The fdlibm / musl algorithm (argument reduced to [sqrt(1/2), sqrt(2)), s = f / (2 + f), degree-7 polynomial in
// s^2, split ln 2), error < 1 ulp -- about 45 instructions where the device libm's log takes 80-135, and these two
// logarithms are a quarter of the half-step kernel's VALU work.  Not used for log-pdfs.
//
// The constants (ln2_hi, ln2_lo, Lg1..Lg7) and the evaluation order are those of FreeBSD msun's e_log.c as carried by
// musl 1.2.x (src/math/log.c, the pre-1.2.0 fdlibm form; checked against fdlibm 5.3's e_log.c: identical values), whose
// notice is preserved here as it asks:
//
//   ====================================================
//   Copyright (C) 1993 by Sun Microsystems, Inc. All rights reserved.
//
//   Developed at SunSoft, a Sun Microsystems, Inc. business.
//   Permission to use, copy, modify, and distribute this
//   software is freely granted, provided that this notice
//   is preserved.
//   ====================================================
__device__ __forceinline__ double log_pos_normal(double x)
{
    constexpr double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                     Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                     Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                     Lg7 = 1.479819860511658591e-01;
    uint32_t hx = (uint32_t)__double2hiint(x);
    const uint32_t lx = (uint32_t)__double2loint(x);
    hx += 0x3ff00000u - 0x3fe6a09eu;                      // reduce x into [sqrt(2)/2, sqrt(2))
    const int k = (int)(hx >> 20) - 0x3ff;
    hx = (hx & 0x000fffffu) + 0x3fe6a09eu;
    const double m = __hiloint2double((int)hx, (int)lx);
    const double f = m - 1.0;
    const double hfsq = 0.5 * f * f;
    const double s = f / (2.0 + f);
    const double z = s * s;
    const double w = z * z;
    const double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    const double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    const double R = t2 + t1;
    const double dk = (double)k;
    return s * (hfsq + R) + dk * ln2_lo - hfsq + f + dk * ln2_hi;
}

// Split in two so a kernel can issue the partner-row loads (which need only the partner index)
// before it spends ~100 instructions on the two logarithms.
__device__ __forceinline__ U4 draw_bits(const DrawConsts& dc, uint64_t step, uint64_t walker)
{
    return philox4x32_10((uint32_t)step, (uint32_t)(step >> 32),
                         (uint32_t)walker, (uint32_t)(walker >> 32), dc.seed_lo, dc.seed_hi);
}
__device__ __forceinline__ uint32_t draw_partner(const DrawConsts& dc, const U4& w)
{
    return __umulhi(w.x, dc.nhalf);
}
__device__ __forceinline__ Draw draw_finish(const DrawConsts& dc, const U4& w)
{
    Draw d;
    d.partner = __umulhi(w.x, dc.nhalf);
    const double uz = ((double)w.y + 0.5) * 0x1.0p-32;
    const double t  = fma(uz, dc.c1, dc.c0);
    d.z = t * t;
    const uint64_t k = ((uint64_t)w.z << 20) | (uint64_t)(w.w >> 12);
    const double ua = ((double)k + 0.5) * 0x1.0p-52;
    d.t1 = dc.nm1 * log_pos_normal(d.z);
    d.lu = log_pos_normal(ua);
    return d;
}
__device__ __forceinline__ Draw draw_step(const DrawConsts& dc, uint64_t step, uint64_t walker)
{
    return draw_finish(dc, draw_bits(dc, step, walker));
}

// Workgroup barrier for kernels whose shared state is in LDS only: wait for this wave's LDS operations, then the barrier.
// __syncthreads() also waits for every outstanding GLOBAL load (its fence is over all address spaces), which would put a
// load issued a generation ahead of its use back on the critical path.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// (N-1) log z + p1 - p0 >= log u, evaluated left to right like the reference (:260).
__device__ __forceinline__ bool accept_test(const Draw& d, double p1, double p0)
{
    return ((d.t1 + p1) - p0) >= d.lu;
}

// ------------------------------------------------------------------------------------------
// Density menu.  Two interfaces per density:
//   Seq  : element-by-element in index order (generic one-walker-per-lane kernel, initial
//          log-pdf evaluation); same summation order as the oracle's scalar loop.
//   frag : a walker's row striped over L lanes, K chunks of 2 doubles per lane
//          (lane j, chunk k holds elements 2(kL+j), 2(kL+j)+1); returns the lane's partial
//          sum S_j; the kernel reduces over lanes and calls finish(S).
// Parameters are pre-digested on the host into DensityParams.
// ------------------------------------------------------------------------------------------
struct DensityParams { double p[6]; int32_t ndim; int32_t pad_; };

// ---- cross-lane helpers (wave64; DPP where the pattern allows, LDS crossbar otherwise) ----
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    if constexpr (CTRL == 0x130) {                  // wave_shl: lane 63 has no source and keeps its own value
        lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
        hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
    } else {
        // every lane has a source (quad_perm / row_mirror / row_ror / row_newbcast, all rows and banks enabled), so
        // the "old" operand is never read: leave it undefined instead of paying a v_mov per word to initialise it
        int ol, oh;
        asm volatile("" : "=v"(ol));
        asm volatile("" : "=v"(oh));
        lo = __builtin_amdgcn_update_dpp(ol, lo, CTRL, 0xF, 0xF, false);
        hi = __builtin_amdgcn_update_dpp(oh, hi, CTRL, 0xF, 0xF, false);
    }
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double bperm_f64(int byte_addr, double v)
{
    const int lo = __builtin_amdgcn_ds_bpermute(byte_addr, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(byte_addr, __double2hiint(v));
    return __hiloint2double(hi, lo);
}
// v + (v of lane ^ 16) and v + (v of lane ^ 32) with gfx950's v_permlane{16,32}_swap: the swap of
// a value with a copy of itself leaves {even rows, even rows} in one register and {odd rows, odd
// rows} in the other, so their sum is the xor-butterfly level -- two VALU ops per 32-bit word,
// no LDS crossbar round trip.
__device__ __forceinline__ double xor16_sum(double v)
{
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
}
__device__ __forceinline__ double xor32_sum(double v)
{
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
}

// Sum over the L lanes of a group; every lane of the group ends with the same bits
// (each butterfly level adds the same two operands on both sides).
template <int L>
__device__ __forceinline__ double group_sum(double v)
{
    if constexpr (L >= 2)  v += dpp_f64<0xB1>(v);    // quad_perm [1,0,3,2]  : lane ^ 1
    if constexpr (L >= 4)  v += dpp_f64<0x4E>(v);    // quad_perm [2,3,0,1]  : lane ^ 2
    if constexpr (L >= 8)  v += dpp_f64<0x141>(v);   // row_half_mirror      : other quad pair
    if constexpr (L >= 16) v += dpp_f64<0x140>(v);   // row_mirror           : other 8-lane half
    if constexpr (L >= 32) v = xor16_sum(v);
    if constexpr (L >= 64) v = xor32_sum(v);
    return v;
}
// Sum of the values held by lanes {j, j+L, j+2L, ...} (one per group); valid in group 0.
template <int L>
__device__ __forceinline__ double wave_fold(double v)
{
    if constexpr (L <= 1)  v += dpp_f64<0xB1>(v);
    if constexpr (L <= 2)  v += dpp_f64<0x4E>(v);
    if constexpr (L <= 4)  v += dpp_f64<0x124>(v);   // row_ror:4
    if constexpr (L <= 8)  v += dpp_f64<0x128>(v);   // row_ror:8
    if constexpr (L <= 16) v = xor16_sum(v);
    if constexpr (L <= 32) v = xor32_sum(v);
    return v;
}
template <int L>
__device__ __forceinline__ double group_shfl_down1(double v)
{   // value of lane j+1 (wave_shl:1); lane L-1 of a group receives the next group's lane 0 -- unused
    return dpp_f64<0x130>(v);
}
template <int L>
__device__ __forceinline__ double group_bcast0(double v)
{   // value of the group's lane 0, in every lane of the group
    if constexpr (L == 1) return v;
    else if constexpr (L == 2) return dpp_f64<0xA0>(v);                  // quad_perm [0,0,2,2]
    else if constexpr (L == 4) return dpp_f64<0x00>(v);                  // quad_perm [0,0,0,0]
    else if constexpr (L == 16) return dpp_f64<0x150>(v);                // row_newbcast:0 (gfx90a+): lane 0 of each 16-lane row
    else if constexpr (L == 8) {                                        // two groups per row: lanes 0 and 8
        const double lo = dpp_f64<0x150>(v), hi = dpp_f64<0x158>(v);
        return (threadIdx.x & 8) ? hi : lo;
    } else return __shfl(v, 0, L);
}

struct GaussianIso {   // p = {mu, 1/sigma}
    static constexpr bool kHasFrag = true;
    struct Seq { double s; };
    __device__ static void seq_init(Seq& q) { q.s = 0.0; }
    __device__ static void seq_add(Seq& q, double x, int, const DensityParams& P)
    {
        const double t = (x - P.p[0]) * P.p[1];
        q.s += t * t;
    }
    __device__ static double seq_finish(const Seq& q, int, const DensityParams&) { return -0.5 * q.s; }

    template <int L, int K>
    __device__ static double frag_partial(const double2 (&y)[K], int j, int ndim, const DensityParams& P)
    {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int e0 = 2 * (k * L + j);
            const double t0 = (y[k].x - P.p[0]) * P.p[1];
            const double t1 = (y[k].y - P.p[0]) * P.p[1];
            s += (e0 < ndim) ? t0 * t0 : 0.0;       // elements >= ndim are padding (ragged last chunk)
            s += (e0 + 1 < ndim) ? t1 * t1 : 0.0;
        }
        return s;
    }
    __device__ static double finish(double S, const DensityParams&) { return -0.5 * S; }
};

struct Exponential {   // p = {rate};  README.md:15  x<0 ? -Inf : -x
    static constexpr bool kHasFrag = true;
    struct Seq { double s; bool neg; };
    __device__ static void seq_init(Seq& q) { q.s = 0.0; q.neg = false; }
    __device__ static void seq_add(Seq& q, double x, int, const DensityParams&)
    {
        q.neg = q.neg || (x < 0.0);
        q.s += x;
    }
    __device__ static double seq_finish(const Seq& q, int, const DensityParams& P)
    {
        return q.neg ? -INFINITY : -(P.p[0] * q.s);
    }
    template <int L, int K>
    __device__ static double frag_partial(const double2 (&y)[K], int j, int ndim, const DensityParams&)
    {
        double s = 0.0;
        bool neg = false;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int e0 = 2 * (k * L + j);
            const bool v0 = e0 < ndim, v1 = e0 + 1 < ndim;
            neg = neg || (v0 && y[k].x < 0.0) || (v1 && y[k].y < 0.0);
            s += v0 ? y[k].x : 0.0;
            s += v1 ? y[k].y : 0.0;
        }
        return neg ? INFINITY : s;   // +inf propagates through the lane reduction
    }
    __device__ static double finish(double S, const DensityParams& P)
    {
        return (S == INFINITY) ? -INFINITY : -(P.p[0] * S);
    }
};

struct Rosenbrock {   // p = {a, b, 1/scale}; chained form, reduces to test/runtests.jl:68 at N = 2
    static constexpr bool kHasFrag = true;
    struct Seq { double s, prev; };
    __device__ static void seq_init(Seq& q) { q.s = 0.0; q.prev = 0.0; }
    __device__ static void seq_add(Seq& q, double x, int d, const DensityParams& P)
    {
        if (d > 0) {
            const double dd = x - q.prev * q.prev;
            const double e  = P.p[0] - q.prev;
            q.s += P.p[1] * (dd * dd) + e * e;
        }
        q.prev = x;
    }
    __device__ static double seq_finish(const Seq& q, int, const DensityParams& P) { return -(q.s * P.p[2]); }

    template <int L, int K>
    __device__ static double frag_partial(const double2 (&y)[K], int j, int ndim, const DensityParams& P)
    {
        const double a = P.p[0], b = P.p[1];
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const double x0 = y[k].x, x1 = y[k].y;
            // term i = 2(kL+j): uses x_{i+1} = x1 (exists unless the row ends inside this chunk)
            if (2 * (k * L + j) < ndim - 1) {
                const double dd = x1 - x0 * x0;
                const double e  = a - x0;
                s += b * (dd * dd) + e * e;
            }
            // term i+1: needs x_{i+2} = first element of the next lane (or of chunk k+1, lane 0)
            double nxt = group_shfl_down1<L>(x0);
            if (k + 1 < K) {
                const double wrap = group_bcast0<L>(y[k + 1 < K ? k + 1 : k].x);
                nxt = (j == L - 1) ? wrap : nxt;
            }
            const int i1 = 2 * (k * L + j) + 1;
            if (i1 < ndim - 1) {
                const double dd = nxt - x1 * x1;
                const double e  = a - x1;
                s += b * (dd * dd) + e * e;
            }
        }
        return s;
    }
    __device__ static double finish(double S, const DensityParams& P) { return -(S * P.p[2]); }
};

struct LogNormal {   // p = {mu, sigma}
    static constexpr bool kHasFrag = true;
    template <int L, int K>
    __device__ static double frag_partial(const double2 (&y)[K], int j, int ndim, const DensityParams& P)
    {
        double s = 0.0;
        bool bad = false;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int e0 = 2 * (k * L + j);
            const double xs[2] = {y[k].x, y[k].y};
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                if (e0 + e < ndim) {
                    if (!(xs[e] > 0.0)) bad = true;
                    else {
                        const double lx = log(xs[e]);
                        const double t = (lx - P.p[0]) / P.p[1];
                        s += -lx - 0.5 * t * t;
                    }
                }
            }
        }
        return bad ? -INFINITY : s;   // -inf propagates through the lane reduction
    }
    __device__ static double finish(double S, const DensityParams&) { return S; }
    struct Seq { double s; bool bad; };
    __device__ static void seq_init(Seq& q) { q.s = 0.0; q.bad = false; }
    __device__ static void seq_add(Seq& q, double x, int, const DensityParams& P)
    {
        if (!(x > 0.0)) { q.bad = true; return; }
        const double lx = log(x);
        const double t = (lx - P.p[0]) / P.p[1];
        q.s += -lx - 0.5 * t * t;
    }
    __device__ static double seq_finish(const Seq& q, int, const DensityParams&) { return q.bad ? -INFINITY : q.s; }
};

struct MvNormal2 {   // p = {m1, m2, P11, P12, P22}
    static constexpr bool kHasFrag = true;
    template <int L, int K>
    __device__ static double frag_partial(const double2 (&y)[K], int j, int, const DensityParams& P)
    {   // ndim == 2: the whole row is chunk 0 of lane 0 of the group
        const double d0 = y[0].x - P.p[0], d1 = y[0].y - P.p[1];
        const double r = -0.5 * (P.p[2] * d0 * d0 + 2.0 * P.p[3] * d0 * d1 + P.p[4] * d1 * d1);
        return j == 0 ? r : 0.0;
    }
    __device__ static double finish(double S, const DensityParams&) { return S; }
    struct Seq { double d0, r; };
    __device__ static void seq_init(Seq& q) { q.d0 = 0.0; q.r = 0.0; }
    __device__ static void seq_add(Seq& q, double x, int d, const DensityParams& P)
    {
        if (d == 0) q.d0 = x - P.p[0];
        else if (d == 1) {
            const double d1 = x - P.p[1];
            q.r = -0.5 * (P.p[2] * q.d0 * q.d0 + 2.0 * P.p[3] * q.d0 * d1 + P.p[4] * d1 * d1);
        }
    }
    __device__ static double seq_finish(const Seq& q, int, const DensityParams&) { return q.r; }
};

// ------------------------------------------------------------------------------------------
// Host-evaluated density (KMC_HOST_DENSITY): the log-pdf of a proposal is whatever the caller's
// callback returned for it -- the reference's arbitrary `pdf` closure (src/samplers.jl:257) kept on
// the host.  A half-step is then two launches of the generic kernel: a PROPOSE pass that writes the
// proposals for the host to evaluate, and an ACCEPT pass that recomputes the same proposals (same
// counter-based draws, untouched rows) and reads their log-pdfs from p1_in.
// ------------------------------------------------------------------------------------------
struct HostEval {
    static constexpr bool kHasFrag = false;
    static constexpr bool kHostEval = true;
    struct Seq { };
    __device__ static void seq_init(Seq&) { }
    __device__ static void seq_add(Seq&, double, int, const DensityParams&) { }
    __device__ static double seq_finish(const Seq&, int, const DensityParams&) { return 0.0; }
};
template <class D, class = void> struct HostEvalTrait { static constexpr bool value = false; };
template <class D> struct HostEvalTrait<D, decltype((void)D::kHostEval)> { static constexpr bool value = true; };

// ------------------------------------------------------------------------------------------
// User-supplied densities (runtime-compiled with hiprtc, the device-side answer to the reference's
// arbitrary `pdf` closure, src/samplers.jl:257):
//     log p(x) = sum_d F::term(x_d, d, n, p)  +  sum_{d < n-1} F::pair(x_d, x_{d+1}, d, n, p)
// F is a functor struct generated from the user's two C expressions.  A term may be -INFINITY to
// reject a proposal (it propagates through the sums).
// ------------------------------------------------------------------------------------------
template <class F>
struct TermPairDensity {
    static constexpr bool kHasFrag = true;
    struct Seq { double s, prev; };
    __device__ static void seq_init(Seq& q) { q.s = 0.0; q.prev = 0.0; }
    __device__ static void seq_add(Seq& q, double x, int d, const DensityParams& P)
    {
        if (F::kHasPair && d > 0) q.s += F::pair(q.prev, x, d - 1, P.ndim, P.p);
        q.s += F::term(x, d, P.ndim, P.p);
        q.prev = x;
    }
    __device__ static double seq_finish(const Seq& q, int, const DensityParams&) { return q.s; }

    template <int L, int K>
    __device__ static double frag_partial(const double2 (&y)[K], int j, int ndim, const DensityParams& P)
    {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int e0 = 2 * (k * L + j);
            const double x0 = y[k].x, x1 = y[k].y;
            if (e0 < ndim) s += F::term(x0, e0, ndim, P.p);
            if (e0 + 1 < ndim) s += F::term(x1, e0 + 1, ndim, P.p);
            if constexpr (F::kHasPair) {
                double nxt = group_shfl_down1<L>(x0);           // executed by every lane (no divergence)
                if (k + 1 < K) {
                    const double wrap = group_bcast0<L>(y[k + 1 < K ? k + 1 : k].x);
                    nxt = (j == L - 1) ? wrap : nxt;
                }
                if (e0 < ndim - 1) s += F::pair(x0, x1, e0, ndim, P.p);
                if (e0 + 1 < ndim - 1) s += F::pair(x1, nxt, e0 + 1, ndim, P.p);
            }
        }
        return s;
    }
    __device__ static double finish(double S, const DensityParams&) { return S; }
};

// A function body recognised as a sum over elements (kmc_rtc.hip: recognise_separable): F::term / F::pair are the loop body,
// F::finish the return expression over the sum.
template <class F>
struct SepDensity : TermPairDensity<F> {
    __device__ static double seq_finish(const typename TermPairDensity<F>::Seq& q, int, const DensityParams& P) { return F::finish(q.s, P.ndim, P.p); }
    __device__ static double finish(double S, const DensityParams& P) { return F::finish(S, P.ndim, P.p); }
};

// ... with SEVERAL sums (2..4): F::elem(x_d, x_{d+1}, d, n, p, acc) adds one element's increments to all of them in one pass
// (the loop body, verbatim), F::finish(acc, n, p) is the return expression.  kHasPair: the loop ran over d < n - 1 and may read x_{d+1}.
template <class F>
struct SepDensityN {
    static constexpr bool kHasFrag = true;
    static constexpr int kNSums = F::kNAcc;
    struct Seq { double s[F::kNAcc]; double prev; };
    __device__ static void seq_init(Seq& q) {
#pragma unroll
        for (int a = 0; a < F::kNAcc; ++a) q.s[a] = 0.0;
        q.prev = 0.0;
    }
    __device__ static void seq_add(Seq& q, double x, int d, const DensityParams& P)
    {
        if constexpr (F::kHasPair) { if (d > 0) F::elem(q.prev, x, d - 1, P.ndim, P.p, q.s); }
        else F::elem(x, 0.0, d, P.ndim, P.p, q.s);
        q.prev = x;
    }
    __device__ static double seq_finish(const Seq& q, int, const DensityParams& P) { return F::finish(q.s, P.ndim, P.p); }

    // this lane's chunks' increments to every sum (same element <-> lane map as TermPairDensity::frag_partial)
    template <int L, int K>
    __device__ static void frag_partial_n(const double2 (&y)[K], int j, int ndim, const DensityParams& P, double (&S)[F::kNAcc])
    {
#pragma unroll
        for (int a = 0; a < F::kNAcc; ++a) S[a] = 0.0;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int e0 = 2 * (k * L + j);
            const double x0 = y[k].x, x1 = y[k].y;
            if constexpr (F::kHasPair) {
                double nxt = group_shfl_down1<L>(x0);           // executed by every lane (no divergence)
                if (k + 1 < K) {
                    const double wrap = group_bcast0<L>(y[k + 1 < K ? k + 1 : k].x);
                    nxt = (j == L - 1) ? wrap : nxt;
                }
                if (e0 < ndim - 1) F::elem(x0, x1, e0, ndim, P.p, S);
                if (e0 + 1 < ndim - 1) F::elem(x1, nxt, e0 + 1, ndim, P.p, S);
            } else {
                if (e0 < ndim) F::elem(x0, 0.0, e0, ndim, P.p, S);
                if (e0 + 1 < ndim) F::elem(x1, 0.0, e0 + 1, ndim, P.p, S);
            }
        }
    }
    __device__ static double finish_n(const double (&S)[F::kNAcc], const DensityParams& P) { return F::finish(S, P.ndim, P.p); }
    // (unused single-sum forms, so that every vector-kernel instantiation finds the names)
    template <int L, int K> __device__ static double frag_partial(const double2 (&)[K], int, int, const DensityParams&) { return 0.0; }
    __device__ static double finish(double S, const DensityParams&) { return S; }
};
template <class D, class = void> struct MultiSumTrait { static constexpr int n = 0; };
template <class D> struct MultiSumTrait<D, decltype((void)D::kNSums)> { static constexpr int n = D::kNSums; };

// ------------------------------------------------------------------------------------------
// A log-density given as a whole function over the proposal vector (runtime-compiled, kmc_user_density_create_body):
// the kernels that walk a row element by element (generic half-step, initial log-pdfs, initial ball, Metropolis) collect
// the MAXD = ndim elements in per-lane storage and evaluate F::eval(x, n, p) once.  No lane-striped form (kHasFrag).
// ------------------------------------------------------------------------------------------
template <class F, int MAXD>
struct BodyDensity {
    static constexpr bool kHasFrag = false;
    static constexpr int kRowEval = MAXD;            // the vector kernel may hand it whole proposals: eval_row(x[0..MAXD), n, P)
    struct Seq { double x[MAXD]; };
    __device__ static void seq_init(Seq&) {}
    __device__ static void seq_add(Seq& q, double v, int d, const DensityParams&) { if (d < MAXD) q.x[d] = v; }
    __device__ static double seq_finish(const Seq& q, int ndim, const DensityParams& P) { return F::eval(q.x, ndim, P.p); }
    __device__ static double eval_row(const double* x, int ndim, const DensityParams& P) { return F::eval(x, ndim, P.p); }
    // (never called: a body has no lane-striped form; the vector kernel evaluates whole rows instead, see RowEvalTrait)
    template <int L, int K> __device__ static double frag_partial(const double2 (&)[K], int, int, const DensityParams&) { return 0.0; }
    __device__ static double finish(double S, const DensityParams&) { return S; }
};
// A density the vector kernel evaluates per WALKER on the whole proposal (collected through LDS) instead of per lane on its chunks
template <class D, class = void> struct RowEvalTrait { static constexpr int n = 0; };
template <class D> struct RowEvalTrait<D, decltype((void)D::kRowEval)> { static constexpr int n = D::kRowEval; };

// ... returning a BLOB next to the log-density: the reference's `pdf(theta) -> (p, blob)` with hasblob=true
// (src/samplers.jl:150-151, :194-196, :257), as NB doubles per evaluation.  F::eval(x, n, p, blob) fills blob[0..NB); the
// kernels keep the blob of every walker's CURRENT position next to its log-pdf (blob0s[nc] = blob1 on accept, :264) and
// store it with every sample (reduce_blob!, :270).
template <class F, int MAXD, int NB>
struct BodyBlobDensity {
    static constexpr bool kHasFrag = false;
    static constexpr int kBlob = NB;
    struct Seq { double x[MAXD]; double blob[NB]; };
    __device__ static void seq_init(Seq& q) {
#pragma unroll 1
        for (int i = 0; i < NB; ++i) q.blob[i] = 0.0;
    }
    __device__ static void seq_add(Seq& q, double v, int d, const DensityParams&) { if (d < MAXD) q.x[d] = v; }
    __device__ static double seq_finish(Seq& q, int ndim, const DensityParams& P) { return F::eval(q.x, ndim, P.p, q.blob); }
    // the vector kernel hands it whole proposals like a BodyDensity; blob[0..NB) is zero on entry
    static constexpr int kRowEval = MAXD;
    __device__ static double eval_row(const double* x, int ndim, const DensityParams& P, double* blob) { return F::eval(x, ndim, P.p, blob); }
    template <int L, int K> __device__ static double frag_partial(const double2 (&)[K], int, int, const DensityParams&) { return 0.0; }
    __device__ static double finish(double S, const DensityParams&) { return S; }
};
template <class D, class = void> struct BlobTrait { static constexpr int n = 0; };
template <class D> struct BlobTrait<D, decltype((void)D::kBlob)> { static constexpr int n = D::kBlob; };

}  // namespace kmc
