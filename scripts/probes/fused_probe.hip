// fused_probe.hip -- ONE launch per generation instead of two, without a join inside the launch?  (round 4, second bounded
// experiment for ensembles of 2 050 .. ~65 536 walkers with short rows, VERDICT r03 "missing" 3.)
//
// The multi-launch kernels pay the dependent-launch boundary (1.2-1.35 us) twice per generation, and for short rows that boundary IS
// the half-step (2.5 us at 4 096 x 4 and at 16 384 x 4).  A join inside the launch costs more than the boundary
// (profiles/r04_join_probe.txt).  This probe measures the third way: no join at all.  The draws are state-free (Philox keyed by
// (step, walker)), so a walker k of the SECOND half (reference src/samplers.jl:246-247, batch 2) does not have to wait for its partner
// j of the first half to be updated: it recomputes j's first-half-step update itself -- j's row and log-pdf as they were before the
// generation, j's own partner row (a second-half row, which nobody changes in the first half-step), j's draws -- through the SAME
// code the owner of j runs, and then makes its own move against the result.  Rows go from an input copy of the state to an output copy
// (every thread reads only the input copy), the copies swap per generation.  Cost: second-half walkers do two updates instead of
// one (1.5 x the arithmetic, 2.5 x the row reads -- nothing for 32-byte rows), the chain of a wave gets about 1.7 x as long; gain:
// one boundary per generation.
//
// The probe: Gaussian, ND doubles per row, one walker per lane, waves of 64; (a) two launches per generation, in place (the
// shape of the product's kernels), (b) the fused launch; both replayed from hipGraphs of 64 generations; final states compared
// bit for bit.
//
// Build + run (GPU box):  hipcc -O3 -ffp-contract=off --offload-arch=gfx950 -I kissmcmc.jl_amd/csrc scripts/probes/fused_probe.hip -o gpurun_out/fused_probe && gpurun_out/fused_probe
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "kmc_device.hpp"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

using namespace kmc;

struct Args {
    const double* pin;     // [nw][ND]
    double*       pout;    // [nw][ND]   (== pin for the two-launch form)
    const double* lin;     // [nw]
    double*       lout;
    uint32_t*     nacc;    // [nw]
    DrawConsts    dc;
    DensityParams dp;
    uint32_t      h;       // walkers per half
    uint32_t      gen;     // generation
};

template <int ND>
__device__ __forceinline__ void load_row(const double* p, double (&x)[ND])
{
    if constexpr (ND % 2 == 0) {
#pragma unroll
        for (int c = 0; c < ND / 2; ++c) { const double2 t = reinterpret_cast<const double2*>(p)[c]; x[2 * c] = t.x; x[2 * c + 1] = t.y; }
    } else {
#pragma unroll
        for (int d = 0; d < ND; ++d) x[d] = p[d];
    }
}
template <int ND>
__device__ __forceinline__ void store_row(double* p, const double (&x)[ND])
{
    if constexpr (ND % 2 == 0) {
#pragma unroll
        for (int c = 0; c < ND / 2; ++c) reinterpret_cast<double2*>(p)[c] = make_double2(x[2 * c], x[2 * c + 1]);
    } else {
#pragma unroll
        for (int d = 0; d < ND; ++d) p[d] = x[d];
    }
}

// (a) one half-step, in place: walker w = half * h + i
template <class Dens, int ND>
__global__ __launch_bounds__(64) void half_step(const Args a, int half)
{
    const uint32_t i = blockIdx.x * 64u + threadIdx.x;
    if (i >= a.h) return;
    const uint32_t w = (uint32_t)half * a.h + i;
    const uint64_t step = 2ull * a.gen + (uint64_t)half;
    const U4 bits = draw_bits(a.dc, step, w);
    const uint32_t j = (uint32_t)(1 - half) * a.h + draw_partner(a.dc, bits);
    double own[ND], oth[ND];
    load_row<ND>(a.pin + (size_t)j * ND, oth);
    load_row<ND>(a.pin + (size_t)w * ND, own);
    const double p0 = a.lin[w];
    const Draw dr = draw_finish(a.dc, bits);
    typename Dens::Seq q;
    Dens::seq_init(q);
    double y[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d) { y[d] = fma(dr.z, own[d] - oth[d], oth[d]); Dens::seq_add(q, y[d], d, a.dp); }
    const double p1 = Dens::seq_finish(q, ND, a.dp);
    if (accept_test(dr, p1, p0)) {
        store_row<ND>(a.pout + (size_t)w * ND, y);
        a.lout[w] = p1;
        a.nacc[w] += 1u;
    }
}

// (b) one generation: blocks [0, nb) carry the second half (the longer chain first), blocks [nb, 2 nb) the first half
template <class Dens, int ND>
__global__ __launch_bounds__(64) void generation(const Args a, uint32_t nb)
{
    const bool second = blockIdx.x < nb;
    const uint32_t i = (second ? blockIdx.x : blockIdx.x - nb) * 64u + threadIdx.x;
    if (i >= a.h) return;
    const uint32_t me = (second ? a.h : 0u) + i;
    // level 1 = my own move (step 2 gen + my half); level 0 (second half only) = my partner's first-half-step move
    const U4 mybits = draw_bits(a.dc, 2ull * a.gen + (second ? 1u : 0u), me);
    const uint32_t mypartner = (second ? 0u : a.h) + draw_partner(a.dc, mybits);
    double own[ND], oth[ND];
    double p0;
    U4 bits;
    uint32_t w;
    if (second) {
        w = mypartner;                                           // a first-half walker: its move of step 2 gen
        bits = draw_bits(a.dc, 2ull * a.gen, w);
        const uint32_t jp = a.h + draw_partner(a.dc, bits);       // its partner: a second-half row as it was before the generation
        load_row<ND>(a.pin + (size_t)jp * ND, oth);
        load_row<ND>(a.pin + (size_t)w * ND, own);
        p0 = a.lin[w];
    } else {
        w = me;
        bits = mybits;
        load_row<ND>(a.pin + (size_t)mypartner * ND, oth);
        load_row<ND>(a.pin + (size_t)w * ND, own);
        p0 = a.lin[w];
    }
    double myown[ND];
    double myp0 = 0.0;
    if (second) { load_row<ND>(a.pin + (size_t)me * ND, myown); myp0 = a.lin[me]; }
    bool acc = false;
    double p1 = 0.0;
    double y[ND];
    Draw dr;
#pragma unroll 1
    for (int level = second ? 0 : 1; level < 2; ++level) {       // ONE copy of the move's code: the partner's move and my own are the same instructions
        dr = draw_finish(a.dc, bits);
        typename Dens::Seq q;
        Dens::seq_init(q);
#pragma unroll
        for (int d = 0; d < ND; ++d) { y[d] = fma(dr.z, own[d] - oth[d], oth[d]); Dens::seq_add(q, y[d], d, a.dp); }
        p1 = Dens::seq_finish(q, ND, a.dp);
        acc = accept_test(dr, p1, p0);
        if (level == 0) {                                        // the partner as it stands after the first half-step -> my move
#pragma unroll
            for (int d = 0; d < ND; ++d) { oth[d] = acc ? y[d] : own[d]; own[d] = myown[d]; }
            p0 = myp0;
            bits = mybits;
        }
    }
#pragma unroll
    for (int d = 0; d < ND; ++d) y[d] = acc ? y[d] : own[d];
    store_row<ND>(a.pout + (size_t)me * ND, y);
    a.lout[me] = acc ? p1 : p0;
    if (acc) a.nacc[me] += 1u;
}

template <int ND>
static void run_case(uint32_t nw, int reps)
{
    const uint32_t h = nw / 2;
    const int G = 64;
    std::vector<double> pos0((size_t)nw * ND), lp0(nw);
    uint64_t s = 12345;
    for (auto& v : pos0) { s = s * 6364136223846793005ull + 1442695040888963407ull; v = ((double)(s >> 11) * 0x1.0p-53 - 0.5) * 4.0; }
    for (uint32_t w = 0; w < nw; ++w) { double q = 0; for (int d = 0; d < ND; ++d) { const double t = (pos0[(size_t)w * ND + d] - 0.0) * 1.0; q += t * t; } lp0[w] = -0.5 * q; }
    double *pA, *pB, *lA, *lB, *pC, *lC; uint32_t *nF, *nC;
    CK(hipMalloc(&pA, pos0.size() * 8)); CK(hipMalloc(&pB, pos0.size() * 8)); CK(hipMalloc(&pC, pos0.size() * 8));
    CK(hipMalloc(&lA, nw * 8)); CK(hipMalloc(&lB, nw * 8)); CK(hipMalloc(&lC, nw * 8));
    CK(hipMalloc(&nF, nw * 4)); CK(hipMalloc(&nC, nw * 4));
    auto reset = [&] {
        CK(hipMemcpy(pA, pos0.data(), pos0.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(pC, pos0.data(), pos0.size() * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(lA, lp0.data(), nw * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(lC, lp0.data(), nw * 8, hipMemcpyHostToDevice));
        CK(hipMemset(nF, 0, nw * 4)); CK(hipMemset(nC, 0, nw * 4));
    };
    Args base{};
    base.dc.seed_lo = 77u; base.dc.seed_hi = 1u; base.dc.nhalf = h;
    base.dc.c0 = std::sqrt(0.5); base.dc.c1 = std::sqrt(2.0) - std::sqrt(0.5); base.dc.nm1 = (double)(ND - 1);
    base.dp.p[0] = 0.0; base.dp.p[1] = 1.0; base.dp.ndim = ND;
    base.h = h;
    const uint32_t nb = (h + 63) / 64;
    hipStream_t st; CK(hipStreamCreate(&st));
    // graphs
    hipGraph_t g2, g1; hipGraphExec_t e2, e1;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
    for (int g = 0; g < G; ++g)
        for (int half = 0; half < 2; ++half) {
            Args a = base; a.pin = pC; a.pout = pC; a.lin = lC; a.lout = lC; a.nacc = nC; a.gen = (uint32_t)g;
            hipLaunchKernelGGL((half_step<GaussianIso, ND>), dim3(nb), dim3(64), 0, st, a, half);
        }
    CK(hipStreamEndCapture(st, &g2)); CK(hipGraphInstantiate(&e2, g2, nullptr, nullptr, 0));
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
    for (int g = 0; g < G; ++g) {
        Args a = base; a.gen = (uint32_t)g; a.nacc = nF;
        if (g % 2 == 0) { a.pin = pA; a.pout = pB; a.lin = lA; a.lout = lB; } else { a.pin = pB; a.pout = pA; a.lin = lB; a.lout = lA; }
        hipLaunchKernelGGL((generation<GaussianIso, ND>), dim3(2 * nb), dim3(64), 0, st, a, nb);
    }
    CK(hipStreamEndCapture(st, &g1)); CK(hipGraphInstantiate(&e1, g1, nullptr, nullptr, 0));
    // correctness: one replay each from the same state
    reset();
    CK(hipGraphLaunch(e2, st)); CK(hipGraphLaunch(e1, st)); CK(hipStreamSynchronize(st));
    std::vector<double> a1(pos0.size()), a2(pos0.size()), l1(nw), l2(nw);
    std::vector<uint32_t> n1(nw), n2(nw);
    CK(hipMemcpy(a1.data(), pA, a1.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(a2.data(), pC, a2.size() * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(l1.data(), lA, nw * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(l2.data(), lC, nw * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(n1.data(), nF, nw * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(n2.data(), nC, nw * 4, hipMemcpyDeviceToHost));
    const bool same = std::memcmp(a1.data(), a2.data(), a1.size() * 8) == 0 && std::memcmp(l1.data(), l2.data(), nw * 8) == 0 && n1 == n2;
    unsigned long long tot = 0; for (auto v : n1) tot += v;
    // timing
    hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    float ms2 = 0, ms1 = 0;
    for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(e2, st));
    CK(hipEventRecord(t0, st)); for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(e2, st)); CK(hipEventRecord(t1, st)); CK(hipEventSynchronize(t1)); CK(hipEventElapsedTime(&ms2, t0, t1));
    for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(e1, st));
    CK(hipEventRecord(t0, st)); for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(e1, st)); CK(hipEventRecord(t1, st)); CK(hipEventSynchronize(t1)); CK(hipEventElapsedTime(&ms1, t0, t1));
    const double us2 = ms2 * 1e3 / (reps * G * 2.0), us1 = ms1 * 1e3 / (reps * G * 2.0);
    std::printf("%8u x %d | two launches %6.3f us per half-step | one launch per generation %6.3f us per half-step (%.2fx) | final state %s, accept %.3f\n",
                nw, ND, us2, us1, us2 / us1, same ? "bit-identical" : "DIFFERENT", (double)tot / ((double)nw * G));
    CK(hipGraphExecDestroy(e1)); CK(hipGraphExecDestroy(e2)); CK(hipGraphDestroy(g1)); CK(hipGraphDestroy(g2));
    CK(hipFree(pA)); CK(hipFree(pB)); CK(hipFree(pC)); CK(hipFree(lA)); CK(hipFree(lB)); CK(hipFree(lC)); CK(hipFree(nF)); CK(hipFree(nC));
    CK(hipStreamDestroy(st));
}

int main()
{
    std::printf("fused_probe: Gaussian, one walker per lane, waves of 64, hipGraph replay of 64 generations; us per half-step = replay time / 128\n");
    for (uint32_t nw : {2050u, 4096u, 10000u, 16384u, 65536u, 262144u}) run_case<4>(nw, 40);
    for (uint32_t nw : {4096u, 16384u, 65536u}) run_case<2>(nw, 40);
    for (uint32_t nw : {4096u, 16384u, 65536u}) run_case<8>(nw, 40);
    return 0;
}
