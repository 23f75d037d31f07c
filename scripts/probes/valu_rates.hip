// Probe: what a gfx950 SIMD issues per nanosecond, by instruction kind -- the VALU instructions the generation kernels are made of.  Every SIMD of the chip runs
// W waves (1, 2, 4) of the same straight-line stream (four independent chains, 64 instructions per loop trip); HIP events around the launch -> ns per
// wave-instruction per SIMD.  Why: profiles/r06_c3_counters.json (generation_group<Rosenbrock,16,2>: 4 waves per SIMD x ~413 VALU instructions, VALU busy 18 %
// of the wave cycles) -- is that kernel near the SIMDs' issue limit or far from it?
//   hipcc --offload-arch=gfx950 -O2 scripts/probes/valu_rates.hip -o scripts/probes/valu_rates && scripts/probes/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))

template <int KIND>
__global__ void rates(double* sink, double seed, int trips)
{
    double a = seed + threadIdx.x, b = a * 1.5, c = a - 2.0, d = a + 3.0;
    unsigned x = threadIdx.x * 2654435761u + 1u, y = x ^ 0x9e3779b9u, z = x + 7u, w = y + 11u;
    unsigned long long p = x, q = y, r = z, s = w;
    for (int t = 0; t < trips; ++t) {
        if constexpr (KIND == 0) { REP16(asm volatile("v_fma_f64 %0, %0, %0, %0\n v_fma_f64 %1, %1, %1, %1\n v_fma_f64 %2, %2, %2, %2\n v_fma_f64 %3, %3, %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if constexpr (KIND == 1) { REP16(asm volatile("v_add_f64 %0, %0, %0\n v_add_f64 %1, %1, %1\n v_add_f64 %2, %2, %2\n v_add_f64 %3, %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if constexpr (KIND == 2) { REP16(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %5, %6, %1\n v_mad_u64_u32 %2, vcc, %6, %7, %2\n v_mad_u64_u32 %3, vcc, %7, %4, %3" : "+v"(p), "+v"(q), "+v"(r), "+v"(s) : "v"(x), "v"(y), "v"(z), "v"(w) : "vcc");) }
        if constexpr (KIND == 3) { REP16(asm volatile("v_mul_hi_u32 %0, %0, %1\n v_mul_hi_u32 %1, %1, %2\n v_mul_hi_u32 %2, %2, %3\n v_mul_hi_u32 %3, %3, %0" : "+v"(x), "+v"(y), "+v"(z), "+v"(w));) }
        if constexpr (KIND == 4) { REP16(asm volatile("v_xor_b32 %0, %0, %1\n v_xor_b32 %1, %1, %2\n v_xor_b32 %2, %2, %3\n v_xor_b32 %3, %3, %0" : "+v"(x), "+v"(y), "+v"(z), "+v"(w));) }
        if constexpr (KIND == 5) { REP16(asm volatile("v_lshl_add_u64 %0, %0, 1, %1\n v_lshl_add_u64 %1, %1, 1, %2\n v_lshl_add_u64 %2, %2, 1, %3\n v_lshl_add_u64 %3, %3, 1, %0" : "+v"(p), "+v"(q), "+v"(r), "+v"(s));) }
        if constexpr (KIND == 6) { REP16(asm volatile("v_mov_b32_dpp %0, %1 row_ror:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 row_ror:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 row_ror:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %0 row_ror:1 row_mask:0xf bank_mask:0xf" : "+v"(x), "+v"(y), "+v"(z), "+v"(w));) }
        if constexpr (KIND == 7) { REP16(asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc" : "+v"(x), "+v"(y), "+v"(z), "+v"(w) :: "vcc");) }
        if constexpr (KIND == 8) { REP16(asm volatile("s_xor_b32 s40, s40, s41\n s_add_u32 s41, s41, s42\n s_lshl_b32 s42, s42, 1\n s_xor_b32 s43, s43, s40" ::: "s40", "s41", "s42", "s43", "scc");) }
        if constexpr (KIND == 10) { REP16(asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[40:41]\n v_cndmask_b32_e64 %1, %1, %2, s[40:41]\n v_cndmask_b32_e64 %2, %2, %3, s[40:41]\n v_cndmask_b32_e64 %3, %3, %0, s[40:41]" : "+v"(x), "+v"(y), "+v"(z), "+v"(w) :: "s40", "s41");) }
        if constexpr (KIND == 11) { REP16(asm volatile("v_cmp_gt_u32 vcc, %0, %1\n v_cndmask_b32 %1, %1, %2, vcc\n v_cmp_gt_u32 vcc, %2, %3\n v_cndmask_b32 %3, %3, %0, vcc" : "+v"(x), "+v"(y), "+v"(z), "+v"(w) :: "vcc");) }
        if constexpr (KIND == 12) { REP16(asm volatile("v_bfi_b32 %0, %4, %0, %1\n v_bfi_b32 %1, %4, %1, %2\n v_bfi_b32 %2, %4, %2, %3\n v_bfi_b32 %3, %4, %3, %0" : "+v"(x), "+v"(y), "+v"(z), "+v"(w) : "v"(0xffff0000u));) }
        if constexpr (KIND == 13) { REP16(asm volatile("v_cmp_gt_f64 vcc, %0, %1\n v_cmp_gt_f64 vcc, %1, %2\n v_cmp_gt_f64 vcc, %2, %3\n v_cmp_gt_f64 vcc, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) :: "vcc");) }
        if constexpr (KIND == 14) { REP16(asm volatile("v_max_f64 %0, %0, %1\n v_min_f64 %1, %1, %2\n v_max_f64 %2, %2, %3\n v_min_f64 %3, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if constexpr (KIND == 15) { REP16(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0" : "+v"(x), "+v"(y), "+v"(z), "+v"(w));) }
        if constexpr (KIND == 16) { REP16(asm volatile("v_mul_f64 %0, %0, %1\n v_mul_f64 %1, %1, %2\n v_mul_f64 %2, %2, %3\n v_mul_f64 %3, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if constexpr (KIND == 9) { REP16(asm volatile("v_fma_f64 %0, %0, %0, %0\n s_xor_b32 s40, s40, s41\n v_xor_b32 %4, %4, %5\n s_add_u32 s41, s41, s42" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(x), "+v"(y) :: "s40", "s41", "s42", "scc");) }
    }
    if (a + b + c + d + (double)(x + y + z + w) + (double)(p + q + r + s) == 1.2345e300) sink[0] = a;
}

template <int KIND>
float run(int waves_per_simd, int trips, double* sink)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    rates<KIND><<<dim3(256), dim3(256 * waves_per_simd)>>>(sink, 1.0, 8);          // warm
    (void)hipEventRecord(e0);
    rates<KIND><<<dim3(256), dim3(256 * waves_per_simd)>>>(sink, 1.0, trips);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main()
{
    const char* names[] = {"v_fma_f64", "v_add_f64", "v_mad_u64_u32", "v_mul_hi_u32", "v_xor_b32", "v_lshl_add_u64", "v_mov_b32 dpp", "v_cndmask_b32", "s_xor/s_add/s_lshl (SALU)",
                           "mixed: fma_f64, s_xor, v_xor, s_add", "v_cndmask_b32_e64 (SGPR-pair mask)", "v_cmp_gt_u32 + v_cndmask_b32 pairs", "v_bfi_b32", "v_cmp_gt_f64", "v_max_f64 / v_min_f64", "v_mov_b32", "v_mul_f64"};
    double* sink;
    (void)hipMalloc(&sink, 8);
    const int trips = 4096;                                   // x 64 instructions per wave
    printf("grid: 256 workgroups (one per CU) of 4 W waves, W waves per SIMD; %d instructions per wave; ns = kernel time / instructions of ONE wave (>= one SIMD's issue time per W instructions)\n", trips * 64);
    for (int k = 0; k < 17; ++k) {
        printf("  %-36s", names[k]);
        for (int W : {1, 2, 4}) {
            float ms = 0.f;
            switch (k) { case 0: ms = run<0>(W, trips, sink); break; case 1: ms = run<1>(W, trips, sink); break; case 2: ms = run<2>(W, trips, sink); break; case 3: ms = run<3>(W, trips, sink); break;
                         case 4: ms = run<4>(W, trips, sink); break; case 5: ms = run<5>(W, trips, sink); break; case 6: ms = run<6>(W, trips, sink); break; case 7: ms = run<7>(W, trips, sink); break;
                         case 8: ms = run<8>(W, trips, sink); break; case 9: ms = run<9>(W, trips, sink); break; case 10: ms = run<10>(W, trips, sink); break; case 11: ms = run<11>(W, trips, sink); break;
                         case 12: ms = run<12>(W, trips, sink); break; case 13: ms = run<13>(W, trips, sink); break; case 14: ms = run<14>(W, trips, sink); break; case 15: ms = run<15>(W, trips, sink); break; default: ms = run<16>(W, trips, sink); }
            const double ns = ms * 1e6 / (trips * 64.0);
            printf("  W=%d: %6.2f ns per wave-instr, %5.2f ns per SIMD-instr", W, ns, ns / W);
        }
        printf("\n");
    }
    return 0;
}
