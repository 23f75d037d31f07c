// aql_kernels.hip -- device side of aql_probe.cpp: one half-step of the stretch move, one walker per lane, Gaussian, rows of ND doubles,
// in place (the shape of the product's two-launch kernels on short rows), in two forms:
//   half_step_plain  ordinary loads and stores: correct between launches only when the launch boundary makes the rows visible (acquire /
//                    release fences of the dispatch packets at agent scope, as the HIP runtime sets them);
//   half_step_sc     every row / log-pdf load `sc1` (never served from this XCD's L2), every store write-through `sc0 sc1`: the hand-off the
//                    guide lists as valid WITHOUT cache maintenance at the boundary -- for dispatch packets whose fences are NONE.
// Build: hipcc --genco --offload-arch=gfx950 -O3 -ffp-contract=off -I kissmcmc.jl_amd/csrc scripts/probes/aql_kernels.hip -o gpurun_out/aql_kernels.hsaco
#include <hip/hip_runtime.h>
#include "kmc_device.hpp"

using namespace kmc;

struct Args {
    double*       pos;     // [nw][ND]
    double*       logp;    // [nw]
    uint32_t*     nacc;    // [nw]
    DrawConsts    dc;
    DensityParams dp;
    uint32_t      h;
    uint32_t      step;    // 2 * generation + half
};

constexpr int ND = 4;

__device__ __forceinline__ void ld_plain(const double* p, double (&x)[ND])
{
#pragma unroll
    for (int c = 0; c < ND / 2; ++c) { const double2 t = reinterpret_cast<const double2*>(p)[c]; x[2 * c] = t.x; x[2 * c + 1] = t.y; }
}
__device__ __forceinline__ void ld_sc(const double* p, double (&x)[ND])
{
    typedef double v2d __attribute__((ext_vector_type(2)));
    v2d t, u;
    asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(t), "=&v"(u) : "v"(p) : "memory");
    x[0] = t.x; x[1] = t.y; x[2] = u.x; x[3] = u.y;
}
// `sc0` loads: this CU's L1 is bypassed, the XCD's L2 answers -- enough when every reader and writer sits on ONE XCD
__device__ __forceinline__ void ld_sc0(const double* p, double (&x)[ND])
{
    typedef double v2d __attribute__((ext_vector_type(2)));
    v2d t, u;
    asm volatile("global_load_dwordx4 %0, %2, off sc0\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(t), "=&v"(u) : "v"(p) : "memory");
    x[0] = t.x; x[1] = t.y; x[2] = u.x; x[3] = u.y;
}
__device__ __forceinline__ double ld_sc0(const double* p)
{
    double v;
    asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ double ld_sc(const double* p)
{
    double v;
    asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void st_wt(double* p, double a, double b)
{
    typedef double v2d __attribute__((ext_vector_type(2)));
    const v2d t = {a, b};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(p), "v"(t) : "memory");
}
__device__ __forceinline__ void st_wt(double* p, double a)
{
    asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(p), "v"(a) : "memory");
}

// dp.pad_ selects the coherence form (uniform): bit 0 `buffer_inv sc1` at wave entry (agent-scope invalidate of this CU's L1 and this XCD's
// non-coherent L2 lines), bit 1 write-through stores, bit 2 `buffer_wbl2 sc1` + wait at the end, bit 3 row loads `sc1`, bit 4 row loads `sc0` (one-XCD queues), bit 5 wait for every store's acknowledgement before the wave ends
template <bool SC>
__device__ __forceinline__ void body(const Args& a)
{
    const uint32_t i = blockIdx.x * 64u + threadIdx.x;
    const int mode = a.dp.pad_;
    if (mode & 1) asm volatile("buffer_inv sc1" ::: "memory");
    if (i >= a.h) return;
    const uint32_t half = a.step & 1u;
    const uint32_t w = half * a.h + i;
    const U4 bits = draw_bits(a.dc, (uint64_t)a.step, w);
    const uint32_t j = (1u - half) * a.h + draw_partner(a.dc, bits);
    double own[ND], oth[ND], p0;
    if (mode & 16) { ld_sc0(a.pos + (size_t)j * ND, oth); ld_sc0(a.pos + (size_t)w * ND, own); p0 = ld_sc0(a.logp + w); }
    else if (SC || (mode & 8)) { ld_sc(a.pos + (size_t)j * ND, oth); ld_sc(a.pos + (size_t)w * ND, own); p0 = ld_sc(a.logp + w); }
    else { ld_plain(a.pos + (size_t)j * ND, oth); ld_plain(a.pos + (size_t)w * ND, own); p0 = a.logp[w]; }
    const Draw dr = draw_finish(a.dc, bits);
    GaussianIso::Seq q;
    GaussianIso::seq_init(q);
    double y[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d) { y[d] = fma(dr.z, own[d] - oth[d], oth[d]); GaussianIso::seq_add(q, y[d], d, a.dp); }
    const double p1 = GaussianIso::seq_finish(q, ND, a.dp);
    if (accept_test(dr, p1, p0)) {
        if (SC || (mode & 2)) { st_wt(a.pos + (size_t)w * ND, y[0], y[1]); st_wt(a.pos + (size_t)w * ND + 2, y[2], y[3]); st_wt(a.logp + w, p1); }
        else {
#pragma unroll
            for (int c = 0; c < ND / 2; ++c) reinterpret_cast<double2*>(a.pos + (size_t)w * ND)[c] = make_double2(y[2 * c], y[2 * c + 1]);
            a.logp[w] = p1;
        }
        a.nacc[w] += 1u;          // (owner-only: the same lane's value of the launch two back)
    }
    if (mode & 4) asm volatile("s_waitcnt vmcnt(0)\n\tbuffer_wbl2 sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
    if (mode & 32) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

extern "C" __global__ __launch_bounds__(64) void half_step_plain(const Args a) { body<false>(a); }
extern "C" __global__ __launch_bounds__(64) void half_step_sc(const Args a) { body<true>(a); }

// where does a workgroup run?  out[blockIdx.x] = XCC_ID (bits 3:0) | HW_ID << 8  (one-XCD queue masks: which mask bits are which XCD)
extern "C" __global__ __launch_bounds__(64) void where_am_i(uint32_t* out)
{
    uint32_t xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    if (threadIdx.x == 0) out[blockIdx.x] = (xcc & 0xfu) | (hw << 8);
    for (volatile int spin = 0; spin < 2000; ++spin) { }                 // (stay a while: later workgroups must go elsewhere)
}
