// join_probe.hip -- what would a half-step cost INSIDE one launch?  (round 4, bounded experiment: a multi-workgroup resident kernel
// for 2 050 .. 16 384 walkers with short rows)
//
// The skeleton of such a kernel without its arithmetic: W workgroups of T threads, one active walker per thread and half-step, the
// thread's two own rows (one per half, ND doubles) in registers, per half-step
//     partner row <- global memory, sc1 loads (written by any workgroup in an earlier half-step)
//     "accept" (a hash of (walker, step): ~1/2) -> the new row goes to global memory, sc1 write-through stores
//     every storing wave drains (s_waitcnt vmcnt(0)), workgroup barrier, ONE lane signals, ONE wave polls, workgroup barrier
// which is the hand-off of MI355X_MICROARCH.md "Valid forms", first table row (every store and every load of the handed-off bytes
// sc1, one signalling lane per workgroup behind the barrier, one workgroup per CU) -- the join of reference src/samplers.jl:273
// inside the launch instead of at a kernel boundary (1.2-1.35 us measured, profiles/r03_probe_timeline.txt).
// Two signalling forms: "flags" (workgroup w stores its step into word w of one 128-byte line, the poller reads the line) and
// "counter" (agent-scope atomic add, the poller reads the one word).
// Every partner row is CHECKED: row content is a pure function of (walker, last accepted step), and the reader recomputes the
// step the partner was last accepted at -- a stale row (a hand-off that does not hold) is counted, not assumed away.
//
// Build + run (GPU box):  hipcc -O3 -ffp-contract=off --offload-arch=gfx950 scripts/probes/join_probe.hip -o gpurun_out/join_probe && gpurun_out/join_probe
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

constexpr int ND = 4;                       // doubles per row (two 16-byte chunks)

__device__ __forceinline__ uint32_t mix(uint32_t a, uint32_t b)
{
    uint32_t x = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA77u;
    x ^= x >> 15; x *= 0xC2B2AE3Du; x ^= x >> 13; x *= 0x27D4EB2Fu; x ^= x >> 16;
    return x;
}
// walker i (global index over both halves) is "accepted" at half-step s (its half's turn) iff bit 0 of the hash
__device__ __forceinline__ bool accepted(uint32_t walker, uint32_t step) { return (mix(walker, step) & 1u) != 0u; }
__device__ __forceinline__ double row_value(uint32_t walker, uint32_t stamp, int d) { return (double)walker + 1e-6 * (double)stamp + 0.125 * d; }

__device__ __forceinline__ void store_sc1(double* p, double a, double b)
{
    typedef double v2d __attribute__((ext_vector_type(2)));
    const v2d t = {a, b};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(p), "v"(t) : "memory");
}
// one row: both 16-byte pieces requested, then one wait
__device__ __forceinline__ void load_row_sc1(const double* p, double& a, double& b, double& c, double& d)
{
    typedef double v2d __attribute__((ext_vector_type(2)));
    v2d t, u;
    asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(t), "=&v"(u) : "v"(p) : "memory");
    a = t.x; b = t.y; c = u.x; d = u.y;
}

struct Args {
    double*   pos;        // [2 h][ND]
    uint32_t* flags;      // [32] one 128-byte line (form 0), or counter at [0] (form 1)
    unsigned long long* stale;   // [0] stale partner rows seen, [1] timeouts
    unsigned long long* t_cycles; // [W] s_memrealtime ticks of the loop, per workgroup
    int       nsteps;     // half-steps
    int       form;       // 0 flags, 1 counter
    int       skip_join;  // 1: no join at all (the lower bound: loads + stores only; rows are then NOT checked)
};

__global__ void join_probe(const Args a)
{
    const int W = gridDim.x, T = blockDim.x;
    const uint32_t h = (uint32_t)W * (uint32_t)T;            // walkers per half
    const uint32_t me = blockIdx.x * T + threadIdx.x;        // index inside a half
    const int lane = threadIdx.x & 63;
    // initial rows: stamp 0 (written by the host)
    uint32_t last[2] = {0u, 0u};                             // step + 1 of this thread's walkers' last accept
    unsigned long long nstale = 0;
    unsigned long long t0 = 0, t1 = 0;
    __syncthreads();
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
    for (int s = 0; s < a.nsteps; ++s) {
        const int half = s & 1;
        // join: every workgroup has completed half-step s - 1
        if (!a.skip_join && s > 0) {
            if (threadIdx.x < 64) {
                unsigned spins = 0;
                bool ok;
                do {
                    if (a.form == 0) {
                        const uint32_t f = lane < W ? __hip_atomic_load(a.flags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xffffffffu;
                        ok = __all(f >= (uint32_t)s);
                    } else {
                        const uint32_t c = __hip_atomic_load(a.flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ok = c >= (uint32_t)W * (uint32_t)s;
                    }
                    if (!ok && ++spins > 4000000u) { if (lane == 0) atomicAdd(a.stale + 1, 1ull); break; }
                } while (!ok);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __syncthreads();
        }
        // partner: uniform over the complementary half
        const uint32_t walker = (uint32_t)half * h + me;
        const uint32_t p = mix(walker ^ 0xABCD1234u, (uint32_t)s) % h;
        const uint32_t pw = (uint32_t)(1 - half) * h + p;
        double r0, r1, r2, r3;
        const double* prow = a.pos + (size_t)pw * ND;
        load_row_sc1(prow, r0, r1, r2, r3);
        if (!a.skip_join) {
            // the partner's half was last updated at steps s - 1, s - 3, ...: find its last accept
            uint32_t stamp = 0u;
            for (int q = s - 1; q >= 0; q -= 2) if (accepted(pw, (uint32_t)q)) { stamp = (uint32_t)q + 1u; break; }
            if (r0 != row_value(pw, stamp, 0) || r3 != row_value(pw, stamp, 3)) ++nstale;
        }
        if (accepted(walker, (uint32_t)s)) {
            last[half] = (uint32_t)s + 1u;
            double* own = a.pos + (size_t)walker * ND;
            asm volatile("" :: "v"(r1), "v"(r2));                     // (the whole partner row was requested: keep both pieces alive)
            store_sc1(own, row_value(walker, last[half], 0), row_value(walker, last[half], 1));
            store_sc1(own + 2, row_value(walker, last[half], 2), row_value(walker, last[half], 3));
        }
        if (!a.skip_join) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // every storing wave drains
            __syncthreads();
            if (threadIdx.x == 0) {
                if (a.form == 0) __hip_atomic_store(a.flags + blockIdx.x, (uint32_t)s + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else __hip_atomic_fetch_add(a.flags, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
    if (nstale) atomicAdd(a.stale, nstale);
    if (threadIdx.x == 0) a.t_cycles[blockIdx.x] = t1 - t0;
}

int main(int argc, char** argv)
{
    const int nsteps = argc > 1 ? std::atoi(argv[1]) : 4096;
    struct Case { int W, T; };
    const Case cases[] = {{8, 256}, {16, 128}, {32, 64}, {8, 1024}, {16, 512}, {32, 256}, {64, 128}, {16, 1024}, {32, 512}};
    std::printf("join probe: rows of %d doubles, %d half-steps per launch; us per half-step = in-kernel s_memrealtime (100 MHz) of the slowest workgroup / half-steps\n", ND, nsteps);
    std::printf("%8s %6s %6s | %12s %12s %12s | %s\n", "walkers", "W", "T", "flags", "counter", "no join", "stale rows / timeouts (flags, counter)");
    for (const Case& c : cases) {
        const size_t h = (size_t)c.W * c.T;
        double* pos; uint32_t* flags; unsigned long long* stale; unsigned long long* tc;
        CK(hipMalloc(&pos, 2 * h * ND * sizeof(double)));
        CK(hipMalloc(&flags, 4096));
        CK(hipMalloc(&stale, 64));
        CK(hipMalloc(&tc, 64 * sizeof(unsigned long long)));
        std::vector<double> init(2 * h * ND);
        for (size_t w = 0; w < 2 * h; ++w) for (int d = 0; d < ND; ++d) init[w * ND + d] = (double)w + 0.125 * d;
        double us[3] = {0, 0, 0};
        unsigned long long bad[3][2] = {{0, 0}, {0, 0}, {0, 0}};
        for (int v = 0; v < 3; ++v) {
            for (int rep = 0; rep < 2; ++rep) {                     // second launch: warm
                CK(hipMemcpy(pos, init.data(), init.size() * sizeof(double), hipMemcpyHostToDevice));
                CK(hipMemset(flags, 0, 4096));
                CK(hipMemset(stale, 0, 64));
                Args a{pos, flags, stale, tc, nsteps, v == 1 ? 1 : 0, v == 2 ? 1 : 0};
                hipLaunchKernelGGL(join_probe, dim3(c.W), dim3(c.T), 0, 0, a);
                CK(hipDeviceSynchronize());
            }
            unsigned long long t[64];
            CK(hipMemcpy(t, tc, c.W * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            unsigned long long tm = 0;
            for (int w = 0; w < c.W; ++w) tm = t[w] > tm ? t[w] : tm;
            us[v] = (double)tm * 0.01 / nsteps;
            CK(hipMemcpy(bad[v], stale, 16, hipMemcpyDeviceToHost));
        }
        std::printf("%8zu %6d %6d | %12.3f %12.3f %12.3f | %llu/%llu, %llu/%llu\n", 2 * h, c.W, c.T, us[0], us[1], us[2], bad[0][0], bad[0][1], bad[1][0], bad[1][1]);
        CK(hipFree(pos)); CK(hipFree(flags)); CK(hipFree(stale)); CK(hipFree(tc));
    }
    return 0;
}
