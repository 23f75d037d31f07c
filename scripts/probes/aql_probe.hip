// aql_probe.hip -- what does the dependent-launch boundary cost when the dispatch packets are written by hand?  (round 4, bounded experiment)
//
// Every launch-bound shape of this library pays 1.2-1.35 us between two dependent kernels (profiles/r04_probe_timeline.txt), through the HIP
// runtime's graphs.  The runtime decides the packets' acquire / release fence scopes (cache invalidate before, write-back after) and which
// queue they go to.  This probe submits the SAME half-step kernel (scripts/probes/aql_kernels.hip; 4 096 ... 65 536 walkers x 4 doubles, one
// walker per lane) as AQL kernel-dispatch packets written straight into an HSA queue of its own -- barrier bit set, one doorbell for the whole
// series, the step baked into each packet's own kernarg block -- with
//     fences AGENT / AGENT    (what a dependent launch needs with ordinary loads and stores)
//     fences SYSTEM / SYSTEM
//     fences NONE / NONE      with the kernel whose row loads are `sc1` and whose stores are write-through `sc0 sc1` (no cache maintenance
//                             needed at the boundary: the hand-off MI355X_MICROARCH.md lists as valid inside one launch)
// and compares the time per half-step with a hipGraph replay of the same launches; every series' final state is compared with the graph's.
//
// Build + run (GPU box):
//   hipcc --offload-device-only --no-gpu-bundle-output --offload-arch=gfx950 -O3 -ffp-contract=off -I kissmcmc.jl_amd/csrc scripts/probes/aql_kernels.hip -o gpurun_out/aql_kernels.hsaco
//   hipcc -O2 --offload-arch=gfx950 -I kissmcmc.jl_amd/csrc scripts/probes/aql_probe.hip -lhsa-runtime64 -o gpurun_out/aql_probe && gpurun_out/aql_probe gpurun_out/aql_kernels.hsaco
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "kmc_device.hpp"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
#define HK(x) do { hsa_status_t e_ = (x); if (e_ != HSA_STATUS_SUCCESS) { const char* m_ = nullptr; hsa_status_string(e_, &m_); std::fprintf(stderr, "%s: %s\n", #x, m_ ? m_ : "?"); std::exit(1); } } while (0)

using namespace kmc;

struct Args {            // = aql_kernels.hip
    double*       pos;
    double*       logp;
    uint32_t*     nacc;
    DrawConsts    dc;
    DensityParams dp;
    uint32_t      h;
    uint32_t      step;
};
static_assert(sizeof(Args) == 128, "kernarg layout of aql_kernels.hip");
constexpr int ND = 4;

static hsa_agent_t g_gpu;
static bool g_have_gpu = false;
static hsa_status_t find_gpu(hsa_agent_t a, void*)
{
    hsa_device_type_t t;
    if (hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t) == HSA_STATUS_SUCCESS && t == HSA_DEVICE_TYPE_GPU && !g_have_gpu) { g_gpu = a; g_have_gpu = true; }
    return HSA_STATUS_SUCCESS;
}

struct Kernel { uint64_t object; uint32_t kernarg, group, priv; };
static Kernel symbol(hsa_executable_t exe, const char* name)
{
    hsa_executable_symbol_t s;
    HK(hsa_executable_get_symbol_by_name(exe, (std::string(name) + ".kd").c_str(), &g_gpu, &s));
    Kernel k{};
    HK(hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &k.object));
    HK(hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &k.kernarg));
    HK(hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &k.group));
    HK(hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &k.priv));
    return k;
}

int main(int argc, char** argv)
{
    if (argc < 2) { std::fprintf(stderr, "usage: aql_probe <aql_kernels.hsaco>\n"); return 2; }
    std::ifstream f(argv[1], std::ios::binary);
    std::vector<char> co((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    if (co.empty()) { std::fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
    CK(hipSetDevice(0));
    CK(hipFree(nullptr));
    HK(hsa_init());
    HK(hsa_iterate_agents(find_gpu, nullptr));
    if (!g_have_gpu) { std::fprintf(stderr, "no GPU agent\n"); return 2; }
    // the code object, loaded through HSA (for the packets) and through HIP (for the graph this is compared with)
    hsa_code_object_reader_t reader;
    HK(hsa_code_object_reader_create_from_memory(co.data(), co.size(), &reader));
    hsa_executable_t exe;
    HK(hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &exe));
    HK(hsa_executable_load_agent_code_object(exe, g_gpu, reader, nullptr, nullptr));
    HK(hsa_executable_freeze(exe, nullptr));
    const Kernel k_plain = symbol(exe, "half_step_plain"), k_sc = symbol(exe, "half_step_sc");
    if (k_plain.kernarg != sizeof(Args) || k_plain.group != 0 || k_plain.priv != 0) { std::fprintf(stderr, "unexpected kernel resources\n"); return 2; }
    hipModule_t mod;
    CK(hipModuleLoadData(&mod, co.data()));
    hipFunction_t fn_plain;
    CK(hipModuleGetFunction(&fn_plain, mod, "half_step_plain"));
    hsa_queue_t* q = nullptr;
    const uint32_t QSIZE = 16384;
    HK(hsa_queue_create(g_gpu, QSIZE, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &q));
    hsa_signal_t done;
    HK(hsa_signal_create(1, 0, nullptr, &done));
    // a second queue for the one-XCD experiment: which bits of a CU mask are which XCD?  (where_am_i reports each workgroup's XCC_ID)
    hsa_queue_t* q1 = nullptr;
    HK(hsa_queue_create(g_gpu, QSIZE, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &q1));
    const Kernel k_where = symbol(exe, "where_am_i");
    uint32_t* d_where = nullptr;
    CK(hipMalloc(&d_where, 4096 * 4));
    uint64_t* d_warg = nullptr;
    CK(hipMalloc(&d_warg, 64));
    { const uint64_t pw = reinterpret_cast<uint64_t>(d_where); CK(hipMemcpy(d_warg, &pw, 8, hipMemcpyHostToDevice)); }
    auto where = [&](const char* what, const uint32_t (&mask)[8]) -> int {
        HK(hsa_amd_queue_cu_set_mask(q1, 256, mask));
        CK(hipMemset(d_where, 0xff, 4096 * 4)); CK(hipDeviceSynchronize());
        hsa_signal_store_relaxed(done, 1);
        const uint64_t idx = hsa_queue_add_write_index_relaxed(q1, 1);
        hsa_kernel_dispatch_packet_t* p = reinterpret_cast<hsa_kernel_dispatch_packet_t*>(q1->base_address) + (idx & (QSIZE - 1));
        p->setup = 1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
        p->workgroup_size_x = 64; p->workgroup_size_y = 1; p->workgroup_size_z = 1; p->reserved0 = 0;
        p->grid_size_x = 2048 * 64; p->grid_size_y = 1; p->grid_size_z = 1;
        p->private_segment_size = k_where.priv; p->group_segment_size = k_where.group;
        p->kernel_object = k_where.object; p->kernarg_address = d_warg; p->reserved2 = 0; p->completion_signal = done;
        const uint16_t header = (uint16_t)((HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) |
                                           (HSA_FENCE_SCOPE_SYSTEM << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (HSA_FENCE_SCOPE_SYSTEM << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE));
        __atomic_store_n(reinterpret_cast<uint16_t*>(p), header, __ATOMIC_RELEASE);
        hsa_signal_store_screlease(q1->doorbell_signal, (hsa_signal_value_t)idx);
        const auto t0 = std::chrono::steady_clock::now();
        while (hsa_signal_wait_scacquire(done, HSA_SIGNAL_CONDITION_LT, 1, 2000000000ull, HSA_WAIT_STATE_ACTIVE) != 0)
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(20)) { std::fprintf(stderr, "where_am_i timed out\n"); std::exit(3); }
        std::vector<uint32_t> w(2048);
        CK(hipMemcpy(w.data(), d_where, 2048 * 4, hipMemcpyDeviceToHost));
        int hist[16] = {0}, used = 0;
        for (uint32_t v : w) if (v != 0xffffffffu) ++hist[v & 0xf];
        std::string hs;
        for (int x = 0; x < 16; ++x) if (hist[x]) { ++used; hs += " xcc" + std::to_string(x) + ":" + std::to_string(hist[x]); }
        std::printf("CU mask %-44s -> workgroups by XCC_ID:%s\n", what, hs.c_str());
        return used;
    };
    uint32_t m_all[8], m_first32[8] = {0xffffffffu, 0, 0, 0, 0, 0, 0, 0}, m_every8[8], m_low4each[8];
    for (int i = 0; i < 8; ++i) { m_all[i] = 0xffffffffu; m_every8[i] = 0x01010101u; m_low4each[i] = 0x0000000fu; }
    where("all 256 bits", m_all);
    const int used_first = where("bits 0..31", m_first32);
    const int used_every8 = where("every 8th bit (32 bits)", m_every8);
    where("bits 0..3 of every word (32 bits)", m_low4each);
    const uint32_t* one_xcd = used_first == 1 ? m_first32 : (used_every8 == 1 ? m_every8 : nullptr);
    std::printf("one-XCD mask: %s\n", one_xcd == m_first32 ? "bits 0..31" : one_xcd == m_every8 ? "every 8th bit" : "none of the patterns tried");
    if (one_xcd) HK(hsa_amd_queue_cu_set_mask(q1, 256, one_xcd));
    hsa_queue_t* Q = q;                                     // the queue `series` submits to
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));

    const int G = 64, REPS = 60, N = 2 * G * REPS;          // half-steps per series (< QSIZE: every packet is written before the one doorbell)
    std::printf("aql_probe: Gaussian x %d doubles, one walker per lane, %d dependent half-step launches per series; us per half-step\n", ND, N);
    std::printf("%8s | %10s | %14s %14s %14s %16s | final state against the graph's (AGENT, SYSTEM, NONE+sc, NONE+plain)\n", "walkers", "hipGraph", "AQL AGENT", "AQL SYSTEM", "AQL NONE + sc", "AQL NONE + plain");
    for (uint32_t nw : {4096u, 16384u, 65536u}) {
        const uint32_t h = nw / 2, nb = (h + 63) / 64;
        std::vector<double> pos0((size_t)nw * ND), lp0(nw);
        uint64_t s = 999;
        for (auto& v : pos0) { s = s * 6364136223846793005ull + 1442695040888963407ull; v = ((double)(s >> 11) * 0x1.0p-53 - 0.5) * 4.0; }
        for (uint32_t w = 0; w < nw; ++w) { double qq = 0; for (int d = 0; d < ND; ++d) qq += pos0[(size_t)w * ND + d] * pos0[(size_t)w * ND + d]; lp0[w] = -0.5 * qq; }
        double *pos, *logp; uint32_t* nacc; Args* kargs;
        CK(hipMalloc(&pos, pos0.size() * 8)); CK(hipMalloc(&logp, nw * 8)); CK(hipMalloc(&nacc, nw * 4)); CK(hipMalloc(&kargs, (size_t)N * sizeof(Args)));
        auto reset = [&] {
            CK(hipMemcpy(pos, pos0.data(), pos0.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(logp, lp0.data(), nw * 8, hipMemcpyHostToDevice));
            CK(hipMemset(nacc, 0, nw * 4)); CK(hipDeviceSynchronize());
        };
        Args base{};
        base.pos = pos; base.logp = logp; base.nacc = nacc;
        base.dc.seed_lo = 5u; base.dc.seed_hi = 9u; base.dc.nhalf = h;
        base.dc.c0 = std::sqrt(0.5); base.dc.c1 = std::sqrt(2.0) - std::sqrt(0.5); base.dc.nm1 = (double)(ND - 1);
        base.dp.p[0] = 0.0; base.dp.p[1] = 1.0; base.dp.ndim = ND;
        base.h = h;
        std::vector<Args> hk((size_t)N, base);
        for (int i = 0; i < N; ++i) hk[(size_t)i].step = (uint32_t)i;
        CK(hipMemcpy(kargs, hk.data(), hk.size() * sizeof(Args), hipMemcpyHostToDevice));
        // (a) the HIP graph: 128 kernel nodes per replay, each with its own step; REPS graphs would be needed for different steps per
        // replay -- instead ONE graph over all N launches (instantiated once; what matters here is the GPU-side period)
        reset();
        hipGraph_t graph; hipGraphExec_t gexec;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
        for (int i = 0; i < N; ++i) {
            Args a = hk[(size_t)i];
            void* params[] = {&a};
            CK(hipModuleLaunchKernel(fn_plain, 2 * 0 + nb, 1, 1, 64, 1, 1, 0, st, params, nullptr));
        }
        CK(hipStreamEndCapture(st, &graph));
        CK(hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0));
        CK(hipGraphLaunch(gexec, st)); CK(hipStreamSynchronize(st));          // warm
        reset();
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(gexec, st)); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us_graph = ms * 1e3 / N;
        std::vector<double> ref_pos(pos0.size()), ref_lp(nw); std::vector<uint32_t> ref_n(nw);
        CK(hipMemcpy(ref_pos.data(), pos, ref_pos.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(ref_lp.data(), logp, nw * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(ref_n.data(), nacc, nw * 4, hipMemcpyDeviceToHost));
        CK(hipGraphExecDestroy(gexec)); CK(hipGraphDestroy(graph));
        // (b) the same launches as hand-written packets
        auto series = [&](const Kernel& k, hsa_fence_scope_t acq, hsa_fence_scope_t rel, double* us, std::string* verdict, int mode = 0) {
            for (auto& a : hk) a.dp.pad_ = mode;
            CK(hipMemcpy(kargs, hk.data(), hk.size() * sizeof(Args), hipMemcpyHostToDevice));
            reset();
            double best = 1e30;
            for (int rep = 0; rep < 2; ++rep) {                       // first pass warm-up (state continues: compare after a fresh third pass below)
                if (rep == 1) reset();
                hsa_signal_store_relaxed(done, 1);
                const uint64_t first = hsa_queue_add_write_index_relaxed(Q, (uint64_t)N);
                // (the queue is idle between series, so N < QSIZE slots are free)
                for (int i = 0; i < N; ++i) {
                    hsa_kernel_dispatch_packet_t* p = reinterpret_cast<hsa_kernel_dispatch_packet_t*>(Q->base_address) + ((first + (uint64_t)i) & (QSIZE - 1));
                    p->setup = 1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
                    p->workgroup_size_x = 64; p->workgroup_size_y = 1; p->workgroup_size_z = 1;
                    p->reserved0 = 0;
                    p->grid_size_x = nb * 64; p->grid_size_y = 1; p->grid_size_z = 1;
                    p->private_segment_size = k.priv; p->group_segment_size = k.group;
                    p->kernel_object = k.object;
                    p->kernarg_address = kargs + i;
                    p->reserved2 = 0;
                    p->completion_signal.handle = i == N - 1 ? done.handle : 0;
                    const bool edge = i == 0 || i == N - 1;            // the series' ends see / publish memory at system scope
                    const uint16_t header = (uint16_t)((HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) |
                                                       ((i == 0 ? HSA_FENCE_SCOPE_SYSTEM : acq) << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) |
                                                       ((i == N - 1 ? HSA_FENCE_SCOPE_SYSTEM : rel) << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE));
                    (void)edge;
                    __atomic_store_n(reinterpret_cast<uint16_t*>(p), header, __ATOMIC_RELEASE);
                }
                const auto t0 = std::chrono::steady_clock::now();
                hsa_signal_store_screlease(Q->doorbell_signal, (hsa_signal_value_t)(first + (uint64_t)N - 1));
                while (hsa_signal_wait_scacquire(done, HSA_SIGNAL_CONDITION_LT, 1, 2000000000ull, HSA_WAIT_STATE_ACTIVE) != 0) {
                    if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(20)) { std::fprintf(stderr, "series timed out\n"); std::exit(3); }
                }
                const double us1 = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
                if (rep == 1) best = us1;
            }
            *us = best;
            std::vector<double> gp(pos0.size()), gl(nw); std::vector<uint32_t> gn(nw);
            CK(hipMemcpy(gp.data(), pos, gp.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(gl.data(), logp, nw * 8, hipMemcpyDeviceToHost));
            CK(hipMemcpy(gn.data(), nacc, nw * 4, hipMemcpyDeviceToHost));
            size_t diff = 0;
            for (uint32_t w = 0; w < nw; ++w) if (std::memcmp(&gp[(size_t)w * ND], &ref_pos[(size_t)w * ND], ND * 8) != 0 || gn[w] != ref_n[w]) ++diff;
            *verdict = diff == 0 ? "same" : std::to_string(diff) + " walkers differ";
        };
        double u1, u2, u3, u4; std::string v1, v2, v3, v4;
        series(k_plain, HSA_FENCE_SCOPE_AGENT, HSA_FENCE_SCOPE_AGENT, &u1, &v1);
        series(k_plain, HSA_FENCE_SCOPE_SYSTEM, HSA_FENCE_SCOPE_SYSTEM, &u2, &v2);
        series(k_sc, HSA_FENCE_SCOPE_NONE, HSA_FENCE_SCOPE_NONE, &u3, &v3);
        series(k_plain, HSA_FENCE_SCOPE_NONE, HSA_FENCE_SCOPE_NONE, &u4, &v4);
        std::printf("%8u | %10.3f | %14.3f %14.3f %14.3f %16.3f | %s, %s, %s, %s\n", nw, us_graph, u1, u2, u3, u4, v1.c_str(), v2.c_str(), v3.c_str(), v4.c_str());
        struct V { const char* what; hsa_fence_scope_t acq, rel; int mode; };
        const V more[] = {
            {"acquire AGENT, release NONE, write-through stores", HSA_FENCE_SCOPE_AGENT, HSA_FENCE_SCOPE_NONE, 2},
            {"acquire AGENT, release NONE, write-through stores, every wave waits for its stores' acknowledgements", HSA_FENCE_SCOPE_AGENT, HSA_FENCE_SCOPE_NONE, 2 | 32},
            {"acquire NONE, release AGENT, buffer_inv sc1 at wave entry", HSA_FENCE_SCOPE_NONE, HSA_FENCE_SCOPE_AGENT, 1},
            {"NONE / NONE, buffer_inv sc1 at entry + write-through stores", HSA_FENCE_SCOPE_NONE, HSA_FENCE_SCOPE_NONE, 3},
            {"NONE / NONE, buffer_inv sc1 at entry + buffer_wbl2 sc1 at the end", HSA_FENCE_SCOPE_NONE, HSA_FENCE_SCOPE_NONE, 5},
            {"NONE / NONE, sc1 loads + buffer_wbl2 sc1 at the end", HSA_FENCE_SCOPE_NONE, HSA_FENCE_SCOPE_NONE, 12},
            {"AGENT / AGENT, write-through stores (what the product's kernels do)", HSA_FENCE_SCOPE_AGENT, HSA_FENCE_SCOPE_AGENT, 2},
        };
        for (const V& v : more) {
            double u; std::string vd;
            series(k_plain, v.acq, v.rel, &u, &vd, v.mode);
            std::printf("         |            | %8.3f  %s: %s\n", u, v.what, vd.c_str());
        }
        if (one_xcd) {
            Q = q1;
            const V xs[] = {
                {"one XCD (CU mask): AGENT / AGENT, ordinary loads and stores", HSA_FENCE_SCOPE_AGENT, HSA_FENCE_SCOPE_AGENT, 0},
                {"one XCD (CU mask): NONE / NONE, sc0 loads (L1 bypassed, the XCD's L2 answers), ordinary stores", HSA_FENCE_SCOPE_NONE, HSA_FENCE_SCOPE_NONE, 16},
                {"one XCD (CU mask): NONE / NONE, ordinary loads and stores", HSA_FENCE_SCOPE_NONE, HSA_FENCE_SCOPE_NONE, 0},
            };
            for (const V& v : xs) {
                double u; std::string vd;
                series(k_plain, v.acq, v.rel, &u, &vd, v.mode);
                std::printf("         |            | %8.3f  %s: %s\n", u, v.what, vd.c_str());
            }
            Q = q;
        }
        std::fflush(stdout);
        CK(hipFree(pos)); CK(hipFree(logp)); CK(hipFree(nacc)); CK(hipFree(kargs));
    }
    HK(hsa_signal_destroy(done));
    HK(hsa_queue_destroy(q));
    return 0;
}
