// Does the host memory that hipGraphExecKernelNodeSetParams keeps (~80 B per call, round 2) also go with the other ways of rewriting a graph's kernel parameters?
//   mode 0: hipGraphExecKernelNodeSetParams per node (what the updated-graph launch mode does)
//   mode 1: hipGraphKernelNodeSetParams per node on the template graph + one hipGraphExecUpdate per replay
//   mode 2: no updates (control)
//   mode 3: as mode 0 with a ring of six executables and event waits, no stream synchronisation per replay (the library's feeding loop)
// Prints RSS growth and host time per replay of a 128-node graph.   hipcc --offload-arch=gfx950 -O2 graph_update_leak.hip -o graph_update_leak && ./graph_update_leak <mode> [replays]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
struct Big { double pad[40]; };                    // (the library's kernels take a ~300-byte struct by value behind the scalars)
__global__ void bump(unsigned long long* acc, unsigned step, Big b) { if (threadIdx.x == 0 && blockIdx.x == 0) *acc += step + (unsigned)b.pad[3]; }
static long rss_kb()
{
    FILE* f = fopen("/proc/self/status", "r"); char line[256]; long kb = -1;
    while (fgets(line, sizeof line, f)) if (!strncmp(line, "VmRSS:", 6)) kb = atol(line + 6);
    fclose(f); return kb;
}
int main(int argc, char** argv)
{
    const int mode = argc > 1 ? atoi(argv[1]) : 0, replays = argc > 2 ? atoi(argv[2]) : 4000, N = 128;
    hipStream_t st; CK(hipStreamCreate(&st));
    unsigned long long* acc; CK(hipMalloc(&acc, 8)); CK(hipMemset(acc, 0, 8));
    hipGraph_t g; CK(hipGraphCreate(&g, 0));
    std::vector<hipGraphNode_t> nodes(N);
    std::vector<unsigned> steps(N, 0);
    std::vector<void*> argv2(3 * N);
    Big big{};
    for (int i = 0; i < N; ++i) {
        argv2[3 * i] = &acc; argv2[3 * i + 1] = &steps[i]; argv2[3 * i + 2] = &big;
        hipKernelNodeParams p{}; p.func = (void*)bump; p.gridDim = dim3(1); p.blockDim = dim3(64); p.kernelParams = &argv2[3 * i];
        CK(hipGraphAddKernelNode(&nodes[i], g, i ? &nodes[i - 1] : nullptr, i ? 1 : 0, &p));
    }
    hipGraphExec_t exs[6]; hipEvent_t done[6]; bool inflight[6] = {};
    for (int k = 0; k < 6; ++k) { CK(hipGraphInstantiate(&exs[k], g, nullptr, nullptr, 0)); CK(hipEventCreateWithFlags(&done[k], hipEventDisableTiming)); }
    unsigned long long want = 0;
    auto replay = [&](int r) {
        const int k = mode == 3 ? r % 6 : 0;
        hipGraphExec_t ex = exs[k];
        if (inflight[k]) { CK(hipEventSynchronize(done[k])); inflight[k] = false; }
        for (int i = 0; i < N; ++i) {
            steps[i] = mode == 2 ? 0u : (unsigned)(r * N + i);
            want += steps[i];
            if (mode == 2) continue;
            hipKernelNodeParams p{}; p.func = (void*)bump; p.gridDim = dim3(1); p.blockDim = dim3(64); p.kernelParams = &argv2[3 * i];
            if (mode == 0 || mode == 3) CK(hipGraphExecKernelNodeSetParams(ex, nodes[i], &p));
            else CK(hipGraphKernelNodeSetParams(nodes[i], &p));
        }
        if (mode == 1) {
            hipGraphNode_t bad = nullptr; hipGraphExecUpdateResult res;
            CK(hipGraphExecUpdate(ex, g, &bad, &res));
            if (res != hipGraphExecUpdateSuccess) { fprintf(stderr, "update result %d\n", (int)res); exit(1); }
        }
        CK(hipGraphLaunch(ex, st));
        if (mode == 3) { CK(hipEventRecord(done[k], st)); inflight[k] = true; }
        else CK(hipStreamSynchronize(st));       // (one executable: the parameters may not change under a replay in flight)
    };
    for (int r = 0; r < 200; ++r) replay(r);
    const long rss0 = rss_kb();
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 200; r < 200 + replays; ++r) replay(r);
    CK(hipStreamSynchronize(st));
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / replays;
    const long rss1 = rss_kb();
    unsigned long long got; CK(hipMemcpy(&got, acc, 8, hipMemcpyDeviceToHost));
    printf("mode %d: %d replays x %d nodes: RSS %+ld KiB (%.1f B per node update), %.1f us per replay (updates + launch + sync), sum %s\n", mode, replays, N, rss1 - rss0,
           (rss1 - rss0) * 1024.0 / ((double)replays * N), us, got == want ? "ok" : "WRONG");
    return got == want ? 0 : 1;
}
