// Probe: two processes on ONE GPU exchange IPC handles for (a) a plain hipMalloc buffer and
// (b) a fine-grained flag buffer; ping-pong through kernels that poll a local flag and write the
// peer's flag.  Tells us whether the p2p exchange protocol works on this stack and what a
// signal->observe hop costs.   build: hipcc --offload-arch=gfx950 -O2 ipc_probe.hip -o ipc_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>
#include <sys/wait.h>
#include <chrono>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "[%d] %s -> %s\n", getpid(), #x, hipGetErrorString(e)); exit(2); } } while (0)

__global__ void fill(double* p, int n, double v) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = v + i; }

__global__ void signal_peer(unsigned long long* peer_flag, unsigned long long value)
{
    __threadfence_system();
    __hip_atomic_store(peer_flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// wait until *local_flag >= value (bounded), then sum the peer's data
__global__ void wait_and_read(unsigned long long* local_flag, unsigned long long value, const double* peer_data, int n,
                              double* out, unsigned long long* spins_out)
{
    unsigned long long spins = 0;
    if (threadIdx.x == 0) {
        while (__hip_atomic_load(local_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < value) {
            __builtin_amdgcn_s_sleep(8);
            if (++spins > 50000000ull) break;
        }
        *spins_out = spins;
    }
    __syncthreads();
    double s = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += peer_data[i];
    atomicAdd(out, s);
}

struct Handles { hipIpcMemHandle_t data, flag; };

int main(int argc, char** argv)
{
    int finegrained = argc > 1 ? atoi(argv[1]) : 1;
    int ab[2], ba[2];
    if (pipe(ab) || pipe(ba)) return 1;
    pid_t pid = fork();
    const int me = pid == 0 ? 1 : 0;
    int rd = me == 0 ? ba[0] : ab[0], wr = me == 0 ? ab[1] : ba[1];
    CK(hipSetDevice(0));
    const int n = 1 << 17;
    double* data; unsigned long long* flag;
    CK(hipMalloc(&data, n * sizeof(double)));
    if (finegrained) CK(hipExtMallocWithFlags((void**)&flag, 4096, hipDeviceMallocFinegrained));
    else CK(hipMalloc((void**)&flag, 4096));
    CK(hipMemset(flag, 0, 4096));
    CK(hipMemset(data, 0, n * sizeof(double)));
    CK(hipDeviceSynchronize());
    Handles mine, peer;
    CK(hipIpcGetMemHandle(&mine.data, data));
    CK(hipIpcGetMemHandle(&mine.flag, flag));
    if (write(wr, &mine, sizeof(mine)) != sizeof(mine)) return 3;
    if (read(rd, &peer, sizeof(peer)) != sizeof(peer)) return 3;
    double* pdata; unsigned long long* pflag;
    CK(hipIpcOpenMemHandle((void**)&pdata, peer.data, hipIpcMemLazyEnablePeerAccess));
    CK(hipIpcOpenMemHandle((void**)&pflag, peer.flag, hipIpcMemLazyEnablePeerAccess));
    double* out; unsigned long long* spins;
    CK(hipMalloc(&out, 8)); CK(hipMalloc(&spins, 8));
    hipStream_t st; CK(hipStreamCreate(&st));
    const int rounds = 200;
    auto t0 = std::chrono::steady_clock::now();
    double bad = 0; unsigned long long max_spins = 0;
    for (int r = 1; r <= rounds; ++r) {
        // round r: rank (r&1) produces, the other consumes
        if ((r & 1) == me) {
            fill<<<n / 256, 256, 0, st>>>(data, n, (double)r);
            signal_peer<<<1, 1, 0, st>>>(pflag, (unsigned long long)r);
        } else {
            CK(hipMemsetAsync(out, 0, 8, st));
            wait_and_read<<<1, 256, 0, st>>>(flag, (unsigned long long)r, pdata, n, out, spins);
            double h; unsigned long long sp;
            CK(hipMemcpyAsync(&h, out, 8, hipMemcpyDeviceToHost, st));
            CK(hipMemcpyAsync(&sp, spins, 8, hipMemcpyDeviceToHost, st));
            CK(hipStreamSynchronize(st));
            double want = (double)r * n + (double)n * (n - 1) / 2;
            if (h != want) bad += 1;
            if (sp > max_spins) max_spins = sp;
        }
    }
    CK(hipStreamSynchronize(st));
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    printf("[rank %d] finegrained=%d rounds=%d wrong=%g max_spins=%llu  %.1f us/round\n", me, finegrained, rounds, bad, max_spins, us / rounds);
    CK(hipIpcCloseMemHandle(pdata)); CK(hipIpcCloseMemHandle(pflag));
    if (me == 0) { int stt; waitpid(pid, &stt, 0); return WEXITSTATUS(stt); }
    return bad != 0;
}
