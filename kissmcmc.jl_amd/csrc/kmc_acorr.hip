// kmc_acorr.hip -- integrated autocorrelation time (kmc_int_acorr; the device part of kmc_sampler_int_acorr).
#include <cmath>
#include <cstdlib>
#include <dlfcn.h>
#include <mutex>
#include <vector>

#include "kmc_host.hpp"
#include <hipfft/hipfft.h>

using namespace kmc_host;

// ------------------------------------------------------------------------------------------
// Integrated autocorrelation time: int_acorr / acor1d / auto_window of reference src/analysis.jl:140-167,
// :252-273, :280-285.  That file is 100 % commented out in the reference -- there is no live behaviour
// to match -- so this follows the code as written: per dimension, the autocorrelation function of every
// walker's chain by FFT WITHOUT zero padding (circular, :258-260), normalised by its lag-0 value (:264), first
// half kept (:267), averaged over the walkers (:149-151), tau(M) = 2 sum_{l<=M} rho_l - 1 (:153) at the first
// window M >= c tau(M) (:280-285).  FFTs by hipFFT (batched D2Z / Z2D over all walker x dimension series at
// once, resolved with dlopen so the samplers do not depend on it); centring, power spectrum and the average
// over walkers are kernels here; the final scan over <= nsamples/2 lags per dimension runs on the host.
// ------------------------------------------------------------------------------------------

namespace {

__global__ __launch_bounds__(256) void acorr_center(const double* chain, double* y, int64_t nsamples, int64_t batch)
{
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;          // series = (walker, dimension)
    if (b >= batch) return;
    double m = 0.0;
    for (int64_t t = 0; t < nsamples; ++t) m += chain[t * batch + b];
    m /= (double)nsamples;                                               // :258 x - mean(x)
    for (int64_t t = 0; t < nsamples; ++t) y[t * batch + b] = chain[t * batch + b] - m;
}

__global__ __launch_bounds__(256) void acorr_power(double2* z, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double2 v = z[i];
    z[i] = make_double2(v.x * v.x + v.y * v.y, 0.0);                     // :259 f .* conj(f)
}

// rho[d][l] = mean over walkers of acf_w,d[l] / acf_w,d[0], l < nsamples / 2                    (:149-151, :264, :267)
__global__ __launch_bounds__(256) void acorr_rho(const double* acf, double* rho, int64_t nlag, int64_t nwalkers, int64_t ndim)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;          // i = l * ndim + d
    if (i >= nlag * ndim) return;
    const int64_t l = i / ndim, d = i - l * ndim, batch = nwalkers * ndim;
    double s = 0.0;
    for (int64_t w = 0; w < nwalkers; ++w) s += acf[l * batch + w * ndim + d] / acf[w * ndim + d];
    rho[d * nlag + l] = s / (double)nwalkers;
}

struct HipfftApi {
    void* lib = nullptr;
    decltype(&hipfftPlanMany) plan_many = nullptr;
    decltype(&hipfftExecD2Z) exec_d2z = nullptr;
    decltype(&hipfftExecZ2D) exec_z2d = nullptr;
    decltype(&hipfftDestroy) destroy = nullptr;
    decltype(&hipfftSetStream) set_stream = nullptr;
};

const HipfftApi* hipfft_api()
{
    static HipfftApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"libhipfft.so", "libhipfft.so.0", "/opt/rocm/lib/libhipfft.so"}) {
            api.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (api.lib) break;
        }
        if (!api.lib) return;
        api.plan_many = reinterpret_cast<decltype(api.plan_many)>(dlsym(api.lib, "hipfftPlanMany"));
        api.exec_d2z = reinterpret_cast<decltype(api.exec_d2z)>(dlsym(api.lib, "hipfftExecD2Z"));
        api.exec_z2d = reinterpret_cast<decltype(api.exec_z2d)>(dlsym(api.lib, "hipfftExecZ2D"));
        api.destroy = reinterpret_cast<decltype(api.destroy)>(dlsym(api.lib, "hipfftDestroy"));
        api.set_stream = reinterpret_cast<decltype(api.set_stream)>(dlsym(api.lib, "hipfftSetStream"));
    });
    return (api.lib && api.plan_many && api.exec_d2z && api.exec_z2d && api.destroy && api.set_stream) ? &api : nullptr;
}

struct AcorrBuffers {
    double *chain = nullptr, *y = nullptr, *rho = nullptr;
    double2* z = nullptr;
    hipfftHandle fwd = nullptr, inv = nullptr;
    const HipfftApi* api = nullptr;
    ~AcorrBuffers()
    {
        if (api && fwd) (void)api->destroy(fwd);
        if (api && inv) (void)api->destroy(inv);
        (void)hipFree(chain); (void)hipFree(y); (void)hipFree(rho); (void)hipFree(z);
    }
};

}  // namespace

namespace kmc_host {

// chain_dev: [nsamples][nwalkers][ndim] on the current device
kmc_status int_acorr_device(const double* chain_dev, int64_t nsamples, int64_t nwalkers, int64_t ndim, double c,
                            double* tau, double* converged)
{
    const int64_t batch = nwalkers * ndim, nlag = nsamples / 2, nfreq = nsamples / 2 + 1;
    ScopedStream ss;                              // never the legacy stream (kmc_host.hpp: copy_sync)
    HIP_TRY(ss.create());
    const hipStream_t st = ss.st;
    AcorrBuffers b;
    b.api = hipfft_api();
    if (!b.api) return fail(KMC_ERR_UNSUPPORTED, "libhipfft.so could not be loaded");
    const size_t nreal = (size_t)nsamples * (size_t)batch;
    HIP_TRY(hipMalloc(&b.y, nreal * sizeof(double)));
    HIP_TRY(hipMalloc((void**)&b.z, (size_t)nfreq * (size_t)batch * sizeof(double2)));
    HIP_TRY(hipMalloc(&b.rho, (size_t)nlag * (size_t)ndim * sizeof(double)));
    // series b = (walker, dimension) is element b of every sample's [nwalkers][ndim] block: stride batch, distance 1
    int n[1] = {(int)nsamples}, inembed[1] = {(int)nsamples}, onembed[1] = {(int)nfreq};
    if (b.api->plan_many(&b.fwd, 1, n, inembed, (int)batch, 1, onembed, (int)batch, 1, HIPFFT_D2Z, (int)batch) != HIPFFT_SUCCESS ||
        b.api->plan_many(&b.inv, 1, n, onembed, (int)batch, 1, inembed, (int)batch, 1, HIPFFT_Z2D, (int)batch) != HIPFFT_SUCCESS)
        return fail(KMC_ERR_HIP, "hipfftPlanMany failed");
    if (b.api->set_stream(b.fwd, st) != HIPFFT_SUCCESS || b.api->set_stream(b.inv, st) != HIPFFT_SUCCESS) return fail(KMC_ERR_HIP, "hipfftSetStream failed");
    hipLaunchKernelGGL(acorr_center, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, st, chain_dev, b.y, nsamples, batch);
    HIP_TRY(hipGetLastError());
    if (b.api->exec_d2z(b.fwd, b.y, reinterpret_cast<hipfftDoubleComplex*>(b.z)) != HIPFFT_SUCCESS) return fail(KMC_ERR_HIP, "hipfftExecD2Z failed");
    const int64_t nz = nfreq * batch;
    hipLaunchKernelGGL(acorr_power, dim3((unsigned)((nz + 255) / 256)), dim3(256), 0, st, b.z, nz);
    HIP_TRY(hipGetLastError());
    if (b.api->exec_z2d(b.inv, reinterpret_cast<hipfftDoubleComplex*>(b.z), b.y) != HIPFFT_SUCCESS) return fail(KMC_ERR_HIP, "hipfftExecZ2D failed");
    hipLaunchKernelGGL(acorr_rho, dim3((unsigned)((nlag * ndim + 255) / 256)), dim3(256), 0, st, b.y, b.rho, nlag, nwalkers, ndim);
    HIP_TRY(hipGetLastError());
    std::vector<double> rho((size_t)nlag * (size_t)ndim);
    HIP_TRY(copy_sync(rho.data(), b.rho, rho.size() * sizeof(double), hipMemcpyDeviceToHost, st));
    bool bad = false;
    for (int64_t d = 0; d < ndim; ++d) {
        const double* r = rho.data() + d * nlag;
        double cum = 0.0;
        int64_t window = nlag - 2;                                       // :284 length(taus)-1 (1-based)
        std::vector<double> taus((size_t)nlag);
        for (int64_t i = 0; i < nlag; ++i) { cum += r[i]; taus[(size_t)i] = 2.0 * cum - 1.0; }   // :153
        for (int64_t i = 0; i < nlag; ++i)
            if ((double)(i + 1) >= c * taus[(size_t)i]) { window = i; break; }                   // :281-283
        if (window < 0) window = 0;
        const double t = taus[(size_t)window];                           // :155
        tau[d] = t;
        converged[d] = (double)nsamples / t;                             // :157
        if (t != t || converged[d] != converged[d]) bad = true;
    }
    if (bad) for (int64_t d = 0; d < ndim; ++d) { tau[d] = -1.0; converged[d] = -1.0; }          // :161-165
    return KMC_OK;
}

kmc_status int_acorr_check(int64_t nsamples, int64_t nwalkers, int64_t ndim, double c, const double* tau, const double* converged)
{
    if (!tau || !converged) return fail(KMC_ERR_BAD_ARG, "null argument");
    if (!(c > 1.0)) return fail(KMC_ERR_BAD_ARG, "c>1");                                      // :141 @assert c>1
    if (nsamples < 4 || nwalkers <= 0 || ndim <= 0) return fail(KMC_ERR_BAD_ARG, "need nsamples >= 4, nwalkers, ndim > 0");
    if (nsamples >= ((int64_t)1 << 31) || nwalkers * ndim >= ((int64_t)1 << 31)) return fail(KMC_ERR_UNSUPPORTED, "chain too large");
    return KMC_OK;
}

}  // namespace kmc_host

KMC_EXPORT kmc_status kmc_int_acorr(const double* chain_host, int64_t nsamples, int64_t nwalkers, int64_t ndim, double c,
                                    int device, double* tau, double* converged)
{
    if (!chain_host) return fail(KMC_ERR_BAD_ARG, "null argument");
    KMC_TRY(int_acorr_check(nsamples, nwalkers, ndim, c, tau, converged));
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        (void)hipGetLastError();
        return fail(KMC_ERR_NO_DEVICE, "no HIP device visible");
    }
    if (device < 0 || device >= ndev) return fail(KMC_ERR_BAD_ARG, "device ordinal out of range");
    HIP_TRY(hipSetDevice(device));
    AcorrBuffers b;
    const size_t nreal = (size_t)nsamples * (size_t)nwalkers * (size_t)ndim;
    HIP_TRY(hipMalloc(&b.chain, nreal * sizeof(double)));
    {
        ScopedStream up;
        HIP_TRY(up.create());
        HIP_TRY(copy_sync(b.chain, chain_host, nreal * sizeof(double), hipMemcpyHostToDevice, up.st));
    }
    return int_acorr_device(b.chain, nsamples, nwalkers, ndim, c, tau, converged);
}

