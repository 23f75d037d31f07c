// kmc_metropolis_api.hip -- host side of the many-chain Metropolis entry points of include/kissmcmc_hip.h
// (kmc_metropolis_validate / kmc_metropolis_run; kernels: kmc_metropolis.hpp).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sstream>

#define KMC_DEFINE_METROPOLIS_KERNELS
#include "kmc_host.hpp"

using namespace kmc;
using namespace kmc_host;

// ------------------------------------------------------------------------------------------
// Many-chain Metropolis: metropolis / _metropolis, reference src/samplers.jl:59-128.
// ------------------------------------------------------------------------------------------
namespace {

MetropolisFn metropolis_fn(int density, int ndim)
{
    switch (density) {
    case KMC_GAUSSIAN_ISO: return metropolis_gaussian_iso(ndim);
    case KMC_EXPONENTIAL: return metropolis_exponential(ndim);
    case KMC_ROSENBROCK: return metropolis_rosenbrock(ndim);
    case KMC_LOGNORMAL: return metropolis_lognormal(ndim);
    case KMC_MVNORMAL2: return metropolis_mvnormal2(ndim);
    default: return nullptr;
    }
}

MetropolisTabledFn metropolis_tabled_fn(int density, int ndim)
{
    switch (density) {
    case KMC_GAUSSIAN_ISO: return metropolis_tabled_gaussian_iso(ndim);
    case KMC_EXPONENTIAL: return metropolis_tabled_exponential(ndim);
    case KMC_ROSENBROCK: return metropolis_tabled_rosenbrock(ndim);
    case KMC_LOGNORMAL: return metropolis_tabled_lognormal(ndim);
    case KMC_MVNORMAL2: return metropolis_tabled_mvnormal2(ndim);
    default: return nullptr;
    }
}

int metropolis_nd(int64_t ndim) { return ndim <= 1 ? 1 : ndim <= 2 ? 2 : ndim <= 4 ? 4 : ndim <= 8 ? 8 : ndim <= 16 ? 16 : ndim <= 32 ? 32 : 0; }

// runtime-compiled density: the Metropolis kernel (and the initial log-pdf kernel) for one register geometry
// (TND > 0: also the few-chains kernel that reads its draws from a table, chains in registers -- body densities included)
kmc_status compile_user_metropolis(kmc_user_density* ud, int ND, const std::vector<char>** out, int64_t ndim = 0, int TND = 0)
{
    if (ud->is_body) ND = 0;                     // a body density: the chain-in-memory kernel (the proposal is collected per lane)
    char key[64];
    std::snprintf(key, sizeof(key), "M:%d:%lld:%d", ND, ud->is_body ? (long long)ndim : 0ll, TND);
    std::lock_guard<std::mutex> lock(ud->mu);
    auto it = ud->code.find(key);
    if (it != ud->code.end()) { *out = &it->second; return KMC_OK; }
    const std::string dir = user_header_dir();
    const std::string h_dev = read_file(dir + "/kmc_device.hpp"), h_ker = read_file(dir + "/kmc_kernels.hpp"),
                      h_met = read_file(dir + "/kmc_metropolis.hpp");
    if (h_dev.empty() || h_ker.empty() || h_met.empty())
        return fail(KMC_ERR_BAD_ARG, "user density: kernel headers not found in " + dir + " (set KMC_CSRC_DIR)");
    std::ostringstream src;
    src << "#include \"kmc_kernels.hpp\"\n#include \"kmc_metropolis.hpp\"\n" << user_functor_source(ud) << user_density_alias(ud, ndim)
        << "extern \"C\" __global__ __launch_bounds__(256) void kmc_user_logpdf(const kmc::LogpdfArgs a) { kmc::logpdf_rows_body<UD>(a); }\n"
        << "extern \"C\" __global__ __launch_bounds__(256) void kmc_user_metropolis(const kmc::MetropolisArgs a) { ";
    if (ND > 0) src << "kmc::metropolis_chains_body<UD, " << ND << ">(a); }\n";
    else src << "kmc::metropolis_chains_any_body<UD>(a); }\n";
    if (TND > 0)
        src << "extern \"C\" __global__ __launch_bounds__(128) void kmc_user_metropolis_tabled(const kmc::MetropolisTabledArgs t) { "
            << "kmc::metropolis_chains_tabled_body<UD, " << TND << ">(t.a, t.draws, t.nsteps); }\n";
    const std::string text = src.str();
    const char* headers[3] = {h_ker.c_str(), h_dev.c_str(), h_met.c_str()};
    const char* names[3] = {"kmc_kernels.hpp", "kmc_device.hpp", "kmc_metropolis.hpp"};
    const char* opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off"};
    std::vector<char> code;
    std::string log;
    const kmc_status cst = rtc_compile_cached(text, "kmc_user_metropolis.hip", 3, headers, names, 4, opts, &code, &log);
    if (cst == KMC_ERR_BAD_ARG) return fail(KMC_ERR_BAD_ARG, "user density does not compile:\n" + log);
    if (cst != KMC_OK) return cst;
    auto ins = ud->code.emplace(key, std::move(code));
    *out = &ins.first->second;
    return KMC_OK;
}

template <class T>
hipError_t metro_alloc(T** p, size_t bytes) { return cache_alloc(reinterpret_cast<void**>(p), bytes); }

// the loaded module of a code object, shared per (density, code object, device) like the samplers' (kmc_rtc.hip: load_user)
kmc_status shared_module(kmc_user_density* ud, const std::vector<char>* code, std::shared_ptr<void>* keep, hipModule_t* mod)
{
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(ud->mu);
    auto& slot = ud->modules[{static_cast<const void*>(code), dev}];
    if (!slot) {
        hipModule_t m = nullptr;
        HIP_TRY(hipModuleLoadData(&m, code->data()));
        slot = std::shared_ptr<void>(static_cast<void*>(m), [](void* p) { if (p) (void)hipModuleUnload(static_cast<hipModule_t>(p)); });
    }
    *keep = slot;
    *mod = static_cast<hipModule_t>(slot.get());
    return KMC_OK;
}

// device buffers of one kmc_metropolis_run call
struct MetroBuffers {
    double *pos = nullptr, *logp = nullptr, *chain = nullptr, *chain_logp = nullptr, *csum = nullptr, *csumsq = nullptr,
           *step = nullptr, *xt = nullptr, *yt = nullptr, *st1 = nullptr, *st2 = nullptr, *blob = nullptr, *chain_blob = nullptr;
    uint32_t* naccept = nullptr;
    double* draws = nullptr;          // few chains: the draw table of a launch (metro_draw_fill)
    hipModule_t mod = nullptr;        // (held through keep)
    std::shared_ptr<void> keep;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipStream_t stream = nullptr;     // the stream the buffers were used on: waited for before they go back to the allocation cache
    ~MetroBuffers()
    {
        if (stream) (void)hipStreamSynchronize(stream);
        cache_free(pos); cache_free(logp); cache_free(chain); cache_free(chain_logp); cache_free(csum);
        cache_free(csumsq); cache_free(step); cache_free(xt); cache_free(yt); cache_free(st1); cache_free(st2);
        cache_free(naccept); cache_free(blob); cache_free(chain_blob); cache_free(draws);
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
    }
};

}  // namespace

namespace {

// The host route of kmc_metropolis_run: `pdf` and / or `sample_ppdf` are caller's closures (src/samplers.jl:59-61).  One
// iteration of all chains per pass: proposals (device Gaussian step, or host_propose on the current states), their
// log-pdfs (host_logpdf on the proposals, or the device density), then the accept test, counters and storage on the device.
kmc_status metropolis_host_route(const kmc_metropolis_config* c, const double* theta0, kmc_metropolis_outputs* out,
                                 const DensityParams& dp, int64_t nsamples)
{
    const int64_t nc = c->nchains, nd = c->ndim;
    ScopedStream ss;                              // never the legacy stream (kmc_host.hpp: copy_sync)
    HIP_TRY(ss.create());
    const hipStream_t st = ss.st;
    const size_t rows = (size_t)nc * (size_t)nd * sizeof(double), vec = (size_t)nc * sizeof(double);
    const bool host_pdf = c->density == KMC_HOST_DENSITY;
    const bool want_chain = (c->flags & KMC_STORE_CHAIN) != 0, want_logp = (c->flags & KMC_STORE_LOGP) != 0, want_mom = (c->flags & KMC_MOMENTS) != 0;
    struct Buf {
        double *pos = nullptr, *logp = nullptr, *prop = nullptr, *p1 = nullptr, *chain = nullptr, *chain_logp = nullptr, *csum = nullptr,
               *csumsq = nullptr, *step = nullptr, *h_rows = nullptr, *h_prop = nullptr, *h_p1 = nullptr;
        uint32_t* naccept = nullptr;
        unsigned char *acc = nullptr, *h_acc = nullptr;
        hipModule_t mod = nullptr;        // (held through keep)
        std::shared_ptr<void> keep;
        hipEvent_t ev0 = nullptr, ev1 = nullptr;
        hipStream_t stream = nullptr;     // waited for before the buffers go back to the allocation cache
        ~Buf()
        {
            if (stream) (void)hipStreamSynchronize(stream);
            cache_free(pos); cache_free(logp); cache_free(prop); cache_free(p1); cache_free(chain); cache_free(chain_logp);
            cache_free(csum); cache_free(csumsq); cache_free(step); cache_free(naccept); cache_free(acc);
            if (h_rows) (void)hipHostFree(h_rows);
            if (h_prop) (void)hipHostFree(h_prop);
            if (h_p1) (void)hipHostFree(h_p1);
            if (h_acc) (void)hipHostFree(h_acc);
            if (ev0) (void)hipEventDestroy(ev0);
            if (ev1) (void)hipEventDestroy(ev1);
        }
    } b;
    b.stream = st;
    HIP_TRY(metro_alloc(&b.pos, rows));
    HIP_TRY(metro_alloc(&b.prop, rows));
    HIP_TRY(metro_alloc(&b.logp, vec));
    HIP_TRY(metro_alloc(&b.p1, vec));
    HIP_TRY(metro_alloc(&b.naccept, (size_t)nc * sizeof(uint32_t)));
    HIP_TRY(fill_sync(b.naccept, 0, (size_t)nc * sizeof(uint32_t), st));
    HIP_TRY(copy_sync(b.pos, theta0, rows, hipMemcpyHostToDevice, st));                              // :68 deepcopy
    if (c->step) {
        HIP_TRY(metro_alloc(&b.step, (size_t)nd * sizeof(double)));
        HIP_TRY(copy_sync(b.step, c->step, (size_t)nd * sizeof(double), hipMemcpyHostToDevice, st));
    }
    if ((want_chain || want_logp) && nsamples > 0)
        KMC_TRY(check_device_room((size_t)nsamples * ((want_chain ? rows : 0) + (want_logp ? vec : 0)), "the Metropolis chain"));
    if (want_chain && nsamples > 0) HIP_TRY(metro_alloc(&b.chain, (size_t)nsamples * rows));
    if (want_logp && nsamples > 0) HIP_TRY(metro_alloc(&b.chain_logp, (size_t)nsamples * vec));
    if (want_mom) {
        HIP_TRY(metro_alloc(&b.csum, rows));
        HIP_TRY(metro_alloc(&b.csumsq, rows));
        HIP_TRY(fill_sync(b.csum, 0, rows, st));
        HIP_TRY(fill_sync(b.csumsq, 0, rows, st));
    }
    HIP_TRY(hipHostMalloc((void**)&b.h_rows, rows, hipHostMallocDefault));
    HIP_TRY(hipHostMalloc((void**)&b.h_prop, rows, hipHostMallocDefault));
    HIP_TRY(hipHostMalloc((void**)&b.h_p1, vec, hipHostMallocDefault));
    if (c->host_accepted) {
        HIP_TRY(hipMalloc((void**)&b.acc, (size_t)nc));
        HIP_TRY(hipHostMalloc((void**)&b.h_acc, (size_t)nc, hipHostMallocDefault));
    }
    // log-pdf of device rows -> device vector, for a device density
    LogpdfFn lp = nullptr;
    hipFunction_t ulp = nullptr;
    if (!host_pdf) {
        if (c->density == KMC_USER_DENSITY) {
            const std::vector<char>* code = nullptr;
            KMC_TRY(compile_user_metropolis(static_cast<kmc_user_density*>(c->user_density), metropolis_nd(nd), &code, nd));
            KMC_TRY(shared_module(static_cast<kmc_user_density*>(c->user_density), code, &b.keep, &b.mod));
            HIP_TRY(hipModuleGetFunction(&ulp, b.mod, "kmc_user_logpdf"));
        } else {
            HalfStepFn v, g;
            if (!lookup(c->density, 0, 0, 1, false, false, false, &v, &g, &lp)) return fail(KMC_ERR_BAD_ARG, "unknown density id");
        }
    }
    const unsigned grid = (unsigned)((nc + 255) / 256);
    auto device_logpdf = [&](const double* rows_dev, double* out_dev) -> hipError_t {
        const LogpdfArgs la{rows_dev, out_dev, nc, (int32_t)nd, (int32_t)nd, dp, nullptr};
        if (ulp) return launch_module(ulp, grid, 256u, st, la);
        hipLaunchKernelGGL(lp, dim3(grid), dim3(256), 0, st, la);
        return hipGetLastError();
    };
    // p0 = pdf(theta0) (:70); whatever comes out is carried, -Inf included, as in the reference
    if (host_pdf) {
        if (c->host_logpdf(theta0, nc, nd, b.h_p1, c->host_user) != 0) return fail(KMC_ERR_BAD_ARG, "the host log-pdf callback failed on the initial states");
        HIP_TRY(copy_sync(b.logp, b.h_p1, vec, hipMemcpyHostToDevice, st));
    } else {
        HIP_TRY(device_logpdf(b.pos, b.logp));
    }
    HIP_TRY(hipEventCreate(&b.ev0));
    HIP_TRY(hipEventCreate(&b.ev1));
    HIP_TRY(hipEventRecord(b.ev0, st));
    int64_t cnt = 0, slot = 0;
    for (int64_t it = 0; it < c->niter; ++it) {
        const int64_t n = it + 1 - c->nburnin;                                                   // :96
        MetroHostArgs a{};
        a.pos = b.pos; a.logp = b.logp; a.naccept = b.naccept; a.prop = b.prop; a.p1 = b.p1;
        a.chain = b.chain; a.chain_logp = b.chain_logp; a.csum = b.csum; a.csumsq = b.csumsq; a.step = b.step; a.acc_out = b.acc;
        a.nchains = nc; a.it = it; a.n = n; a.ndim = (int32_t)nd;
        a.seed_lo = (uint32_t)c->seed; a.seed_hi = (uint32_t)(c->seed >> 32);
        if (n > 0 && ++cnt == c->nthin) {                                                        // :108, :112
            cnt = 0;
            if (slot < nsamples) { a.store = 1; a.slot = slot; }
            ++slot;
        }
        if (c->host_propose) {                                                                   // :98 theta1 = sample_ppdf(theta0)
            HIP_TRY(copy_sync(b.h_rows, b.pos, rows, hipMemcpyDeviceToHost, st));
            if (c->host_propose(b.h_rows, nc, nd, b.h_prop, c->host_user) != 0)
                return fail(KMC_ERR_BAD_ARG, "the host proposal callback failed in iteration " + std::to_string(it));
            HIP_TRY(copy_sync(b.prop, b.h_prop, rows, hipMemcpyHostToDevice, st));
        } else {
            hipLaunchKernelGGL(metro_host_propose, dim3(grid), dim3(256), 0, st, a);
            HIP_TRY(hipGetLastError());
        }
        if (host_pdf) {                                                                          // :99 p1 = pdf(theta1)
            if (!c->host_propose) HIP_TRY(copy_sync(b.h_prop, b.prop, rows, hipMemcpyDeviceToHost, st));
            if (c->host_logpdf(b.h_prop, nc, nd, b.h_p1, c->host_user) != 0)
                return fail(KMC_ERR_BAD_ARG, "the host log-pdf callback failed in iteration " + std::to_string(it));
            HIP_TRY(copy_sync(b.p1, b.h_p1, vec, hipMemcpyHostToDevice, st));
        } else {
            HIP_TRY(device_logpdf(b.prop, b.p1));
        }
        hipLaunchKernelGGL(metro_host_accept, dim3(grid), dim3(256), 0, st, a);
        HIP_TRY(hipGetLastError());
        if (c->host_accepted) {
            HIP_TRY(copy_sync(b.h_acc, b.acc, (size_t)nc, hipMemcpyDeviceToHost, st));
            if (c->host_accepted(b.h_acc, nc, 0, it, a.store, c->host_user) != 0)
                return fail(KMC_ERR_BAD_ARG, "the host accept callback failed in iteration " + std::to_string(it));
        }
    }
    HIP_TRY(hipEventRecord(b.ev1, st));
    HIP_TRY(hipStreamSynchronize(st));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, b.ev0, b.ev1));
    out->device_ms = (double)ms;
    if (c->flags & KMC_CHAIN_BY_WALKER) {       // thetas[chain][sample], as the reference returns them (:113, :128)
        if (out->chain && b.chain) KMC_TRY(download_by_walker(b.chain, false, nc, nd, nd, nsamples, out->chain, st));
        if (out->chain_logp && b.chain_logp) KMC_TRY(download_by_walker(b.chain_logp, false, nc, 1, 1, nsamples, out->chain_logp, st));
    } else {
        if (out->chain && b.chain) HIP_TRY(copy_sync(out->chain, b.chain, (size_t)nsamples * rows, hipMemcpyDeviceToHost, st));
        if (out->chain_logp && b.chain_logp) HIP_TRY(copy_sync(out->chain_logp, b.chain_logp, (size_t)nsamples * vec, hipMemcpyDeviceToHost, st));
    }
    if (out->final_pos) HIP_TRY(copy_sync(out->final_pos, b.pos, rows, hipMemcpyDeviceToHost, st));
    if (out->final_logp) HIP_TRY(copy_sync(out->final_logp, b.logp, vec, hipMemcpyDeviceToHost, st));
    if (out->chain_sum && b.csum) HIP_TRY(copy_sync(out->chain_sum, b.csum, rows, hipMemcpyDeviceToHost, st));
    if (out->chain_sumsq && b.csumsq) HIP_TRY(copy_sync(out->chain_sumsq, b.csumsq, rows, hipMemcpyDeviceToHost, st));
    if (out->naccept || out->accept_ratio) {
        std::vector<uint32_t> na((size_t)nc);
        HIP_TRY(copy_sync(na.data(), b.naccept, (size_t)nc * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        const double denom = (double)(c->niter - c->nburnin);                                   // :127
        for (int64_t i = 0; i < nc; ++i) {
            if (out->naccept) out->naccept[i] = (int64_t)na[(size_t)i];
            if (out->accept_ratio) out->accept_ratio[i] = (double)na[(size_t)i] / denom;
        }
    }
    return KMC_OK;
}

}  // namespace

KMC_EXPORT kmc_status kmc_metropolis_validate(const kmc_metropolis_config* c)
{
    if (!c) return fail(KMC_ERR_BAD_ARG, "null config");
    if (c->dtype != KMC_F64) return fail(KMC_ERR_UNSUPPORTED, "only KMC_F64 is implemented");
    if (c->nchains <= 0 || c->ndim <= 0 || c->nthin <= 0 || c->niter < 0 || c->nburnin < 0)
        return fail(KMC_ERR_BAD_ARG, "nchains, ndim, nthin must be positive; niter, nburnin non-negative");
    if (c->nchains > 0xffffffffll) return fail(KMC_ERR_BAD_ARG, "at most 2^32 - 1 chains (the chain index is one Philox counter word)");
    if (!c->step && !c->host_propose) return fail(KMC_ERR_BAD_ARG, "step (proposal scale per dimension) is NULL and there is no host_propose");
    for (int64_t d = 0; c->step && d < c->ndim; ++d)
        if (!std::isfinite(c->step[d])) return fail(KMC_ERR_BAD_ARG, "step must be finite");
    if (c->flags & ~(uint32_t)(KMC_STORE_CHAIN | KMC_STORE_LOGP | KMC_MOMENTS | KMC_CHAIN_BY_WALKER | KMC_STORE_BLOBS))
        return fail(KMC_ERR_BAD_ARG, "metropolis flags: KMC_STORE_CHAIN | KMC_STORE_LOGP | KMC_MOMENTS | KMC_CHAIN_BY_WALKER | KMC_STORE_BLOBS");
    {
        const int nb = c->density == KMC_USER_DENSITY && c->user_density ? static_cast<const kmc_user_density*>(c->user_density)->nblob : 0;
        if ((c->flags & KMC_STORE_BLOBS) && nb == 0)
            return fail(KMC_ERR_BAD_ARG, "KMC_STORE_BLOBS needs a body density with blobs (kmc_user_density_create_body_blob)");
        if (nb > 0 && c->host_propose)
            return fail(KMC_ERR_UNSUPPORTED, "a density with blobs runs in the in-kernel chains: not with host_propose (use a host log-pdf returning blobs)");
    }
    if (c->density == KMC_HOST_DENSITY) {
        if (!c->host_logpdf) return fail(KMC_ERR_BAD_ARG, "KMC_HOST_DENSITY needs kmc_metropolis_config.host_logpdf");
        return KMC_OK;
    }
    if (c->host_logpdf) return fail(KMC_ERR_BAD_ARG, "host_logpdf needs density == KMC_HOST_DENSITY");
    if (c->density == KMC_USER_DENSITY) {
        if (!c->user_density) return fail(KMC_ERR_BAD_ARG, "KMC_USER_DENSITY needs kmc_metropolis_config.user_density");
        return KMC_OK;
    }
    if (c->density == KMC_ROSENBROCK && c->ndim < 2) return fail(KMC_ERR_BAD_ARG, "rosenbrock needs ndim >= 2");
    if (c->density == KMC_MVNORMAL2 && c->ndim != 2) return fail(KMC_ERR_BAD_ARG, "mvnormal2 needs ndim == 2");
    kmc_config e{};
    e.density = c->density;
    for (int i = 0; i < 8; ++i) e.params[i] = c->params[i];
    e.ndim = c->ndim;
    DensityParams dp;
    return digest_params(e, &dp);
}

KMC_EXPORT kmc_status kmc_metropolis_run(const kmc_metropolis_config* c, const double* theta0, kmc_metropolis_outputs* out)
{
    if (!theta0 || !out) return fail(KMC_ERR_BAD_ARG, "null argument");
    KMC_TRY(kmc_metropolis_validate(c));
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        (void)hipGetLastError();
        return fail(KMC_ERR_NO_DEVICE, "no HIP device visible: the Metropolis path has no CPU fallback");
    }
    if (c->device < 0 || c->device >= ndev) return fail(KMC_ERR_BAD_ARG, "device ordinal out of range");
    HIP_TRY(hipSetDevice(c->device));
    const int64_t nc = c->nchains, nd = c->ndim;
    const int64_t nsamples = c->niter > c->nburnin ? (c->niter - c->nburnin) / c->nthin : 0;      // :88
    out->nsamples = nsamples;
    out->device_ms = 0.0;
    const bool want_chain = (c->flags & KMC_STORE_CHAIN) != 0, want_logp = (c->flags & KMC_STORE_LOGP) != 0,
               want_mom = (c->flags & KMC_MOMENTS) != 0;

    kmc_config e{};
    e.density = c->density;
    for (int i = 0; i < 8; ++i) e.params[i] = c->params[i];
    e.ndim = nd;
    e.user_density = c->user_density;
    e.device = c->device;
    DensityParams dp{};
    if (c->density == KMC_USER_DENSITY) { for (int i = 0; i < 6; ++i) dp.p[i] = c->params[i]; dp.ndim = (int32_t)nd; }
    else if (c->density == KMC_HOST_DENSITY) dp.ndim = (int32_t)nd;
    else KMC_TRY(digest_params(e, &dp));
    if (c->density == KMC_HOST_DENSITY || c->host_propose)
        return metropolis_host_route(c, theta0, out, dp, nsamples);
    const int nblob = c->density == KMC_USER_DENSITY ? static_cast<const kmc_user_density*>(c->user_density)->nblob : 0;
    const bool want_blobs = nblob > 0 && ((c->flags & KMC_STORE_BLOBS) != 0 || out->blobs != nullptr);

    ScopedStream ss;                              // never the legacy stream (kmc_host.hpp: copy_sync)
    HIP_TRY(ss.create());
    const hipStream_t st = ss.st;
    MetroBuffers b;
    b.stream = st;
    const size_t rows = (size_t)nc * (size_t)nd * sizeof(double);
    HIP_TRY(metro_alloc(&b.pos, rows));
    HIP_TRY(metro_alloc(&b.logp, (size_t)nc * sizeof(double)));
    HIP_TRY(metro_alloc(&b.naccept, (size_t)nc * sizeof(uint32_t)));
    HIP_TRY(fill_sync(b.naccept, 0, (size_t)nc * sizeof(uint32_t), st));
    HIP_TRY(metro_alloc(&b.step, (size_t)nd * sizeof(double)));
    HIP_TRY(copy_sync(b.step, c->step, (size_t)nd * sizeof(double), hipMemcpyHostToDevice, st));
    HIP_TRY(copy_sync(b.pos, theta0, rows, hipMemcpyHostToDevice, st));                              // :68 deepcopy
    if ((want_chain || want_logp) && nsamples > 0)
        KMC_TRY(check_device_room((size_t)nsamples * ((want_chain ? rows : 0) + (want_logp ? (size_t)nc * sizeof(double) : 0)), "the Metropolis chain"));
    if (want_chain && nsamples > 0) HIP_TRY(metro_alloc(&b.chain, (size_t)nsamples * rows));
    if (want_logp && nsamples > 0) HIP_TRY(metro_alloc(&b.chain_logp, (size_t)nsamples * (size_t)nc * sizeof(double)));
    if (want_mom) {
        HIP_TRY(metro_alloc(&b.csum, rows));
        HIP_TRY(metro_alloc(&b.csumsq, rows));
        HIP_TRY(fill_sync(b.csum, 0, rows, st));
        HIP_TRY(fill_sync(b.csumsq, 0, rows, st));
    }
    if (nblob > 0) {
        const size_t bb = (size_t)nc * (size_t)nblob * sizeof(double);
        HIP_TRY(metro_alloc(&b.blob, bb));
        HIP_TRY(fill_sync(b.blob, 0, bb, st));
        if (want_blobs && nsamples > 0) {
            KMC_TRY(check_device_room((size_t)nsamples * bb, "the stored blobs"));
            HIP_TRY(metro_alloc(&b.chain_blob, (size_t)nsamples * bb));
        }
    }
    int ND = metropolis_nd(nd);
    if (c->density == KMC_USER_DENSITY && static_cast<const kmc_user_density*>(c->user_density)->is_body) ND = 0;   // body density: chain in memory
    if (ND == 0) {      // chains too long for registers (or a body density): state kept dimension-major in memory
        HIP_TRY(metro_alloc(&b.xt, rows));
        HIP_TRY(metro_alloc(&b.yt, rows));
        if (want_mom) {
            HIP_TRY(metro_alloc(&b.st1, rows));
            HIP_TRY(metro_alloc(&b.st2, rows));
            HIP_TRY(fill_sync(b.st1, 0, rows, st));
            HIP_TRY(fill_sync(b.st2, 0, rows, st));
        }
    }

    // Few chains (the reference's call is one): the draws of a stretch of iterations come from a wide kernel first, the chains
    // only read them (kmc_metropolis.hpp: metropolis_chains_tabled).  Up to a wave per CU (measured: x6.8 at one chain, x4 at
    // 4096, x2 at 16 384, x0.5 at 65 536 -- there the chains fill the chip themselves).  KMC_DEBUG=metro-table=0|1 forces it off / on.
    bool tabled = false;
    {
        std::string opt;
        const bool have = debug_opt("metro-table", &opt);
        const bool off = have && opt == "0", on = have && opt == "1";
        // (up to 8 dimensions for up to 16 384 chains -- measured --; longer rows for the few chains that leave most of the chip idle)
        const int64_t nd_max = c->density == KMC_USER_DENSITY ? 32 : 8;       // (menu densities: instantiated up to 8 -- build time)
        tabled = !off && nblob == 0 && ((nd <= 8 && nc <= 16384) || (nd <= nd_max && nc <= 1024) || (on && nd <= nd_max));
    }
    MetropolisFn fn = nullptr;
    hipFunction_t ufn = nullptr, ulp = nullptr, utfn = nullptr;
    if (c->density == KMC_USER_DENSITY) {
        const std::vector<char>* code = nullptr;
        KMC_TRY(compile_user_metropolis(static_cast<kmc_user_density*>(c->user_density), ND, &code, nd, tabled ? metropolis_nd(nd) : 0));
        KMC_TRY(shared_module(static_cast<kmc_user_density*>(c->user_density), code, &b.keep, &b.mod));
        HIP_TRY(hipModuleGetFunction(&ufn, b.mod, "kmc_user_metropolis"));
        HIP_TRY(hipModuleGetFunction(&ulp, b.mod, "kmc_user_logpdf"));
        if (tabled) HIP_TRY(hipModuleGetFunction(&utfn, b.mod, "kmc_user_metropolis_tabled"));
    } else {
        fn = metropolis_fn(c->density, (int)std::min<int64_t>(nd, 1 << 20));    // the geometry follows ndim (registers <= 32)
        if (!fn) return fail(KMC_ERR_BAD_ARG, "unknown density id");
    }
    const unsigned grid = (unsigned)((nc + 255) / 256);
    MetropolisTabledFn tfn = nullptr;
    int64_t table_steps = 0;
    if (tabled) {
        if (c->density != KMC_USER_DENSITY) {
            tfn = metropolis_tabled_fn(c->density, (int)nd);
            if (!tfn) return fail(KMC_ERR_BAD_ARG, "unknown density id");
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(tfn), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kMetroTileBytes));
        }
        const int64_t per_step = nc * (nd + 1) * (int64_t)sizeof(double);
        table_steps = std::max<int64_t>(64, std::min<int64_t>(1 << 16, ((int64_t)128 << 20) / per_step));
        { const long ts = debug_opt_long("metro-table-steps", 0); if (ts > 0) table_steps = ts; }   // (tests: seams)
        table_steps = std::min<int64_t>(table_steps, std::max<int64_t>(c->niter, 1));
        HIP_TRY(metro_alloc(&b.draws, (size_t)(table_steps * per_step)));
    }

    // p0 = pdf(theta0)  (:70); unlike emcee the reference carries whatever comes out, -Inf included
    const LogpdfArgs la{b.pos, b.logp, nc, (int32_t)nd, (int32_t)nd, dp, b.blob};      // (and blob0, :70-72)
    if (ulp) HIP_TRY(launch_module(ulp, grid, 256u, st, la));
    else {
        HalfStepFn v, g;
        LogpdfFn lp = nullptr;
        lookup(c->density, 0, 0, 1, false, false, false, &v, &g, &lp);
        hipLaunchKernelGGL(lp, dim3(grid), dim3(256), 0, st, la);
        HIP_TRY(hipGetLastError());
    }

    auto transpose = [&](const double* src, double* dst, bool to_dim_major) {
        const TransposeArgs ta{src, dst, nc, (int32_t)nd, to_dim_major ? 1 : 0};
        hipLaunchKernelGGL(metropolis_transpose, dim3(grid), dim3(256), 0, st, ta);
        return hipGetLastError();
    };
    if (ND == 0 && !tabled) HIP_TRY(transpose(b.pos, b.xt, true));     // (the few-chains kernel keeps its chains in registers whatever the density)

    HIP_TRY(hipEventCreate(&b.ev0));
    HIP_TRY(hipEventCreate(&b.ev1));
    HIP_TRY(hipEventRecord(b.ev0, st));
    const int64_t kItersPerLaunch = tabled ? table_steps : (int64_t)1 << 16;   // chains are independent: launches only bound a kernel's run time (and the table)
    for (int64_t it0 = 0; it0 < c->niter; it0 += kItersPerLaunch) {
        MetropolisArgs a{};
        a.pos = b.pos; a.logp = b.logp; a.naccept = b.naccept;
        a.chain = b.chain; a.chain_logp = b.chain_logp; a.csum = b.csum; a.csumsq = b.csumsq;
        a.step = b.step; a.xt = b.xt; a.yt = b.yt; a.st1 = b.st1; a.st2 = b.st2;
        a.nchains = nc;
        a.it0 = it0; a.it1 = std::min<int64_t>(c->niter, it0 + kItersPerLaunch);
        a.nburnin = c->nburnin; a.nthin = c->nthin; a.nsamples = nsamples;
        const int64_t npos = it0 - c->nburnin;          // steps with n > 0 taken before it0
        a.cnt0 = npos > 0 ? npos % c->nthin : 0;
        a.slot0 = npos > 0 ? npos / c->nthin : 0;
        a.ndim = (int32_t)nd;
        a.seed_lo = (uint32_t)c->seed; a.seed_hi = (uint32_t)(c->seed >> 32);
        a.dp = dp;
        a.blob = b.blob; a.chain_blob = b.chain_blob;
        if (tabled) {
            const int64_t nit = a.it1 - a.it0;
            MetroDrawArgs da{};
            da.out = b.draws; da.nchains = nc; da.it0 = it0; da.nsteps = (int32_t)nit; da.ndim = (int32_t)nd;
            da.seed_lo = a.seed_lo; da.seed_hi = a.seed_hi;
            hipLaunchKernelGGL(metro_draw_fill, dim3((unsigned)((nit * nc + 255) / 256)), dim3(256), 0, st, da);
            HIP_TRY(hipGetLastError());
            if (utfn) {
                MetropolisTabledArgs ta{};
                ta.a = a; ta.draws = b.draws; ta.nsteps = (int32_t)nit;
                HIP_TRY(launch_module(utfn, (unsigned)((nc + 63) / 64), 128u, st, ta, 2u * kMetroTileBytes));
            } else {
                hipLaunchKernelGGL(tfn, dim3((unsigned)((nc + 63) / 64)), dim3(128), 2 * kMetroTileBytes, st, a, (const double*)b.draws, (int)nit);
                HIP_TRY(hipGetLastError());
            }
        } else if (ufn) HIP_TRY(launch_module(ufn, grid, 256u, st, a));
        else {
            hipLaunchKernelGGL(fn, dim3(grid), dim3(256), 0, st, a);
            HIP_TRY(hipGetLastError());
        }
    }
    HIP_TRY(hipEventRecord(b.ev1, st));
    if (ND == 0 && !tabled) {
        HIP_TRY(transpose(b.xt, b.pos, false));
        if (want_mom) { HIP_TRY(transpose(b.st1, b.csum, false)); HIP_TRY(transpose(b.st2, b.csumsq, false)); }
    }
    HIP_TRY(hipEventSynchronize(b.ev1));
    HIP_TRY(hipStreamSynchronize(st));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, b.ev0, b.ev1));
    out->device_ms = (double)ms;

    if (c->flags & KMC_CHAIN_BY_WALKER) {       // thetas[chain][sample], as the reference returns them (:113, :128)
        if (out->chain && b.chain) KMC_TRY(download_by_walker(b.chain, false, nc, nd, nd, nsamples, out->chain, st));
        if (out->chain_logp && b.chain_logp) KMC_TRY(download_by_walker(b.chain_logp, false, nc, 1, 1, nsamples, out->chain_logp, st));
    } else {
        if (out->chain && b.chain) HIP_TRY(copy_sync(out->chain, b.chain, (size_t)nsamples * rows, hipMemcpyDeviceToHost, st));
        if (out->chain_logp && b.chain_logp)
            HIP_TRY(copy_sync(out->chain_logp, b.chain_logp, (size_t)nsamples * (size_t)nc * sizeof(double), hipMemcpyDeviceToHost, st));
    }
    if (out->blobs && b.chain_blob) {            // blobs[chain][sample] (KMC_CHAIN_BY_WALKER) or [sample][chain], nblob doubles each (:117)
        if (c->flags & KMC_CHAIN_BY_WALKER) KMC_TRY(download_by_walker(b.chain_blob, false, nc, nblob, nblob, nsamples, out->blobs, st));
        else HIP_TRY(copy_sync(out->blobs, b.chain_blob, (size_t)nsamples * (size_t)nc * (size_t)nblob * sizeof(double), hipMemcpyDeviceToHost, st));
    }
    if (out->final_blob && b.blob) HIP_TRY(copy_sync(out->final_blob, b.blob, (size_t)nc * (size_t)nblob * sizeof(double), hipMemcpyDeviceToHost, st));
    if (out->final_pos) HIP_TRY(copy_sync(out->final_pos, b.pos, rows, hipMemcpyDeviceToHost, st));
    if (out->final_logp) HIP_TRY(copy_sync(out->final_logp, b.logp, (size_t)nc * sizeof(double), hipMemcpyDeviceToHost, st));
    if (out->chain_sum && b.csum) HIP_TRY(copy_sync(out->chain_sum, b.csum, rows, hipMemcpyDeviceToHost, st));
    if (out->chain_sumsq && b.csumsq) HIP_TRY(copy_sync(out->chain_sumsq, b.csumsq, rows, hipMemcpyDeviceToHost, st));
    if (out->naccept || out->accept_ratio) {
        std::vector<uint32_t> na((size_t)nc);
        HIP_TRY(copy_sync(na.data(), b.naccept, (size_t)nc * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        const double denom = (double)(c->niter - c->nburnin);                                   // :127 (0/0 as in the reference)
        for (int64_t i = 0; i < nc; ++i) {
            if (out->naccept) out->naccept[i] = (int64_t)na[(size_t)i];
            if (out->accept_ratio) out->accept_ratio[i] = (double)na[(size_t)i] / denom;
        }
    }
    return KMC_OK;
}

