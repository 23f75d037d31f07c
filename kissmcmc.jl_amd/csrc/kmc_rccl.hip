// kmc_rccl.hip -- RCCL, resolved at run time (dlopen), for the replica-sharded exchange of the exact partner rule:
// the all-gather of the updated half after every half-step (include/kissmcmc_hip.h: kmc_sampler_rccl_init).
// The samplers themselves do not depend on RCCL; a process that never shards never loads it.
#include <dlfcn.h>
#include <mutex>

#include "kmc_host.hpp"
#include <rccl/rccl.h>

namespace kmc_host {

namespace {
struct RcclApi {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) get_unique_id = nullptr;
    decltype(&ncclCommInitRank) comm_init_rank = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclAllGather) all_gather = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
    decltype(&ncclGetVersion) get_version = nullptr;
    std::string path;                 // where the library was found (dladdr of one of its symbols)
};

const RcclApi* rccl_api()
{
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            api.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (api.lib) break;
        }
        if (!api.lib) return;
        api.get_unique_id = reinterpret_cast<decltype(api.get_unique_id)>(dlsym(api.lib, "ncclGetUniqueId"));
        api.comm_init_rank = reinterpret_cast<decltype(api.comm_init_rank)>(dlsym(api.lib, "ncclCommInitRank"));
        api.comm_destroy = reinterpret_cast<decltype(api.comm_destroy)>(dlsym(api.lib, "ncclCommDestroy"));
        api.all_gather = reinterpret_cast<decltype(api.all_gather)>(dlsym(api.lib, "ncclAllGather"));
        api.error_string = reinterpret_cast<decltype(api.error_string)>(dlsym(api.lib, "ncclGetErrorString"));
        api.get_version = reinterpret_cast<decltype(api.get_version)>(dlsym(api.lib, "ncclGetVersion"));
        Dl_info info;
        if (api.all_gather && dladdr(reinterpret_cast<void*>(api.all_gather), &info) && info.dli_fname) api.path = info.dli_fname;
    });
    return (api.lib && api.get_unique_id && api.comm_init_rank && api.comm_destroy && api.all_gather && api.error_string && api.get_version) ? &api : nullptr;
}

kmc_status rccl_fail(const RcclApi* a, const char* what, ncclResult_t r)
{
    return fail(KMC_ERR_HIP, std::string(what) + ": " + a->error_string(r));
}
}  // namespace

static_assert(NCCL_UNIQUE_ID_BYTES == KMC_RCCL_ID_BYTES, "unique id blob size");

// every entry point the library uses resolved from the librccl.so this process finds; its version code and path
kmc_status rccl_version(int* version, const char** path)
{
    const RcclApi* a = rccl_api();
    if (!a) return fail(KMC_ERR_UNSUPPORTED, "librccl.so could not be loaded (or lacks one of ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllGather / ncclGetErrorString / ncclGetVersion)");
    int v = 0;
    const ncclResult_t r = a->get_version(&v);
    if (r != ncclSuccess) return rccl_fail(a, "ncclGetVersion", r);
    if (version) *version = v;
    if (path) *path = a->path.c_str();
    return KMC_OK;
}

kmc_status rccl_unique_id(void* id_out)
{
    const RcclApi* a = rccl_api();
    if (!a) return fail(KMC_ERR_UNSUPPORTED, "librccl.so could not be loaded");
    ncclUniqueId id;
    const ncclResult_t r = a->get_unique_id(&id);
    if (r != ncclSuccess) return rccl_fail(a, "ncclGetUniqueId", r);
    std::memcpy(id_out, &id, sizeof(id));
    return KMC_OK;
}

kmc_status rccl_comm_create(const void* id_bytes, int rank, int nranks, void** comm_out)
{
    const RcclApi* a = rccl_api();
    if (!a) return fail(KMC_ERR_UNSUPPORTED, "librccl.so could not be loaded");
    ncclUniqueId id;
    std::memcpy(&id, id_bytes, sizeof(id));
    ncclComm_t comm = nullptr;
    const ncclResult_t r = a->comm_init_rank(&comm, nranks, id, rank);
    if (r != ncclSuccess) return rccl_fail(a, "ncclCommInitRank", r);
    *comm_out = comm;
    return KMC_OK;
}

void rccl_comm_destroy(void* comm)
{
    const RcclApi* a = rccl_api();
    if (a && comm) (void)a->comm_destroy(static_cast<ncclComm_t>(comm));
}

// in place: rank r's `count` doubles sit at recv + r * count
kmc_status rccl_all_gather_f64(void* comm, const double* send, double* recv, size_t count, hipStream_t stream)
{
    const RcclApi* a = rccl_api();
    if (!a) return fail(KMC_ERR_UNSUPPORTED, "librccl.so could not be loaded");
    const ncclResult_t r = a->all_gather(send, recv, count, ncclDouble, static_cast<ncclComm_t>(comm), stream);
    if (r != ncclSuccess) return rccl_fail(a, "ncclAllGather", r);
    return KMC_OK;
}

}  // namespace kmc_host
