"""ctypes binding of ``libkissmcmc_hip.so`` (the C ABI in ``include/kissmcmc_hip.h``).

There is no CPU fallback: if the library is missing, or no HIP device is visible when a sampler
is created, this fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# KMC_LIB_PATH selects an experiment build of the same library (see build.py); default in-tree.
LIB_PATH = os.environ.get("KMC_LIB_PATH") or os.path.join(_HERE, "libkissmcmc_hip.so")

# kmc_status
OK, ERR_A_SCALE, ERR_ODD_WALKERS, ERR_TOO_FEW_WALKERS, ERR_BAD_ARG, ERR_NONFINITE_LOGP, \
    ERR_HIP, ERR_OOM, ERR_NO_DEVICE, ERR_UNSUPPORTED = range(10)
# kmc_density
GAUSSIAN_ISO, EXPONENTIAL, ROSENBROCK, LOGNORMAL, MVNORMAL2 = range(5)
USER_DENSITY = 100
HOST_DENSITY = 101
HOST_LOGPDF_FN = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.c_int64, C.c_int64, C.POINTER(C.c_double), C.c_void_p)
HOST_PROPOSE_FN = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.c_int64, C.c_int64, C.POINTER(C.c_double), C.c_void_p)
HOST_ACCEPTED_FN = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_uint8), C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_void_p)
F64 = 0
F32 = 1       # rows and chain kept in float on the device; arithmetic and host buffers stay double
STORE_CHAIN, STORE_LOGP, MOMENTS, NO_GRAPH, P2P, ISLANDS, P2P_FINEGRAINED, P2P_PUSH = 1, 2, 4, 8, 16, 64, 128, 512
STREAM_CHAIN = 2048
CHAIN_BY_WALKER = 4096
STORE_BLOBS = 8192
P2P_HANDLE_BYTES = 128
RCCL_ID_BYTES = 128

# Every symbol include/kissmcmc_hip.h declares (tests check they are all exported).
SYMBOLS = [
    "kmc_version", "kmc_device_count", "kmc_last_error", "kmc_status_string", "kmc_validate",
    "kmc_g_pdf", "kmc_cdf_g_inv", "kmc_emcee_run", "kmc_sampler_create", "kmc_sampler_destroy",
    "kmc_sampler_set_stream", "kmc_sampler_bind_positions", "kmc_sampler_p2p_export", "kmc_sampler_p2p_connect", "kmc_sampler_p2p_connect_local", "kmc_sampler_p2p_link_probe", "kmc_sampler_set_positions", "kmc_sampler_init_ball", "kmc_sampler_set_state", "kmc_sampler_run", "kmc_sampler_half_step",
    "kmc_sampler_sync", "kmc_sampler_last_run_ms", "kmc_sampler_generation", "kmc_sampler_nsamples",
    "kmc_sampler_launch_count", "kmc_sampler_describe", "kmc_sampler_device_ptr", "kmc_sampler_get_positions",
    "kmc_sampler_get_logp", "kmc_sampler_get_naccept", "kmc_sampler_get_accept_ratio",
    "kmc_sampler_get_moments", "kmc_sampler_get_chain", "kmc_sampler_get_chain_by_walker", "kmc_logpdf_eval", "kmc_logpdf_eval_host",
    "kmc_user_density_create", "kmc_user_density_create_body", "kmc_user_density_destroy", "kmc_metropolis_validate", "kmc_metropolis_run", "kmc_int_acorr", "kmc_sampler_int_acorr",
    "kmc_sizeof_config", "kmc_sizeof_metropolis_config", "kmc_sizeof_outputs", "kmc_sizeof_metropolis_outputs", "kmc_deal_seed", "kmc_deal_perm", "kmc_sampler_deal_pack", "kmc_sampler_deal_unpack",
    "kmc_sampler_get_walker_ids", "kmc_sampler_set_walker_ids", "kmc_sampler_set_chain_host", "kmc_rccl_unique_id", "kmc_sampler_rccl_init",
    "kmc_sampler_rccl_capture", "kmc_sampler_rccl_set_capture", "kmc_rccl_version", "kmc_device_free_bytes",
    "kmc_sampler_launch_mode", "kmc_updated_budget", "kmc_set_updated_budget_mb", "kmc_debug_accept_terms",
    "kmc_user_density_create_body_blob", "kmc_user_density_nblob", "kmc_logpdf_blob_eval_host", "kmc_sampler_get_blobs",
    "kmc_device_cache_release", "kmc_user_density_is_separable", "kmc_host_prefault",
]


class Config(C.Structure):
    _fields_ = [
        ("dtype", C.c_int32),
        ("density", C.c_int32),
        ("params", C.c_double * 8),
        ("nwalkers", C.c_int64),
        ("ndim", C.c_int64),
        ("ngenerations", C.c_int64),
        ("nburnin", C.c_int64),
        ("nthin", C.c_int64),
        ("a_scale", C.c_double),
        ("seed", C.c_uint64),
        ("flags", C.c_uint32),
        ("device", C.c_int32),
        ("shard_rank", C.c_int32),
        ("shard_count", C.c_int32),
        ("user_density", C.c_void_p),
        ("island_gens", C.c_int32),
        ("island_size", C.c_int32),
        ("host_logpdf", C.c_void_p),
        ("host_user", C.c_void_p),
        ("host_accepted", C.c_void_p),
        ("deal_rank", C.c_int32),
        ("deal_count", C.c_int32),
    ]


class Outputs(C.Structure):
    _fields_ = [
        ("chain", C.POINTER(C.c_double)),
        ("chain_logp", C.POINTER(C.c_double)),
        ("accept_ratio", C.POINTER(C.c_double)),
        ("naccept", C.POINTER(C.c_int64)),
        ("final_pos", C.POINTER(C.c_double)),
        ("final_logp", C.POINTER(C.c_double)),
        ("sum", C.POINTER(C.c_double)),
        ("sumsq", C.POINTER(C.c_double)),
        ("nmoment", C.c_int64),
        ("nsamples", C.c_int64),
        ("device_ms", C.c_double),
        ("blobs", C.POINTER(C.c_double)),
    ]


class MetropolisConfig(C.Structure):
    _fields_ = [
        ("dtype", C.c_int32),
        ("density", C.c_int32),
        ("params", C.c_double * 8),
        ("nchains", C.c_int64),
        ("ndim", C.c_int64),
        ("niter", C.c_int64),
        ("nburnin", C.c_int64),
        ("nthin", C.c_int64),
        ("step", C.POINTER(C.c_double)),
        ("seed", C.c_uint64),
        ("flags", C.c_uint32),
        ("device", C.c_int32),
        ("user_density", C.c_void_p),
        ("host_logpdf", C.c_void_p),
        ("host_user", C.c_void_p),
        ("host_accepted", C.c_void_p),
        ("host_propose", C.c_void_p),
    ]


class MetropolisOutputs(C.Structure):
    _fields_ = [
        ("chain", C.POINTER(C.c_double)),
        ("chain_logp", C.POINTER(C.c_double)),
        ("accept_ratio", C.POINTER(C.c_double)),
        ("naccept", C.POINTER(C.c_int64)),
        ("final_pos", C.POINTER(C.c_double)),
        ("final_logp", C.POINTER(C.c_double)),
        ("chain_sum", C.POINTER(C.c_double)),
        ("chain_sumsq", C.POINTER(C.c_double)),
        ("nsamples", C.c_int64),
        ("device_ms", C.c_double),
        ("blobs", C.POINTER(C.c_double)),
        ("final_blob", C.POINTER(C.c_double)),
    ]


class KmcError(RuntimeError):
    """A non-zero ``kmc_status``; ``.status`` holds the code."""

    def __init__(self, status: int, message: str):
        super().__init__(message)
        self.status = status


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch wheels bundle their own HIP/HSA runtime.  If this library (linked against
    # /opt/rocm) initialises HIP before torch's copies are loaded, torch later finds no GPU; loading
    # torch's libraries first keeps both usable in one process.  Pure C-ABI users are unaffected.
    if "no-torch-preload" not in os.environ.get("KMC_DEBUG", "").split(","):
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m kissmcmc_jl_amd.build` "
            "(hipcc --offload-arch=gfx950). The emcee hot path has no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    dp = C.POINTER(C.c_double)
    ip = C.POINTER(C.c_int64)
    vp = C.c_void_p
    cfgp = C.POINTER(Config)
    L.kmc_version.restype = C.c_int
    L.kmc_device_count.restype = C.c_int
    L.kmc_last_error.restype = C.c_char_p
    L.kmc_status_string.restype = C.c_char_p
    L.kmc_status_string.argtypes = [C.c_int]
    L.kmc_validate.argtypes = [cfgp]
    L.kmc_g_pdf.restype = C.c_double
    L.kmc_g_pdf.argtypes = [C.c_double, C.c_double]
    L.kmc_cdf_g_inv.restype = C.c_double
    L.kmc_cdf_g_inv.argtypes = [C.c_double, C.c_double]
    L.kmc_emcee_run.argtypes = [cfgp, dp, C.POINTER(Outputs)]
    L.kmc_sampler_create.argtypes = [cfgp, C.POINTER(vp)]
    L.kmc_sampler_destroy.restype = None
    L.kmc_sampler_destroy.argtypes = [vp]
    L.kmc_sampler_set_stream.argtypes = [vp, vp]
    L.kmc_sampler_bind_positions.argtypes = [vp, vp]
    L.kmc_sampler_p2p_export.argtypes = [vp, vp]
    L.kmc_sampler_p2p_connect.argtypes = [vp, vp]
    L.kmc_sampler_p2p_connect_local.argtypes = [vp, C.POINTER(C.c_void_p)]
    L.kmc_sampler_p2p_link_probe.argtypes = [vp, C.c_int, C.c_int64, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.kmc_sampler_set_positions.argtypes = [vp, dp]
    L.kmc_sampler_init_ball.argtypes = [vp, dp, dp, C.c_uint64, C.c_int, C.c_int]
    L.kmc_sampler_set_state.argtypes = [vp, dp, dp, ip, C.c_int64]
    L.kmc_sampler_run.argtypes = [vp, C.c_int64]
    L.kmc_sampler_half_step.argtypes = [vp, C.c_int]
    L.kmc_sampler_sync.argtypes = [vp]
    L.kmc_sampler_last_run_ms.argtypes = [vp, dp]
    L.kmc_sampler_generation.restype = C.c_int64
    L.kmc_sampler_generation.argtypes = [vp]
    L.kmc_sampler_nsamples.restype = C.c_int64
    L.kmc_sampler_nsamples.argtypes = [vp]
    L.kmc_sampler_launch_count.restype = C.c_int64
    L.kmc_sampler_launch_count.argtypes = [vp]
    L.kmc_sampler_describe.argtypes = [vp, C.c_char_p, C.c_int64]
    L.kmc_sampler_device_ptr.restype = vp
    L.kmc_sampler_device_ptr.argtypes = [vp, C.c_int]
    L.kmc_sampler_get_positions.argtypes = [vp, dp]
    L.kmc_sampler_get_logp.argtypes = [vp, dp]
    L.kmc_sampler_get_naccept.argtypes = [vp, ip]
    L.kmc_sampler_get_accept_ratio.argtypes = [vp, dp]
    L.kmc_sampler_get_moments.argtypes = [vp, dp, dp, ip]
    L.kmc_sampler_get_chain.argtypes = [vp, dp, dp]
    L.kmc_sampler_get_chain_by_walker.argtypes = [vp, dp, dp]
    L.kmc_logpdf_eval.argtypes = [cfgp, vp, vp, C.c_int64, vp]
    L.kmc_logpdf_eval_host.argtypes = [cfgp, dp, dp, C.c_int64]
    L.kmc_user_density_create.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(vp)]
    L.kmc_user_density_create_body.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.kmc_user_density_create_body_blob.argtypes = [C.c_char_p, C.c_int, C.POINTER(vp)]
    L.kmc_user_density_is_separable.restype = C.c_int
    L.kmc_user_density_is_separable.argtypes = [vp]
    L.kmc_user_density_nblob.restype = C.c_int
    L.kmc_user_density_nblob.argtypes = [vp]
    L.kmc_logpdf_blob_eval_host.argtypes = [cfgp, dp, dp, dp, C.c_int64]
    L.kmc_sampler_get_blobs.argtypes = [vp, dp, dp, C.c_int]
    L.kmc_user_density_destroy.restype = None
    L.kmc_user_density_destroy.argtypes = [vp]
    L.kmc_metropolis_validate.argtypes = [C.POINTER(MetropolisConfig)]
    L.kmc_metropolis_run.argtypes = [C.POINTER(MetropolisConfig), dp, C.POINTER(MetropolisOutputs)]
    L.kmc_int_acorr.argtypes = [dp, C.c_int64, C.c_int64, C.c_int64, C.c_double, C.c_int, dp, dp]
    L.kmc_sampler_int_acorr.argtypes = [vp, C.c_double, dp, dp]
    L.kmc_deal_seed.restype = C.c_uint64
    L.kmc_deal_seed.argtypes = [C.c_uint64, C.c_int32]
    L.kmc_deal_perm.argtypes = [C.c_uint64, C.c_int64, C.c_int32, C.c_int64, ip, ip]
    L.kmc_sampler_deal_pack.argtypes = [vp, C.c_int64, vp]
    L.kmc_sampler_deal_unpack.argtypes = [vp, vp]
    L.kmc_sampler_get_walker_ids.argtypes = [vp, ip]
    L.kmc_sampler_set_walker_ids.argtypes = [vp, ip]
    L.kmc_sampler_set_chain_host.argtypes = [vp, dp, dp]
    L.kmc_rccl_unique_id.argtypes = [vp]
    L.kmc_sampler_rccl_init.argtypes = [vp, vp]
    L.kmc_sampler_rccl_capture.argtypes = [vp, C.POINTER(C.c_int)]
    L.kmc_sampler_rccl_set_capture.argtypes = [vp, C.c_int]
    L.kmc_rccl_version.argtypes = [C.POINTER(C.c_int), C.c_char_p, C.c_int64]
    L.kmc_device_free_bytes.argtypes = [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.kmc_device_cache_release.argtypes = []
    L.kmc_device_cache_release.restype = None
    L.kmc_host_prefault.argtypes = [C.c_void_p, C.c_uint64, C.c_int]
    L.kmc_host_prefault.restype = None
    L.kmc_sampler_launch_mode.restype = C.c_int
    L.kmc_sampler_launch_mode.argtypes = [vp, C.POINTER(C.c_int)]
    L.kmc_updated_budget.restype = None
    L.kmc_updated_budget.argtypes = [ip, ip]
    L.kmc_set_updated_budget_mb.restype = None
    L.kmc_set_updated_budget_mb.argtypes = [C.c_double]
    L.kmc_debug_accept_terms.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_int64, C.c_int64, C.c_double, C.c_int64, C.c_int, ip, dp, dp, dp]
    # layout drift between this mirror and the library fails here, at load, not inside the first real call
    for fn, T in ((L.kmc_sizeof_config, Config), (L.kmc_sizeof_metropolis_config, MetropolisConfig),
                  (L.kmc_sizeof_outputs, Outputs), (L.kmc_sizeof_metropolis_outputs, MetropolisOutputs)):
        if fn() != C.sizeof(T):
            raise ImportError(f"{LIB_PATH}: struct layout mismatch ({fn.__name__}() = {fn()}, the ctypes mirror {T.__name__} is {C.sizeof(T)} bytes): "
                              "rebuild the library or update kissmcmc_jl_amd/_lib.py from include/kissmcmc_hip.h")
    _lib = L
    return L


def check(status: int) -> None:
    if status != OK:
        L = lib()
        msg = L.kmc_last_error().decode() or L.kmc_status_string(status).decode()
        raise KmcError(status, msg)
