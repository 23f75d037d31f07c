"""The C-ABI shared library loads and exports every symbol include/kissmcmc_hip.h declares;
calls that need no device behave (validation, g helpers, error strings).  No compute here."""
import ctypes as C
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "kissmcmc_hip.h")


def _declared_symbols():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(kmc_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(kmc):
    from kissmcmc_jl_amd import _lib
    L = _lib.lib()
    declared = _declared_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(L, name), f"{name} declared in the header but not exported"
    assert sorted(_lib.SYMBOLS) == declared
    assert L.kmc_version() == 100


def test_struct_layout_matches_header(kmc, tmp_path):
    """sizeof/offsetof of kmc_config and kmc_outputs as gcc sees the header == the ctypes mirror."""
    from kissmcmc_jl_amd import _lib
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "kissmcmc_hip.h"\n'
                   'int main(void){printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(kmc_config), offsetof(kmc_config, nwalkers),'
                   ' offsetof(kmc_config, seed), offsetof(kmc_config, host_logpdf), sizeof(kmc_outputs), offsetof(kmc_outputs, device_ms));return 0;}\n')
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    want = [C.sizeof(_lib.Config), _lib.Config.nwalkers.offset, _lib.Config.seed.offset, _lib.Config.host_logpdf.offset,
            C.sizeof(_lib.Outputs), _lib.Outputs.device_ms.offset]
    assert got == want
    # the Metropolis structs (reference src/samplers.jl:59-128)
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "kissmcmc_hip.h"\n'
                   'int main(void){printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(kmc_metropolis_config), offsetof(kmc_metropolis_config, nchains),'
                   ' offsetof(kmc_metropolis_config, step), offsetof(kmc_metropolis_config, user_density), sizeof(kmc_metropolis_outputs),'
                   ' offsetof(kmc_metropolis_outputs, device_ms));return 0;}\n')
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    want = [C.sizeof(_lib.MetropolisConfig), _lib.MetropolisConfig.nchains.offset, _lib.MetropolisConfig.step.offset,
            _lib.MetropolisConfig.user_density.offset, C.sizeof(_lib.MetropolisOutputs), _lib.MetropolisOutputs.device_ms.offset]
    assert got == want
    # ... and what the LIBRARY was compiled with (the load-time guard of every binding: config AND output structs)
    L = _lib.lib()
    assert L.kmc_sizeof_config() == C.sizeof(_lib.Config) and L.kmc_sizeof_outputs() == C.sizeof(_lib.Outputs)
    assert L.kmc_sizeof_metropolis_config() == C.sizeof(_lib.MetropolisConfig) and L.kmc_sizeof_metropolis_outputs() == C.sizeof(_lib.MetropolisOutputs)


def test_flag_and_id_constants_match_header(kmc, tmp_path):
    """The enum values of include/kissmcmc_hip.h as gcc sees them == the constants of the ctypes module (and, through
    test_julia_shim_structs_mirror_the_header, of the Julia shim)."""
    from kissmcmc_jl_amd import _lib
    names = ["STORE_CHAIN", "STORE_LOGP", "MOMENTS", "NO_GRAPH", "P2P", "ISLANDS", "P2P_FINEGRAINED", "P2P_PUSH",
             "STREAM_CHAIN", "CHAIN_BY_WALKER", "STORE_BLOBS", "F64", "F32", "GAUSSIAN_ISO", "EXPONENTIAL", "ROSENBROCK", "LOGNORMAL",
             "MVNORMAL2", "USER_DENSITY", "HOST_DENSITY", "P2P_HANDLE_BYTES", "RCCL_ID_BYTES"]
    src = tmp_path / "enums.c"
    src.write_text('#include <stdio.h>\n#include "kissmcmc_hip.h"\nint main(void){' +
                   "".join(f'printf("%lld\\n", (long long)KMC_{n});' for n in names) + "return 0;}\n")
    exe = tmp_path / "enums"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    assert got == [getattr(_lib, n) for n in names]


def _cfg(_lib, **kw):
    c = _lib.Config()
    c.dtype, c.density = _lib.F64, _lib.GAUSSIAN_ISO
    c.params[0], c.params[1] = 0.0, 1.0
    c.nwalkers, c.ndim, c.ngenerations, c.nburnin, c.nthin = 10, 2, 10, 5, 1
    c.a_scale, c.seed = 2.0, 1
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def test_validate_returns_the_reference_assert_conditions(kmc):
    from kissmcmc_jl_amd import _lib
    L = _lib.lib()
    v = lambda **kw: L.kmc_validate(C.byref(_cfg(_lib, **kw)))
    assert v() == _lib.OK
    assert v(a_scale=1.0) == _lib.ERR_A_SCALE                    # samplers.jl:200
    assert v(nwalkers=11) == _lib.ERR_ODD_WALKERS                # :202
    assert L.kmc_last_error() == b"Use an even number of walkers."
    assert v(nwalkers=2) == _lib.ERR_TOO_FEW_WALKERS             # :205
    assert L.kmc_last_error() == b"Use more walkers: at least DOF+2, but better many more."
    assert v(nwalkers=4) == _lib.OK
    assert v(nthin=0) == _lib.ERR_BAD_ARG
    assert v(density=99) == _lib.ERR_BAD_ARG
    assert v(dtype=7) == _lib.ERR_UNSUPPORTED
    assert v(shard_count=3) == _lib.ERR_BAD_ARG                  # 5 walkers per half not divisible by 3
    assert v(shard_count=5, shard_rank=4) == _lib.OK
    assert v(density=_lib.ROSENBROCK, ndim=1, nwalkers=10) == _lib.ERR_BAD_ARG
    assert L.kmc_validate(None) == _lib.ERR_BAD_ARG
    assert v(ngenerations=2 ** 31) == _lib.ERR_UNSUPPORTED       # the step index 2 g + half is 32 bits
    assert v(ngenerations=2 ** 31 - 1) == _lib.OK
    assert v(nwalkers=2 ** 31, ndim=2) == _lib.ERR_UNSUPPORTED   # row indices are 31 bits


def test_int_acorr_argument_checks_need_no_device(kmc):
    from kissmcmc_jl_amd import _lib
    L = _lib.lib()
    dp = C.POINTER(C.c_double)
    x = np.zeros(64)
    out = np.zeros(2)
    p = lambda a: a.ctypes.data_as(dp)
    assert L.kmc_int_acorr(p(x), 32, 1, 2, 1.0, 0, p(out), p(out)) == _lib.ERR_BAD_ARG      # @assert c>1, analysis.jl:141
    assert L.kmc_int_acorr(p(x), 3, 1, 2, 5.0, 0, p(out), p(out)) == _lib.ERR_BAD_ARG       # too short
    assert L.kmc_int_acorr(None, 32, 1, 2, 5.0, 0, p(out), p(out)) == _lib.ERR_BAD_ARG
    assert L.kmc_sampler_int_acorr(None, 5.0, p(out), p(out)) == _lib.ERR_BAD_ARG


def test_g_helpers_known_answers(kmc):
    """reference test/emcee.jl:6-8 on the product's host helpers."""
    a = 3.5
    assert kmc.cdf_g_inv(1, a) == pytest.approx(a, rel=1e-12)
    assert kmc.cdf_g_inv(0, a) == pytest.approx(1 / a, rel=1e-12)
    assert kmc.g_pdf(1 / a - 1e-9, a) == 0.0 and kmc.g_pdf(a + 1e-9, a) == 0.0
    z = np.arange(1 / a, a, 0.001)
    assert np.sum([kmc.g_pdf(v, a) for v in z]) * 0.001 == pytest.approx(1.0, abs=2e-3)


def test_user_density_compiles_and_reports_syntax_errors(kmc):
    """hiprtc runs offline (no GPU needed): a valid expression pair compiles for gfx950, a broken one
    returns KMC_ERR_BAD_ARG with the compiler's message."""
    from kissmcmc_jl_amd import _lib
    d = kmc.ExprDensity("d < n-1 ? -(p[0]-x)*(p[0]-x)/p[2] : 0.0", "-p[1]*(y-x*x)*(y-x*x)/p[2]", params=[1, 100, 20])
    assert d.user_handle is not None and d.params() == [1.0, 100.0, 20.0]
    with pytest.raises(kmc.KmcError, match="undeclared identifier 'z'") as e:
        kmc.ExprDensity("-0.5*x*z")
    assert e.value.status == _lib.ERR_BAD_ARG
    with pytest.raises(ValueError):
        kmc.ExprDensity("x", params=range(7))
    c = _cfg(_lib, density=_lib.USER_DENSITY)
    assert _lib.lib().kmc_validate(C.byref(c)) == _lib.ERR_BAD_ARG     # handle missing
    c.user_density = d.user_handle
    assert _lib.lib().kmc_validate(C.byref(c)) == _lib.OK


def test_bodies_that_are_not_sums_over_elements_are_not_routed(kmc):
    # (hiprtc compiles without a device: the recogniser and the functor it generates are checked here, the routed kernels in -m gpu)
    """What the recogniser must leave alone (each would change meaning as a per-element function): early returns, more than four
    sums, a sum that reads another, another index, a second loop, state carried between elements, the running sum read inside the
    loop, no loop at all -- and what it must take: one to four sums fed by one pass over the elements."""
    assert not kmc.CDensity("double s = 0.0; for (int i = 0; i < n; ++i) { if (x[i] < 0.0) return -INFINITY; s += x[i]; } return -(p[0] * s);", params=[1.0]).separable
    for body in ("double a=0,b=0,c=0,d=0,e=0; for (int i = 0; i < n; ++i) { a += x[i]; b += x[i]; c += x[i]; d += x[i]; e += x[i]; } return -(a+b+c+d+e);",   # five sums
                 "double s = 0, t = 0; for (int i = 0; i < n; ++i) { s += x[i] * x[i]; t += s; } return -t;",          # a sum that reads another
                 "double s = 0; for (int i = 0; i < n; ++i) s += x[i]*x[i]; for (int i = 0; i + 2 < n; ++i) s += p[0]*x[i]*x[i+2]; return -0.5*s;",   # two loops
                 "double s = 0; for (int i = 0; i < n; ++i) s += x[i] * x[0]; return -s;",
                 "double s = 0; double c = 1.0; for (int i = 0; i < n; ++i) { c = c * 0.5; s += c * x[i]; } return -s;",
                 "double s = 0; for (int i = 0; i < n; ++i) { s += x[i] * (1.0 + s); } return -s;",
                 "double s = 0; for (int i = 0; i < n; ++i) s += x[i + 1]; return -s;",
                 "double s = 0; for (int i = 0; i + 1 < n; ++i) { const double* q = &x[i]; s += q[0] * q[1]; } return -s;",
                 "double s = 0; for (int i = 0; i < n; i += 2) s += x[i]; return -s;",
                 "double s = 1.0; for (int i = 0; i < n; ++i) s += x[i]; return -s;",
                 "const double t = x[0] + 5.0; return -(t * t) / 18.0;",
                 # state carried between elements in ways a first matcher let through (round 4 review): the loop index
                 # assigned, the increment used as a value, a shadowed index, a variable written through its address, a macro
                 "double s = 0; for (int i = 0; i < n; ++i) { s += x[i]; i = i + 1; } return -s;",
                 "double s = 0; for (int i = 0; i < n; ++i) { double t = (s += x[i]); s += t; } return -s;",
                 "double s = 0; for (int i = 0; i < n; ++i) { { int i = 3; s += x[i]; } } return -s;",
                 "double s = 0; double c = 0; for (int i = 0; i < n; ++i) { s += modf(x[i] + c, &c); } return -s;",
                 "double s = 0; for (int i = 0; i < n; ++i) s += x[i] > 0 ? (s += 1.0) : x[i]; return -s;",
                 "#define Z x\ndouble s = 0; for (int i = 0; i < n; ++i) s += Z[i]*Z[(i+2)%n]; return -s;",
                 "double s = 0; for (int i = 0; i < n; ++i) s += x[i]*x[i]; return -s + x[0];",
                 "double m = x[0]; double s = 0; for (int i = 0; i < n; ++i) s += (x[i]-m)*(x[i]-m); return -s;"):
        assert not kmc.CDensity(body, params=[0.5]).separable, body
    for body in ("double s = 0; for (int i = 0; i < n; ++i) s += x[i] * x[i]; return -0.5 * s;",
                 "double s = 0.0, t = 0.0; for (int i = 0; i < n; ++i) { s += x[i] * x[i]; t += x[i]; } return -0.5 * (s + p[0] * t * t);",   # two sums, one pass
                 "double s = 0.0; double t = 0.0; double u = 0; for (int i = 0; i + 1 < n; ++i) { double d = x[i+1]-x[i]; s += d*d; t += x[i]; u += x[i+1]*x[i]; } return -(s + 0.1*t*t + 0.01*u);",
                 "const double w = p[1] * p[1]; double s = 0; for (int i = 0; i < n; i++) { s += w * x[i] * x[i]; } return -0.5 * s / w;",
                 "double s = 0; /* sum */ for (int i = 0; i < n - 1; ++i) { // pairs\n s += (x[i+1]-x[i])*(x[i+1]-x[i]); } return -0.5*s;",
                 "double s = 0; for (int i = 0; i < n; ++i) { if (x[i] > 0) s += x[i]; else s += -2.0 * x[i]; } return -s;",
                 "double s = 0; for (int i = 0; i < n; ++i) { const double t = x[i] - p[0]; s += (i + 1) * t * t / n; } return -0.5 * s;"):
        assert kmc.CDensity(body, params=[0.5, 2.0]).separable, body


def test_runtime_compiled_densities_through_the_offline_compiler(kmc, kmc_debug, tmp_path, monkeypatch):
    """KMC_DEBUG=rtc=hipcc: the ROCm installation's clang as a CHILD process builds the code objects instead of hiprtc in this process
    (inside a PyTorch process hiprtc resolves to the older comgr the torch wheel bundles, whose code for the half-step kernels is
    8-20 % slower; the samplers ask for the offline compiler by themselves for big ensembles).  Same results, cached on disk, syntax
    errors reported with the compiler's message; no device needed."""
    monkeypatch.setenv("KMC_CACHE_DIR", str(tmp_path))
    kmc_debug.set("rtc", "hipcc")
    d = kmc.CDensity("double s = 0; for (int i = 0; i < n; ++i) s += x[i] * x[i]; return -0.5 * s;")
    assert d.separable
    assert len(list(tmp_path.glob("*.co"))) >= 1
    with pytest.raises(kmc.KmcError, match="does not compile"):
        kmc.CDensity("return x[0] +;")
    kmc_debug.set("rtc", "hiprtc")
    n_before = len(list(tmp_path.glob("*.co")))
    kmc.CDensity("double s = 0; for (int i = 0; i < n; ++i) s += x[i] * x[i]; return -0.5 * s;")      # the other compiler: its own cache entries
    assert len(list(tmp_path.glob("*.co"))) > n_before


def test_product_never_touches_the_oracle():
    """The product package must not import, link or execute anything under oracle/."""
    pkg = os.path.join(ROOT, "kissmcmc.jl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp", ".jl")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "kmco_" not in txt, f
                assert "libkmc_oracle" not in txt, f


def test_sampler_fails_loudly_without_a_device(kmc):
    from kissmcmc_jl_amd import _lib
    if _lib.lib().kmc_device_count() > 0:
        pytest.skip("a HIP device is visible")
    with pytest.raises(kmc.KmcError, match="no CPU fallback") as e:
        kmc.Sampler(kmc.GaussianIso(), 10, 2, 10)
    assert e.value.status == _lib.ERR_NO_DEVICE
    import ctypes as C
    free, total = C.c_uint64(0), C.c_uint64(0)
    assert _lib.lib().kmc_device_free_bytes(0, C.byref(free), C.byref(total)) == _lib.ERR_NO_DEVICE
    z = (C.c_double * 4)()
    assert _lib.lib().kmc_debug_accept_terms(1, 0, 0, 4, 8, 2.0, 2, 0, None, z, z, z) == _lib.ERR_NO_DEVICE


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not available")
def test_stream_is_rocrand_philox(oracle, tmp_path):
    """The oracle's draw block == rocRAND's host-callable Philox4x32-10 engine after
    rocrand_init(seed, subsequence = walker, offset = 4 * step): the stream the kernels document."""
    src = tmp_path / "rr.cpp"
    src.write_text(r'''
#include <cstdio>
#include <cstdlib>
#include <rocrand/rocrand_philox4x32_10.h>
int main(int argc, char** argv) {
    unsigned long long seed = strtoull(argv[1], 0, 0), step = strtoull(argv[2], 0, 0), walker = strtoull(argv[3], 0, 0);
    rocrand_device::philox4x32_10_engine e(seed, walker, 4ULL * step);
    uint4 r = e.next4();
    printf("%u %u %u %u\n", r.x, r.y, r.z, r.w);
    return 0;
}''')
    exe = tmp_path / "rr"
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    subprocess.check_call([hipcc, "-O1", "--offload-arch=gfx950", str(src), "-o", str(exe)], stderr=subprocess.DEVNULL)
    for seed, step, walker in [(0, 0, 0), (12345, 7, 65535), (0xDEADBEEFCAFEF00D, 2 ** 33 + 5, 2 ** 32 + 9)]:
        got = tuple(int(v) for v in subprocess.check_output([str(exe), str(seed), str(step), str(walker)]).split())
        want = oracle.philox4x32_10((step & 0xFFFFFFFF, step >> 32, walker & 0xFFFFFFFF, walker >> 32),
                                    (seed & 0xFFFFFFFF, seed >> 32))
        assert got == want


def test_julia_shim_structs_mirror_the_header():
    """No julia in this image: the shim cannot be executed, so its struct mirrors are checked statically -- field names,
    order and sizes of the `Base.@kwdef struct`s in KissMCMCHIP.jl against the ctypes mirrors (which test_struct_layouts
    checks against the C header with gcc).  At run time the shim's __init__ compares sizeof with kmc_sizeof_config()."""
    import ctypes as C
    import re
    from kissmcmc_jl_amd import _lib
    src = open(os.path.join(ROOT, "kissmcmc.jl_amd", "julia", "src", "KissMCMCHIP.jl")).read()
    jl_size = {"Int32": 4, "UInt32": 4, "Int64": 8, "UInt64": 8, "Float64": 8, "Ptr{Cvoid}": 8, "Ptr{Float64}": 8, "Ptr{Int64}": 8,
               "NTuple{8,Float64}": 64}
    for jl_name, mirror in (("KmcConfig", _lib.Config), ("KmcOutputs", _lib.Outputs), ("KmcMetropolisConfig", _lib.MetropolisConfig),
                            ("KmcMetropolisOutputs", _lib.MetropolisOutputs)):
        m = re.search(r"Base\.@kwdef (?:mutable )?struct " + jl_name + r"\n(.*?)\nend\n", src, re.S)
        assert m, jl_name
        fields = re.findall(r"^\s+(\w+)::([\w{},]+)", m.group(1), re.M)
        assert [f for f, _ in fields] == [f for f, _ in mirror._fields_], jl_name
        assert [jl_size[t] for _, t in fields] == [C.sizeof(t) for _, t in mirror._fields_], jl_name
    assert "function __init__()" in src
    for fn in ("kmc_sizeof_config", "kmc_sizeof_metropolis_config", "kmc_sizeof_outputs", "kmc_sizeof_metropolis_outputs"):
        assert f"(:{fn}, LIB)" in src, f"the shim's __init__ does not check {fn}()"
    # the flag constants the shim uses: same bits as the header's (via the ctypes module, itself checked against the header)
    consts = dict(re.findall(r"^const (KMC_\w+) = UInt32\(1\) << (\d+)", src, re.M))
    assert consts and all(1 << int(bit) == getattr(_lib, name[4:]) for name, bit in consts.items()), consts
    # the host-side pre/post-processing is KissMCMC's own, not re-typed here
    assert "import KissMCMC: emcee, metropolis, make_theta0s, squash_walkers" in src
    assert "function make_theta0s" not in src and "function squash_walkers" not in src


def test_julia_shim_ccalls_match_the_header_prototypes():
    """No julia here: every `ccall((:name, LIB), Ret, (ArgTypes...), ...)` of the shim is checked statically against the prototype
    of `name` in include/kissmcmc_hip.h -- the function exists, the argument count matches, and each Julia type is of the same
    kind (pointer / 32-bit int / 64-bit int / double / C string) as the C parameter in that position, the return type likewise."""
    import re
    hdr = open(os.path.join(ROOT, "include", "kissmcmc_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", " ", hdr, flags=re.S)                       # comments out
    protos = {}
    for m in re.finditer(r"\b([A-Za-z_][\w\s\*]*?)\b(kmc_\w+)\s*\(([^;{}]*?)\)\s*;", hdr):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if "typedef" in ret or "(" in ret:
            continue
        protos[name] = (ret, [] if args in ("", "void") else [a.strip() for a in args.split(",")])

    def ckind(t):
        t = t.strip()
        if "*" in t or "[" in t:
            return "cstring" if re.match(r"(const\s+)?char\s*\*", t) else "ptr"
        base = re.sub(r"\b(const|unsigned)\b", "", t).split()
        base = base[0] if base else t
        return {"int": "i32", "int32_t": "i32", "uint32_t": "i32", "kmc_status": "i32", "int64_t": "i64", "uint64_t": "i64", "double": "f64",
                "void": "void"}.get(base, base)

    def jkind(t):
        t = t.strip()
        if t == "Cstring":
            return "cstring"
        if t.startswith("Ptr{") or t.startswith("Ref{"):
            return "ptr"
        return {"Cint": "i32", "Int32": "i32", "UInt32": "i32", "Int64": "i64", "UInt64": "i64", "Float64": "f64", "Cvoid": "void"}[t]

    def split_top(s):
        out, depth, cur = [], 0, ""
        for ch in s:
            if ch in "{(":
                depth += 1
            elif ch in "})":
                depth -= 1
            if ch == "," and depth == 0:
                out.append(cur)
                cur = ""
            else:
                cur += ch
        return [x.strip() for x in out + [cur] if x.strip()]

    src = open(os.path.join(ROOT, "kissmcmc.jl_amd", "julia", "src", "KissMCMCHIP.jl")).read()
    calls = re.findall(r"ccall\(\(:(\w+), LIB\),\s*(\w+),\s*\((.*?)\)\s*(?:,|\))", src, re.S)
    assert len(calls) >= 12
    seen = set()
    for name, ret, argt in calls:
        assert name in protos, f"{name}: not declared in include/kissmcmc_hip.h"
        cret, cargs = protos[name]
        jargs = split_top(argt)
        assert len(jargs) == len(cargs), f"{name}: ccall passes {len(jargs)} arguments, the header declares {len(cargs)}"
        for i, (j, c) in enumerate(zip(jargs, cargs)):
            assert jkind(j) == ckind(c), f"{name}: argument {i + 1} is {j} in the shim, `{c}` in the header"
        rk = ckind(cret)
        assert jkind(ret) == rk or (rk == "cstring" and ret == "Cstring"), f"{name}: return type {ret} vs `{cret}`"
        seen.add(name)
    assert {"kmc_emcee_run", "kmc_metropolis_run", "kmc_user_density_create_body_blob", "kmc_logpdf_blob_eval_host", "kmc_int_acorr"} <= seen


def test_julia_package_layout_and_its_test_file_use_only_what_the_module_has():
    """kissmcmc.jl_amd/julia is a package a maintainer with Julia can `] dev` and `] test`: Project.toml (depends on KissMCMC by its registered
    uuid), src/KissMCMCHIP.jl, test/runtests.jl -- the reference's emcee test (test/emcee.jl:17-48 over the four cases without blobs of test/runtests.jl:52-79)
    on device densities, plus the README call.  No julia here: checked statically -- every name the test file calls is exported by the module (or is Base /
    Test / Statistics), every keyword it passes to `emcee` is one the shim's method takes, the library path of the module points at the built library."""
    import re
    jdir = os.path.join(ROOT, "kissmcmc.jl_amd", "julia")
    proj = open(os.path.join(jdir, "Project.toml")).read()
    assert 'name = "KissMCMCHIP"' in proj and re.search(r'^uuid = "[0-9a-f-]{36}"', proj, re.M)
    assert 'KissMCMC = "79d62d8d-4dfd-5781-bc85-ce78e0ac132a"' in proj               # KissMCMC.jl's uuid (the reference's Project.toml:2)
    assert re.search(r'\[targets\]\s*test = \[[^\]]*"Test"', proj)
    src = open(os.path.join(jdir, "src", "KissMCMCHIP.jl")).read()
    test = open(os.path.join(jdir, "test", "runtests.jl")).read()
    test = re.sub(r"#.*", "", test)                                                   # comments out
    test = re.sub(r'"(?:[^"\\]|\\.)*"', '""', test)                                  # string literals out (the C body of the CDensity case)
    exports = set(re.search(r"^export (.*)$", src, re.M).group(1).replace(",", " ").split())
    assert {"emcee", "make_theta0s", "squash_walkers", "GaussianIso", "LogNormal", "MvNormal2", "Rosenbrock", "Exponential", "CDensity"} <= exports
    known = {"Test", "Statistics", "KissMCMCHIP", "DeviceCase", "UInt64", "AssertionError", "Base", "String", "Int", "Float64"}
    called = set(re.findall(r"\b([A-Za-z_]\w*)\(", test))
    base_fns = {"sqrt", "var", "exp", "length", "mean", "median", "std", "abs", "all", "isapprox", "zeros", "stds"}
    for name in sorted(called - base_fns - known):
        assert name in exports, f"test/runtests.jl calls {name}(...), which KissMCMCHIP does not export"
    # the keywords of every emcee call are keywords of the device method
    sig = re.search(r"function emcee\(pdf::DeviceLogPdf, theta0s;(.*?)\)\s*\n\s+device_blobs", src, re.S).group(1)
    takes = set(re.findall(r"(\w+!?)\s*=", sig))
    for call in re.findall(r"emcee\([^;()]*;([^()]*(?:\([^()]*\)[^()]*)*)\)", test):
        for kw in re.findall(r"(\w+)\s*=", call):
            assert kw in takes, f"emcee(...; {kw}=...) in test/runtests.jl: not a keyword of the device method ({sorted(takes)})"
    # constructors with the arities the test file uses
    assert "struct LogNormal <: DeviceLogPdf; mu::Float64; sigma::Float64; end" in src and "MvNormal2(mean, cov::AbstractMatrix)" in src
    assert "struct Rosenbrock <: DeviceLogPdf; a::Float64; b::Float64; scale::Float64; end" in src
    # src/ sits two levels below the directory that holds the built library
    assert 'joinpath(@__DIR__, "..", "..", "libkissmcmc_hip.so")' in src
    assert os.path.exists(os.path.join(jdir, "..", "build.py"))
    # the test set is the reference's: shapes, acceptance bound, squash, the three moment checks
    for needle in ("tc.niter ÷ tc.nwalkers ÷ 2", "accept_ratio > 0.1", "squash_walkers(samples...", "length(thetas) == tc.niter ÷ 2", "median(thetas)"):
        assert needle in test, needle
