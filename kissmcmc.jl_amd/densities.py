"""The log-density menu that replaces the reference's arbitrary ``pdf`` closure
(reference ``src/samplers.jl:257``) on the device.

Each object carries the ``kmc_density`` id and parameter vector handed to the C ABI, and is
callable on a single host ``theta`` with the same formula -- that host evaluation is used only
where the reference itself evaluates ``pdf`` outside the sampling loop, i.e. by
``make_theta0s`` to test ``pdf(theta) > -Inf`` (``src/samplers.jl:336-338``).
Normalisation constants are dropped.
"""
from __future__ import annotations

import math

import numpy as np

from . import _lib


class DeviceLogPdf:
    """Base class: a log-density the HIP kernels know how to evaluate."""

    density_id: int = -1
    name = "?"
    user_handle = None      # kmc_user_density* for runtime-compiled densities

    def params(self):
        raise NotImplementedError

    def __call__(self, theta):
        raise NotImplementedError

    def check_ndim(self, ndim: int) -> None:
        pass

    def finite_rows(self, X):
        """Vectorised ``pdf(row) > -Inf`` for the rows of ``X`` (used by ``make_theta0s``)."""
        X = np.asarray(X, dtype=np.float64)
        return np.all(np.isfinite(X), axis=1)

    def __repr__(self):
        return f"{type(self).__name__}({', '.join(f'{v:g}' for v in self.params())})"


class GaussianIso(DeviceLogPdf):
    """``-1/2 * sum(((x - mu)/sigma)^2)`` (cf. reference ``test/runtests.jl:80`` for ``mu=-5, sigma=3``)."""

    density_id = _lib.GAUSSIAN_ISO
    name = "gaussian_iso"

    def __init__(self, mu: float = 0.0, sigma: float = 1.0):
        if not sigma > 0:
            raise ValueError("sigma must be > 0")
        self.mu, self.sigma = float(mu), float(sigma)

    def params(self):
        return [self.mu, self.sigma]

    def __call__(self, theta):
        t = (np.asarray(theta, dtype=np.float64) - self.mu) * (1.0 / self.sigma)
        return float(-0.5 * np.sum(t * t))


class Exponential(DeviceLogPdf):
    """``any(x < 0) ? -Inf : -rate * sum(x)`` -- the README density (reference ``README.md:15``)."""

    density_id = _lib.EXPONENTIAL
    name = "exponential"

    def __init__(self, rate: float = 1.0):
        if not rate > 0:
            raise ValueError("rate must be > 0")
        self.rate = float(rate)

    def params(self):
        return [self.rate]

    def __call__(self, theta):
        x = np.atleast_1d(np.asarray(theta, dtype=np.float64))
        if np.any(x < 0):
            return -math.inf
        return float(-(self.rate * np.sum(x)))

    def finite_rows(self, X):
        X = np.asarray(X, dtype=np.float64)
        return np.all(np.isfinite(X), axis=1) & ~np.any(X < 0, axis=1)


class Rosenbrock(DeviceLogPdf):
    """Chained Rosenbrock ``-sum_{i<N-1}[b (x_{i+1} - x_i^2)^2 + (a - x_i)^2] / scale``; at ``N = 2``
    with the defaults this is the reference's test density (``test/runtests.jl:68``)."""

    density_id = _lib.ROSENBROCK
    name = "rosenbrock"

    def __init__(self, a: float = 1.0, b: float = 100.0, scale: float = 20.0):
        if not scale > 0:
            raise ValueError("scale must be > 0")
        self.a, self.b, self.scale = float(a), float(b), float(scale)

    def params(self):
        return [self.a, self.b, self.scale]

    def check_ndim(self, ndim):
        if ndim < 2:
            raise ValueError("Rosenbrock needs at least 2 dimensions")

    def __call__(self, theta):
        x = np.asarray(theta, dtype=np.float64)
        s = np.sum(self.b * (x[1:] - x[:-1] ** 2) ** 2 + (self.a - x[:-1]) ** 2)
        return float(-(s / self.scale))


class LogNormal(DeviceLogPdf):
    """Independent log-normal per dimension (reference test target ``LogNormal(0,1)``, ``test/runtests.jl:57``)."""

    density_id = _lib.LOGNORMAL
    name = "lognormal"

    def __init__(self, mu: float = 0.0, sigma: float = 1.0):
        if not sigma > 0:
            raise ValueError("sigma must be > 0")
        self.mu, self.sigma = float(mu), float(sigma)

    def params(self):
        return [self.mu, self.sigma]

    def __call__(self, theta):
        x = np.atleast_1d(np.asarray(theta, dtype=np.float64))
        if np.any(~(x > 0)):
            return -math.inf
        lx = np.log(x)
        return float(np.sum(-lx - 0.5 * ((lx - self.mu) / self.sigma) ** 2))

    def finite_rows(self, X):
        X = np.asarray(X, dtype=np.float64)
        return np.all(np.isfinite(X), axis=1) & np.all(X > 0, axis=1)


class MvNormal2(DeviceLogPdf):
    """2-D normal with mean ``mean`` and covariance ``cov`` (reference test target ``test/runtests.jl:62``)."""

    density_id = _lib.MVNORMAL2
    name = "mvnormal2"

    def __init__(self, mean, cov):
        mean = np.asarray(mean, dtype=np.float64)
        cov = np.asarray(cov, dtype=np.float64)
        if mean.shape != (2,) or cov.shape != (2, 2):
            raise ValueError("MvNormal2 needs a 2-vector mean and a 2x2 covariance")
        self.mean, self.cov = mean, cov
        self.prec = np.linalg.inv(cov)

    def params(self):
        P = self.prec
        return [float(self.mean[0]), float(self.mean[1]), float(P[0, 0]), float(0.5 * (P[0, 1] + P[1, 0])), float(P[1, 1])]

    def check_ndim(self, ndim):
        if ndim != 2:
            raise ValueError("MvNormal2 is 2-dimensional")

    def __call__(self, theta):
        p = self.params()
        d0, d1 = float(theta[0]) - p[0], float(theta[1]) - p[1]
        return -0.5 * (p[2] * d0 * d0 + 2.0 * p[3] * d0 * d1 + p[4] * d1 * d1)


class ExprDensity(DeviceLogPdf):
    """A user-supplied log-density, compiled at run time (hiprtc) into the same HIP kernels -- the
    device-side answer to the reference's arbitrary ``pdf`` closure (``src/samplers.jl:257``)::

        log p(x) = sum_d  TERM(x_d)  +  sum_{d < n-1}  PAIR(x_d, x_{d+1})

    ``term`` and ``pair`` are C expressions of type double.  ``term`` may use ``x`` (= x_d), ``d``,
    ``n`` (= ndim) and ``p`` (``params``, up to 6 doubles); ``pair`` may use ``x`` (= x_d), ``y``
    (= x_{d+1}), ``d``, ``n``, ``p``.  A term may evaluate to ``-INFINITY`` to reject a proposal.
    Examples: ``ExprDensity("-0.5*x*x")`` (standard normal per dimension);
    ``ExprDensity("d < n-1 ? -(1-x)*(1-x)/20 : 0.0", "-100*(y-x*x)*(y-x*x)/20")`` (the reference's
    Rosenbrock test density, ``test/runtests.jl:68``, chained to any ndim).

    Calling the object on a host ``theta`` (what ``make_theta0s`` does) evaluates it on the device.
    """

    density_id = _lib.USER_DENSITY
    name = "expr"

    def __init__(self, term: str, pair: str | None = None, params=()):
        import ctypes as C
        if len(params) > 6:
            raise ValueError("at most 6 parameters")
        self.term, self.pair = str(term), (str(pair) if pair else None)
        self._params = [float(v) for v in params]
        self._L = _lib.lib()
        h = C.c_void_p()
        _lib.check(self._L.kmc_user_density_create(self.term.encode(), self.pair.encode() if self.pair else None,
                                                   C.byref(h)))
        self.user_handle = h

    def __del__(self):
        try:
            if self.user_handle is not None:
                self._L.kmc_user_density_destroy(self.user_handle)
                self.user_handle = None
        except Exception:
            pass

    def params(self):
        return list(self._params)

    def __repr__(self):
        return f"ExprDensity(term={self.term!r}, pair={self.pair!r}, params={self._params})"

    def _eval_rows(self, X):
        import ctypes as C
        X = np.ascontiguousarray(np.asarray(X, dtype=np.float64))
        n, nd = X.shape
        cfg = _lib.Config()
        cfg.dtype, cfg.density = _lib.F64, self.density_id
        for i, v in enumerate(self._params):
            cfg.params[i] = v
        cfg.nwalkers, cfg.ndim, cfg.nthin, cfg.a_scale = max(2, n + (n & 1)), nd, 1, 2.0
        cfg.user_density = self.user_handle
        out = np.empty(n)
        dp = C.POINTER(C.c_double)
        _lib.check(self._L.kmc_logpdf_eval_host(C.byref(cfg), X.ctypes.data_as(dp), out.ctypes.data_as(dp), n))
        return out

    def __call__(self, theta):
        return float(self._eval_rows(np.atleast_1d(np.asarray(theta, dtype=np.float64))[None, :])[0])

    def finite_rows(self, X):
        return self._eval_rows(X) > -np.inf


class CDensity(ExprDensity):
    """ANY log-density you can write as a C++ function body, compiled at run time (hiprtc) into the device kernels::

        double logpdf(const double* x, int n, const double* p) { BODY }

    over the whole proposal ``x[0..n-1]`` (``n`` = ndim, ``p`` = ``params``, up to 6 doubles): arbitrary coupling between
    dimensions, loops, locals; ``return -INFINITY;`` rejects.  Example (a banana in any dimension)::

        CDensity("double s = 0; for (int i = 0; i + 1 < n; ++i) { double d = x[i+1] - x[i]*x[i]; s += p[1]*d*d + (p[0]-x[i])*(p[0]-x[i]); } "
                 "return -s / p[2];", params=[1, 100, 20])

    Its rows travel lane-striped like a menu density's and only the evaluation is per walker (about 3/4 of a menu
    density's rate at 65 536 x 32), orders of magnitude faster than a host callable (:class:`HostLogPdf`).  ``kmc_user_density_create_body``.
    A body that IS a sum over elements -- ``double s = 0; for (int i = 0; i < n; ++i) s += f(x[i]); return g(s);`` (or ``i + 1 < n``
    with ``x[i + 1]``, like the example above) -- is recognised (:attr:`separable`) and runs in the lane-striped kernels at the menu densities' rate.

    ``nblob=m`` makes it the reference's ``pdf(theta) -> (p, blob)`` of ``hasblob=true`` (``src/samplers.jl:150-151, :194-196``)
    on the device: the body is then that of ``double logpdf(const double* x, int n, const double* p, double* blob)`` and fills
    ``blob[0..m)`` (zero on entry); the sampler carries each walker's current blob next to its log-pdf and ``emcee(...,
    hasblob=True)`` returns ``blobs[w][k]`` as an array ``[nwalkers, nsamples, m]`` (or whatever ``init_blobs`` / ``reduce_blob``
    build from that series) -- without the host round trip per half-step a Python callable costs.
    """

    name = "cbody"

    def __init__(self, body: str, params=(), nblob: int = 0):
        import ctypes as C
        if len(params) > 6:
            raise ValueError("at most 6 parameters")
        self.body = str(body)
        self.term, self.pair = None, None
        self._params = [float(v) for v in params]
        self.nblob = int(nblob)
        self._L = _lib.lib()
        h = C.c_void_p()
        if self.nblob > 0:
            _lib.check(self._L.kmc_user_density_create_body_blob(self.body.encode(), self.nblob, C.byref(h)))
        else:
            _lib.check(self._L.kmc_user_density_create_body(self.body.encode(), C.byref(h)))
        self.user_handle = h

    @property
    def separable(self) -> bool:
        """Whether the library recognised the body as a sum over elements (then the samplers stripe its rows over lanes).  The first
        sampler over the density checks the generated per-element form against the body on test rows; if they differ (or no test row
        has a finite value) this turns False and the body is evaluated per walker, as written."""
        return bool(self._L.kmc_user_density_is_separable(self.user_handle))

    def __repr__(self):
        return f"CDensity({self.body!r}, params={self._params}" + (f", nblob={self.nblob})" if self.nblob else ")")

    def eval_with_blobs(self, X):
        """``(logp [n], blobs [n, nblob])`` of the rows of ``X`` (evaluated on the device): ``pdf.(theta0s)`` of ``src/samplers.jl:209-210``."""
        import ctypes as C
        if self.nblob <= 0:
            raise ValueError("this density returns no blobs (CDensity(..., nblob=m))")
        X = np.ascontiguousarray(np.atleast_2d(np.asarray(X, dtype=np.float64)))
        n, nd = X.shape
        cfg = _lib.Config()
        cfg.density = self.density_id
        p = list(self.params()) + [0.0] * 8
        for i in range(8):
            cfg.params[i] = float(p[i])
        cfg.nwalkers, cfg.ndim, cfg.nthin, cfg.a_scale = max(2, n + n % 2), nd, 1, 2.0
        cfg.user_density = self.user_handle
        lp, bl = np.empty(n), np.empty((n, self.nblob))
        dp = C.POINTER(C.c_double)
        _lib.check(self._L.kmc_logpdf_blob_eval_host(C.byref(cfg), X.ctypes.data_as(dp), lp.ctypes.data_as(dp), bl.ctypes.data_as(dp), n))
        return lp, bl


class HostLogPdf(DeviceLogPdf):
    """ANY Python callable as the log-density -- the reference's ``pdf`` closure
    (``src/samplers.jl:257``) kept on the host.  The stretch move, the random draws, the accept test,
    the counters and the sample storage stay on the GPU; per half-step the device hands the batch of
    proposals to ``fn`` and takes their log-pdfs back (``KMC_HOST_DENSITY`` in
    ``include/kissmcmc_hip.h``).  Bound by ``fn`` and one PCIe round trip per half-step, so this is
    the *general* route, not the fast one -- prefer a menu density or :class:`ExprDensity` when the
    density can be written as one.

    ``fn(theta)`` gets one walker: a float when ``scalar`` (the reference's 1-D convention, where
    ``theta0s`` is a vector of numbers) else a 1-D array.  With ``vectorized=True`` it gets the whole
    batch ``[nrows, ndim]`` and must return ``nrows`` log-pdfs.

    ``hasblob=True``: ``fn`` returns ``(p, blob)`` (``src/samplers.jl:150-151``; vectorized: ``(ps, blobs)``
    with one blob per row).  The blobs of the batch evaluated last are kept in ``last_blobs`` and the
    device reports each half-step's accept outcomes to ``on_accepted(accepted, row0, generation, stored)``
    (``kmc_config.host_accepted``), which is how :func:`kissmcmc_jl_amd.emcee` carries the reference's
    ``blob0s`` / ``reduce_blob!`` (``:264, :270``) on the host.
    """

    density_id = _lib.HOST_DENSITY
    name = "host"

    def __init__(self, fn, vectorized: bool = False, scalar: bool = False, hasblob: bool = False):
        if not callable(fn):
            raise TypeError("pdf must be callable")
        self.fn, self.vectorized, self.scalar, self.hasblob = fn, bool(vectorized), bool(scalar), bool(hasblob)
        self.error = None          # exception raised by fn inside the C callback, re-raised by Sampler
        self.last_blobs = None     # hasblob: blobs of the rows evaluated last, in row order
        self.on_accepted = None    # hasblob: callable(accepted uint8[nrows], row0, generation, stored)
        self.c_accepted = None
        if self.hasblob:
            def _acc(flags, nrows, row0, generation, stored, _user):
                try:
                    if self.on_accepted is not None:
                        self.on_accepted(np.ctypeslib.as_array(flags, shape=(nrows,)), int(row0), int(generation), bool(stored))
                    return 0
                except BaseException as e:  # never unwind through the C frames
                    self.error = e
                    return 1

            self.c_accepted = _lib.HOST_ACCEPTED_FN(_acc)

        def _cb(rows, nrows, ndim, out, _user):
            try:
                X = np.ctypeslib.as_array(rows, shape=(nrows, ndim))
                np.ctypeslib.as_array(out, shape=(nrows,))[:] = self.eval_rows(X)
                return 0
            except BaseException as e:      # never unwind through the C frames
                self.error = e
                return 1

        self.c_callback = _lib.HOST_LOGPDF_FN(_cb)

    def params(self):
        return []

    def eval_rows(self, X):
        X = np.asarray(X, dtype=np.float64)
        if self.vectorized:
            r = self.fn(X[:, 0] if self.scalar else X)
            if self.hasblob:
                r, blobs = r
                blobs = list(blobs)
                if len(blobs) != X.shape[0]:
                    raise ValueError(f"vectorized pdf returned {len(blobs)} blobs for {X.shape[0]} rows")
                self.last_blobs = blobs
            r = np.asarray(r, dtype=np.float64).reshape(-1)
            if r.shape[0] != X.shape[0]:
                raise ValueError(f"vectorized pdf returned {r.shape[0]} values for {X.shape[0]} rows")
            return r
        args = (float(v) for v in X[:, 0]) if self.scalar else (row.copy() for row in X)
        if self.hasblob:
            res = [self.fn(t) for t in args]                   # p1, blob1 = pdf(theta1)  :257
            self.last_blobs = [b for _, b in res]
            return np.array([float(p) for p, _ in res], dtype=np.float64)
        return np.array([float(self.fn(t)) for t in args], dtype=np.float64)

    def __call__(self, theta):
        r = self.fn(theta)
        return float(r[0] if self.hasblob else r)

    def finite_rows(self, X):
        return self.eval_rows(X) > -np.inf

    def __repr__(self):
        return f"HostLogPdf({getattr(self.fn, '__name__', type(self.fn).__name__)})"
