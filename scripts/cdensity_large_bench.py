"""A user-written density (CDensity function body) against the menu density it restates, at large shapes (us per half-step)."""
import sys
sys.path.insert(0, '.')
import numpy as np
import kissmcmc_jl_amd as kmc

gauss_body = "double s = 0; for (int i = 0; i < n; ++i) s += x[i] * x[i]; return -0.5 * s;"
rosen_body = "double s = 0; for (int i = 0; i + 1 < n; ++i) { double a = 1.0 - x[i], b = x[i+1] - x[i]*x[i]; s += a*a + 100.0*b*b; } return -s / 20.0;"
for name, menu, body, nw, nd, G, scale in (("gaussian 65536 x 32", kmc.GaussianIso(), gauss_body, 65536, 32, 512, 1.0),
                                            ("rosenbrock 16384 x 64", kmc.Rosenbrock(), rosen_body, 16384, 64, 512, 0.1),
                                            ("gaussian 65536 x 8", kmc.GaussianIso(), gauss_body, 65536, 8, 512, 1.0),
                                            ("gaussian 8192 x 128", kmc.GaussianIso(), gauss_body, 8192, 128, 256, 1.0)):
    th = scale * np.random.default_rng(1).standard_normal((nw, nd))
    row = []
    for pdf in (menu, kmc.CDensity(body), kmc.ExprDensity("-0.5*x*x") if "gauss" in name else None):
        if pdf is None:
            row.append(float("nan")); continue
        with kmc.Sampler(pdf, nw, nd, 2 * G, 0, 1, 2.0, 3, moments=(len(sys.argv) < 2 or sys.argv[1] != "nomom")) as s:
            s.set_positions(th)
            s.run(G); s.sync()
            s.run(G); s.sync()
            row.append(1e3 * s.last_run_ms() / (2 * G))
            how = s.describe()
    print(f"{name:24s}: menu {row[0]:.3f}  CDensity {row[1]:.3f}  ExprDensity {row[2]:.3f} us per half-step   [{how[:70]}]", flush=True)
