"""Random function bodies against the recogniser of sums over elements (kmc_rtc.hip: recognise_separable) and the check behind it
(kmc_sampler.hip: check_sum_form).  Every body is built from a small grammar -- honest per-element statements mixed with statements
that carry state between elements or read the proposal at other indices -- and, when the recogniser takes it, judged three ways:
  (1) did check_sum_form agree (the density stays `separable` after the first sampler)?  A body it un-routes with "disagrees" is a
      FALSE POSITIVE of the text rules (caught, but worth a rule);
  (2) a body whose grammar tag says "carries state" must never stay routed;
  (3) the routed run must equal the run of the same body evaluated per walker (KMC_DEBUG=no-body-routing): positions and
      acceptance counters bit for bit, log-pdfs to rounding.
python scripts/exp/recogniser_fuzz.py [nbodies] [seed]     (GPU box; ~1.5 s per body: three hiprtc compiles)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
os.environ.setdefault("KMC_NO_RESIDENT", "1")
os.environ.setdefault("KMC_CACHE_DIR", "off")

HONEST_ELEM = ["x[i]", "(x[i] - w)", "(x[i] * c)", "(x[i] - p[0])", "fabs(x[i])", "(x[i] / n)", "((i + 1) * 0.1 * x[i])", "tanh(x[i])"]
HONEST_NEXT = ["x[i + 1]", "(x[i+1] - x[i])", "(x[1 + i] * c)", "(x[i + 1] - w)"]
# statements that make the loop NOT a sum of per-element terms (tag: stateful)
STATEFUL = ["c = c * 0.9;", "k = k + 1;", "s += 1e-3 * t;", "t += x[(i + 1) % n];", "i = i;", "{ double q = (s += 1e-3); t += q; }",
            "t += x[i > 0 ? i - 1 : 0];", "t += x[0] * x[i];", "{ int i = 0; t += x[i]; }", "t += (c = -c) * x[i];", "t += (k++) * 1e-3;",
            "if (t > 1.0) t += x[i];", "t += modf(x[i] + c, &c);", "t += *(x + i);", "t = t + x[i];", "w2 = x[i]; t += w2;"]


def make_body(rng):
    pair = bool(rng.integers(0, 2))
    elems = HONEST_ELEM + (HONEST_NEXT if pair else [])
    stateful = rng.random() < 0.4
    nacc = int(rng.integers(1, 4)) if not stateful else int(rng.integers(2, 4))
    accs = ["s", "t", "u"][:nacc]
    pre = ["const double w = p[0];", "double c = 0.5;", "int k = 2;", "double w2 = 0.0;"]
    decl = ("double " + ", ".join(f"{a} = 0" for a in accs) + ";") if rng.integers(0, 2) else " ".join(f"double {a} = 0.0;" for a in accs)
    cond = {0: "i < n", 1: "i + 1 < n", 2: "i < n - 1"}[int(rng.integers(1, 3)) if pair else 0]
    inc = str(rng.choice(["++i", "i++"]))
    stmts = ["s += x[i] * x[i];"]
    for _ in range(int(rng.integers(0, 4))):
        a = str(rng.choice(accs))
        e1, e2 = (str(rng.choice(elems)) for _ in range(2))
        kind = int(rng.integers(0, 4))
        if kind == 0:
            stmts.append(f"{a} += 0.1 * {e1} * {e2};")
        elif kind == 1:
            stmts.append(f"{{ const double d = {e1}; {a} += 0.05 * d * d; }}")
        elif kind == 2:
            stmts.append(f"if ({e1} > 0.0) {a} += 0.1 * {e2}; else {a} += -0.05 * {e2};")
        else:
            stmts.append(f"{a} += ({e1} > {e2}) ? 0.01 : 0.02;")
    if stateful:
        stmts.append("t += 0.1 * x[i];")                 # (so that `t` IS a running sum wherever a stateful statement reads it)
        stmts.insert(int(rng.integers(1, len(stmts) + 1)), str(rng.choice(STATEFUL)))
    braces = len(stmts) > 1 or rng.integers(0, 2)
    loop = f"for (int i = 0; {cond}; {inc}) " + ("{ " + " ".join(stmts) + " }" if braces else stmts[0])
    ret = "return -0.5 * s" + "".join(f" - 0.01 * {a} * {a}" for a in accs[1:]) + ";"
    return " ".join(pre) + " " + decl + " " + loop + " " + ret, stateful


def run(kmc, pdf, th, G, seed):
    nw, nd = th.shape
    with kmc.Sampler(pdf, nw, nd, G, 0, 1, 2.0, seed) as s:
        how = s.describe()
        s.set_positions(th)
        s.run(G)
        s.sync()
        return how, s.positions(), s.naccept(), s.logp()


def main():
    nbodies = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    import kissmcmc_jl_amd as kmc
    rng = np.random.default_rng(seed)
    nw, nd, G = 256, 8, 30
    th = 0.5 * np.random.default_rng(11).standard_normal((nw, nd))
    taken = kept = false_pos = wrong = ncompile_err = nstateful = 0
    for b in range(nbodies):
        if b and b % 25 == 0:
            print(f"... {b} bodies: {taken} taken, {kept} kept, {false_pos} false positives caught, {wrong} wrong", flush=True)
        body, stateful = make_body(rng)
        nstateful += stateful
        os.environ.pop("KMC_DEBUG", None)
        try:
            pdf = kmc.CDensity(body, params=[0.3])
        except kmc.KmcError as e:
            ncompile_err += 1
            print(f"[{b}] does not compile ({str(e).splitlines()[1][:120] if len(str(e).splitlines()) > 1 else e}): {body}", flush=True)
            continue
        if not pdf.separable:
            continue
        taken += 1
        how, pos, nacc, logp = run(kmc, pdf, th, G, 5 + b)
        if not pdf.separable:
            if "disagrees" in how:
                false_pos += 1
                print(f"[{b}] FALSE POSITIVE of the text rules (caught by the check): {body}\n      {how.split('taken for')[-1][:200]}", flush=True)
            continue
        kept += 1
        if stateful:
            wrong += 1
            print(f"[{b}] STATEFUL BODY STAYED ROUTED: {body}", flush=True)
        os.environ["KMC_DEBUG"] = "no-body-routing"
        plain = kmc.CDensity(body, params=[0.3])
        assert not plain.separable
        _, pos2, nacc2, logp2 = run(kmc, plain, th, G, 5 + b)
        os.environ.pop("KMC_DEBUG", None)
        same = np.array_equal(pos, pos2) and np.array_equal(nacc, nacc2) and np.all(np.abs(logp - logp2) <= 1e-11 * np.maximum(1.0, np.abs(logp2)))
        if not same:
            wrong += 1
            print(f"[{b}] ROUTED RUN DIFFERS from the body evaluated per walker: {body}", flush=True)
    print(f"{nbodies} bodies ({nstateful} with a stateful statement, {ncompile_err} not compiling): {taken} taken by the recogniser, {kept} kept after the check, "
          f"{false_pos} false positives of the text rules caught by the check, {wrong} WRONG (stateful kept, or routed run != per-walker run)")
    return 1 if wrong else 0


if __name__ == "__main__":
    sys.exit(main())
