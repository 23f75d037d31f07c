"""Inputs of the recogniser's sanitizer harness (tests/sanitize/recognise_fuzz.cpp): the grammar of the round-4 GPU fuzz (profiles/r04_recogniser_fuzz.txt:
honest per-element bodies, tag H, and bodies with one statement that carries state between elements, tag S) plus malformed text (tag M): unbalanced
braces and parentheses, very long lines, NUL and non-ASCII bytes, truncations and byte mutations of honest bodies."""
import numpy as np

HONEST_ELEM = ["x[i]", "(x[i] - w)", "(x[i] * c)", "(x[i] - p[0])", "fabs(x[i])", "(x[i] / n)", "((i + 1) * 0.1 * x[i])", "tanh(x[i])"]
HONEST_NEXT = ["x[i + 1]", "(x[i+1] - x[i])", "(x[1 + i] * c)", "(x[i + 1] - w)"]
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
        stmts.append([f"{a} += 0.1 * {e1} * {e2};", f"{{ const double d = {e1}; {a} += 0.05 * d * d; }}",
                      f"if ({e1} > 0.0) {a} += 0.1 * {e2}; else {a} += -0.05 * {e2};", f"{a} += ({e1} > {e2}) ? 0.01 : 0.02;"][kind])
    if stateful:
        stmts.append("t += 0.1 * x[i];")                 # (so that `t` IS a running sum wherever a stateful statement reads it)
        stmts.insert(int(rng.integers(1, len(stmts) + 1)), str(rng.choice(STATEFUL)))
    braces = len(stmts) > 1 or rng.integers(0, 2)
    loop = f"for (int i = 0; {cond}; {inc}) " + ("{ " + " ".join(stmts) + " }" if braces else stmts[0])
    ret = "return -0.5 * s" + "".join(f" - 0.01 * {a} * {a}" for a in accs[1:]) + ";"
    return " ".join(pre) + " " + decl + " " + loop + " " + ret, stateful


def records(nbodies=400, nmutants=1500, seed=1):
    """[(tag, bytes)]"""
    rng = np.random.default_rng(seed)
    out, honest = [], []
    for _ in range(nbodies):
        body, stateful = make_body(rng)
        out.append(("S" if stateful else "H", body.encode()))
        if not stateful:
            honest.append(body.encode())
    good = b"double s = 0; for (int i = 0; i < n; ++i) s += x[i] * x[i]; return -0.5 * s;"
    out += [("M", b""), ("M", b"{"), ("M", b"}" * 300), ("M", b"(" * 4000), ("M", b"for (int i = 0; i < n; ++i) {" * 100),
            ("M", good.replace(b"s += x[i] * x[i];", b"{ s += x[i] * x[i];")),                     # unbalanced brace
            ("M", good.replace(b"return", b"retur")), ("M", good[:-1]), ("M", good + b"}" * 50),
            ("M", b"double s = 0; for (int i = 0; i < n; ++i) s += " + b"x[i] * " * 600 + b"x[i]; return s;"),       # long, under the 4096 limit
            ("M", b"double s = 0; /* " + b"*" * 3000 + b" for (int i = 0; i < n; ++i) s += x[i]; return s;"),          # unterminated comment
            ("M", b"a" * 100000), ("M", b"double s = 0; for (int i = 0; i < n; ++i) s += x[i]; return s;" + b" " * 100000),
            ("M", good.replace(b"x[i] *", b"x[i]\x00*")), ("M", b"\x00" * 64), ("M", bytes(range(256)) * 4),
            ("M", b"for (int \xc3\xa9 = 0; \xc3\xa9 < n; ++\xc3\xa9) s += x[\xc3\xa9]; return s;"),
            ("M", b"double s = 0; for (int i = 0; i < n; ++i) s += x[" + b"[" * 1000 + b"i]; return s;"),
            ("M", b"double s = 0; for (int i = 0; i < n; ++i) if (" + b"(" * 1500 + b"x[i]" + b")" * 1400 + b" s += 1; return s;")]
    for _ in range(nmutants):                                                                        # truncations and byte mutations of honest bodies
        b = bytearray(honest[int(rng.integers(0, len(honest)))])
        kind = int(rng.integers(0, 4))
        if kind == 0:
            b = b[: int(rng.integers(0, len(b)))]
        elif kind == 1:
            for _ in range(int(rng.integers(1, 6))):
                b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        elif kind == 2:
            i = int(rng.integers(0, len(b)))
            b[i:i] = bytes(rng.choice(list(b"{}()[];=+&*#\\\"'/"), size=int(rng.integers(1, 20))).tolist())
        else:
            i, j = sorted(int(v) for v in rng.integers(0, len(b), size=2))
            del b[i:j]
        out.append(("M", bytes(b)))
    return out


def serialise(recs) -> bytes:
    return b"".join(f"{tag} {len(b)}\n".encode() + b + b"\n" for tag, b in recs)
