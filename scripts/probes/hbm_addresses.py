"""Which of a sampler's allocations decides the period of the 2 097 152 x 32 launch (97.8 or 105.5 us)?  -DKMC_PROBE build (it prints where every allocation of 1 MiB or more
lands).  One process, the rows bound into ONE torch arena throughout; before every sampler a dummy allocation of another size is made and kept, so that the sampler's own
buffers -- the per-walker block {logp, naccept, klast} (32 MiB), the moment accumulators (8 MiB) -- land somewhere else each time.  Prints period and addresses per sampler.
    python scripts/probes/hbm_addresses.py [moments 0/1]"""
import os
import re
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np
    import torch
    import kissmcmc_jl_amd as kmc
    mom = bool(int(sys.argv[2]))
    NW, ND, G = 2097152, 32, 200
    arena = torch.zeros((512 << 20) // 8, dtype=torch.float64, device="cuda")
    sys.stderr.write(f"ARENA {arena.data_ptr():#x}\n")
    keep = []
    rng = np.random.default_rng(int.from_bytes(os.urandom(4), "little"))
    for i in range(24):
        if i % 2 == 1:
            keep.append(torch.empty(int(rng.integers(3, 90)) << 20, dtype=torch.uint8, device="cuda"))      # (kept: the next buffers land elsewhere)
        sys.stderr.write(f"SAMPLER {i}\n"); sys.stderr.flush()
        with kmc.Sampler(kmc.GaussianIso(), NW, ND, 2 * G + 64, 64, 1, 2.0, 12345, moments=mom) as s:
            s.bind_positions(arena.data_ptr())
            s.init_ball(np.zeros(ND), np.ones(ND), seed=12345)
            s.run(64); s.sync()
            s.run(G); s.sync()
            sys.stderr.write(f"PERIOD {s.last_run_ms() * 1e3 / (2 * G):.2f}\n"); sys.stderr.flush()
    sys.exit(0)
mom = sys.argv[1] if len(sys.argv) > 1 else "0"
env = dict(os.environ, KMC_LIB_PATH=os.path.join(ROOT, "kissmcmc.jl_amd", "libkmc_var_probe.so"))
r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", mom], env=env, capture_output=True, text=True)
cur = []
for line in r.stderr.splitlines():
    if line.startswith("ARENA"):
        print(line)
    elif line.startswith("SAMPLER"):
        cur = []
    elif "alloc" in line and "bytes at" in line:
        m = re.search(r"alloc (\d+) bytes at (0x[0-9a-f]+)", line)
        cur.append((int(m.group(1)), int(m.group(2), 16)))
    elif line.startswith("PERIOD"):
        print(line.split()[1].rjust(8) + " us   " + "  ".join(f"{b >> 20} MiB @ {p:#x}" for b, p in cur if b < (256 << 20)))
if r.returncode != 0:
    print(r.stderr[-2000:])
