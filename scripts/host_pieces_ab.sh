#!/bin/bash
# Host-evaluated density route: a half-step's proposals in one piece vs. in pieces behind events (KMC_DEBUG=host-pieces=n)
R=${GRAFT_REPO_ROOT:-/root/repo}
for p in 1 2 4 8; do echo "== KMC_DEBUG=host-pieces=$p"; KMC_DEBUG=host-pieces=$p python3 $R/scripts/host_density_bench.py 2>/dev/null | grep "numpy batch closure"; done
