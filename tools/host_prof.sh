#!/bin/bash
# Host-time profile of pie_scan_batch_begin / finish: builds the library once more with -DPIE_HOST_PROF (per-section clock_gettime
# totals, printed when a context is destroyed) and runs the lanes probe's single-lane loop on it.
# usage (on the GPU box): tools/host_prof.sh [rows] [lanes]
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_build
[ -f tools/_build/libpie_hip_prof.so ] || hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -DPIE_HOST_PROF -o tools/_build/libpie_hip_prof.so \
    sph-pie_amd/csrc/pie_scan.hip sph-pie_amd/csrc/pie_comm.hip -ldl
PIE_HIP_LIB=$PWD/tools/_build/libpie_hip_prof.so PIE_BATCH_LANES=${2:-1} python3 tools/lanes_probe.py 64 "${1:-12500000}"
