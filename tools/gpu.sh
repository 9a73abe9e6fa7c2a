#!/bin/bash
# usage: tools/gpu.sh <timeout-seconds> '<command run on the MI355X box>'
# builds libmppi_amd.so here first (a stale or unbuildable library would be rebuilt — slowly — or fail on the GPU box)
set -e
cd "$(dirname "$0")/.."
python mppi-generic_amd/buildlib.py > /tmp/mppi_build.log 2>&1 || { grep -m5 -A3 "error" /tmp/mppi_build.log; echo "BUILD FAILED"; exit 1; }
(cd oracle && make -s 2>/dev/null >/dev/null) || true
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
