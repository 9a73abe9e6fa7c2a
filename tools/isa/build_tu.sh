#!/bin/bash
# usage: tools/isa/build_tu.sh [extra hipcc flags]  ->  /tmp/isa/cp.s (device assembly with sched_barrier comments)
set -e
REPO=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p /tmp/isa
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off "$@" -I"$REPO/include" -I"$REPO/mppi-generic_amd/csrc" \
  -S --cuda-device-only -o /tmp/isa/cp.s "$REPO/tools/isa/cartpole_pipeline_tu.hip"
grep -c "sched_barrier" /tmp/isa/cp.s
