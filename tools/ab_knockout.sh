#!/bin/bash
# What the AutoRally step loop costs without its MFMAs / without its tanh (A/B builds; run tools/gpu.sh 600 'bash tools/ab_knockout.sh')
# build the variants first (CPU):  python mppi-generic_amd/buildlib.py --variant nomfma autorally_nn.hip -DMPPI_KNOCKOUT_MFMA
#                                  python mppi-generic_amd/buildlib.py --variant notanh autorally_nn.hip -DMPPI_KNOCKOUT_TANH
for v in "" _nomfma _notanh; do
  echo "== variant '$v'"
  MPPI_AMD_LIB=$PWD/mppi-generic_amd/lib/libmppi_amd$v.so python tools/time_workloads.py autorally 2>&1 | grep "64,4"
done
