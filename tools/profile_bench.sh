set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
cd $R
python bench.py --steps 2000 --warmup 200 > gpurun_out/prof/bench.json 2> gpurun_out/prof/bench.err
MPPI_BENCH_NO_TSCAN=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/stats -- python bench.py --steps 300 --warmup 50 --no-cpu-baseline > gpurun_out/prof/stats.log 2>&1
MPPI_BENCH_NO_TSCAN=1 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/prof/pmc_fetch -- python bench.py --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/prof/pmc_fetch.log 2>&1
MPPI_BENCH_NO_TSCAN=1 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/prof/pmc_write -- python bench.py --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/prof/pmc_write.log 2>&1
find gpurun_out/prof -type f | head -50
for f in $(find gpurun_out/prof -name "*.csv" | head -12); do echo "== $f"; head -3 $f | cut -c1-400; done
# matrix / vector pipe utilisation of the rollout kernels (tools/pmc_mfma_summary.py)
MPPI_BENCH_NO_TSCAN=1 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/prof/pmc_mfma -- python bench.py --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/prof/pmc_mfma.log 2>&1
# the elevation-map RACER models: kernel trace of tools/time_workloads.py racer, then the SQ pipe counters
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/racer_stats -- python tools/time_workloads.py racer > gpurun_out/prof/racer_stats.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/prof/racer_pmc -- python tools/time_workloads.py racer > gpurun_out/prof/racer_pmc.log 2>&1
# condensed files for profiles/ (copy the ones to keep)
python tools/stats_summary.py gpurun_out/prof/stats gpurun_out/prof/kernel_stats.csv "python bench.py --steps 300 --warmup 50 --no-cpu-baseline" || true
python tools/pmc_summary.py gpurun_out/prof/pmc_fetch gpurun_out/prof/pmc_write gpurun_out/prof/pmc_hbm_traffic.json || true
python tools/pmc_mfma_summary.py gpurun_out/prof/pmc_mfma gpurun_out/prof/pmc_mfma_valu.json "python bench.py --steps 50 --warmup 10 --no-cpu-baseline" || true
python tools/stats_summary.py gpurun_out/prof/racer_stats gpurun_out/prof/racer_kernel_stats.csv "python tools/time_workloads.py racer" || true
python tools/pmc_mfma_summary.py gpurun_out/prof/racer_pmc gpurun_out/prof/racer_pmc_valu.json "python tools/time_workloads.py racer" || true
rm -rf gpurun_out/prof/stats gpurun_out/prof/pmc_fetch gpurun_out/prof/pmc_write gpurun_out/prof/pmc_mfma gpurun_out/prof/racer_stats gpurun_out/prof/racer_pmc
