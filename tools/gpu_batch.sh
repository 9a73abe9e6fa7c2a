set -x
mkdir -p gpurun_out
python tools/ab_kernels.py cartpole autorally racer robust_ar robust_racer --json gpurun_out/r06_f_ab_product.json > gpurun_out/r06_f_ab_product.txt 2>&1
for i in 1 2; do
MPPI_AMD_LIB=$PWD/mppi-generic_amd/lib/libmppi_amd_ssr0.so python tools/ab_kernels.py robust_ar > gpurun_out/r06_f_ab_ssr0_$i.txt 2>&1
python tools/ab_kernels.py robust_ar > gpurun_out/r06_f_ab_ssr1_$i.txt 2>&1
done
bash tools/robust_traffic.sh > gpurun_out/r06_f_robust_traffic_ssr1.txt 2>&1
cp gpurun_out/robust_pmc_hbm_traffic.json gpurun_out/r06_f_robust_pmc_ssr1.json; cp gpurun_out/robust_kernel_stats.csv gpurun_out/r06_f_robust_kernel_stats_ssr1.csv
MPPI_AMD_LIB=$PWD/mppi-generic_amd/lib/libmppi_amd_ssr0.so bash tools/robust_traffic.sh > gpurun_out/r06_f_robust_traffic_ssr0.txt 2>&1
cp gpurun_out/robust_pmc_hbm_traffic.json gpurun_out/r06_f_robust_pmc_ssr0.json
cat gpurun_out/r06_f_ab_product.txt gpurun_out/r06_f_ab_ssr*.txt
tail -n 8 gpurun_out/r06_f_robust_traffic_ssr1.txt gpurun_out/r06_f_robust_traffic_ssr0.txt
