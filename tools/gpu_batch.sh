set -x
mkdir -p gpurun_out
./examples/_build/exec_width > gpurun_out/r06_b_exec_width.txt 2>&1
python tools/ab_kernels.py --json gpurun_out/r06_b_ab_product.json > gpurun_out/r06_b_ab_product.txt 2>&1
MPPI_AMD_LIB=$PWD/mppi-generic_amd/lib/libmppi_amd_kr.so python tools/ab_kernels.py --json gpurun_out/r06_b_ab_kr.json > gpurun_out/r06_b_ab_kr.txt 2>&1
for v in 0 1; do HIP_FORCE_DEV_KERNARG=$v python tools/ab_kernels.py cartpole autorally > gpurun_out/r06_b_devkernarg_$v.txt 2>&1; done
MPPI_AMD_LIB=$PWD/mppi-generic_amd/lib/libmppi_amd_kr.so timeout 900 python -m pytest tests/test_full_size_parity.py tests/test_rmppi.py tests/test_streamed_merge.py tests/test_kernarg_layout.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r06_b_kr_parity.log
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r06_b_gputest.log
tail -5 gpurun_out/r06_b_kr_parity.log gpurun_out/r06_b_gputest.log
cat gpurun_out/r06_b_ab_product.txt gpurun_out/r06_b_ab_kr.txt gpurun_out/r06_b_exec_width.txt gpurun_out/r06_b_devkernarg_*.txt
