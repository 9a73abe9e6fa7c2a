set -x
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r06_n_gputest.log
tail -n 4 gpurun_out/r06_n_gputest.log
bash tools/profile_bench.sh > gpurun_out/r06_n_profile.log 2>&1
tail -n 5 gpurun_out/r06_n_profile.log
ls gpurun_out/prof
