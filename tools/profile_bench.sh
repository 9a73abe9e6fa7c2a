set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
cd $R
python bench.py --steps 2000 --warmup 200 > gpurun_out/prof/bench.json 2> gpurun_out/prof/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/stats -- python bench.py --steps 300 --warmup 50 --no-cpu-baseline > gpurun_out/prof/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/prof/pmc_fetch -- python bench.py --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/prof/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/prof/pmc_write -- python bench.py --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/prof/pmc_write.log 2>&1
find gpurun_out/prof -type f | head -50
for f in $(find gpurun_out/prof -name "*.csv" | head -12); do echo "== $f"; head -3 $f | cut -c1-400; done
