cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/fin
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fin -- python tools/compute_control_latency.py > gpurun_out/fin.log 2>&1
f=$(find gpurun_out/fin -name "*kernel_stats.csv" | head -1)
cut -d, -f1-4 $f | cut -c1-120 | head -6
