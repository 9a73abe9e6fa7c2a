# HBM traffic and duration of the Robust MPPI rollout kernels (AutoRally-NN K=16384 T=150; double integrator K=8192 T=150):
# rocprofv3 kernel trace + the two PMC passes over tools/robust_latency.py -> gpurun_out/robust_traffic.json
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/rt_*
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/rt_stats -- python tools/robust_latency.py > gpurun_out/rt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/rt_fetch -- python tools/robust_latency.py >> gpurun_out/rt.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/rt_write -- python tools/robust_latency.py >> gpurun_out/rt.log 2>&1
python tools/stats_summary.py gpurun_out/rt_stats gpurun_out/robust_kernel_stats.csv "python tools/robust_latency.py"
python tools/pmc_summary.py gpurun_out/rt_fetch gpurun_out/rt_write gpurun_out/robust_pmc_hbm_traffic.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/robust_pmc_hbm_traffic.json"))
for k, v in d["kernels"].items():
    if "RMPPI" in k or "initEval" in k:
        print(k[:80], {kk: round(vv / 1e6, 2) if isinstance(vv, float) else vv for kk, vv in v.items() if "bytes" in kk})
PY
grep -v "^#" gpurun_out/robust_kernel_stats.csv | cut -d, -f1,3,13 | cut -c1-120 | head -8
rm -rf gpurun_out/rt_stats gpurun_out/rt_fetch gpurun_out/rt_write
