# device side of the closed loop with the split hand-over: kernel trace of tools/closed_loop_trace.py, per call the rollout,
# merge, control-phase and trajectory-phase kernels with their start / end relative to the call's rollout start
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/closed_loop_trace.py single > gpurun_out/closed_loop_host.log 2>&1
python tools/closed_loop_trace.py split >> gpurun_out/closed_loop_host.log 2>&1
cat gpurun_out/closed_loop_host.log
rm -rf gpurun_out/cl_trace
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/cl_trace -- python tools/closed_loop_trace.py split 300 > gpurun_out/cl_trace.log 2>&1
f=$(find gpurun_out/cl_trace -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, json, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
seq = [(r['Kernel_Name'].split('(')[0][:60], int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Queue_Id', '')) for r in rows]
tail = seq[-60:]
t0 = tail[0][1]
for k, s, e, q in tail:
    print("%-62s q=%s start %8.2f end %8.2f dur %6.2f" % (k, q, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
PY
