# kernel trace of mppi_compute_control (Cartpole K=16384, T=100) calls from an idle stream: where the call's ~42 us go on the
# device — ingest, rollout, merge, finalize (start of the kernel to the end), the gaps between them, and what is left for the
# host side (launch-to-start latency of the first kernel + the flag's way back).  Writes gpurun_out/compute_control_trace.json
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/cc_trace
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/cc_trace -- python tools/compute_control_latency.py idle > gpurun_out/cc_trace.log 2>&1
f=$(find gpurun_out/cc_trace -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, json, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
seq = [(r['Kernel_Name'], int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows]
calls = []
i = 0
while i + 3 < len(seq):
    a, b, c, d = seq[i:i + 4]
    if 'ingestKernel' in a[0] and 'rolloutPipelineKernel' in b[0] and 'combineKernel' in c[0] and 'finalizeKernel' in d[0]:
        calls.append((a, b, c, d))
        i += 4
    else:
        i += 1
calls = calls[len(calls) // 2:]  # the idle-stream loop is the last one the tool runs
out = collections.OrderedDict()
def med(v):
    v = sorted(v)
    return round(v[len(v) // 2] / 1e3, 2)
out['calls'] = len(calls)
out['ingest_us'] = med([a[2] - a[1] for a, b, c, d in calls])
out['gap_ingest_rollout_us'] = med([b[1] - a[2] for a, b, c, d in calls])
out['rollout_us'] = med([b[2] - b[1] for a, b, c, d in calls])
out['gap_rollout_merge_us'] = med([c[1] - b[2] for a, b, c, d in calls])
out['merge_us'] = med([c[2] - c[1] for a, b, c, d in calls])
out['gap_merge_finalize_us'] = med([d[1] - c[2] for a, b, c, d in calls])
out['finalize_whole_us'] = med([d[2] - d[1] for a, b, c, d in calls])
out['ingest_start_to_finalize_start_us'] = med([d[1] - a[1] for a, b, c, d in calls])
out['ingest_start_to_finalize_end_us'] = med([d[2] - a[1] for a, b, c, d in calls])
print(json.dumps(out, indent=1))
json.dump(out, open('gpurun_out/compute_control_trace.json', 'w'), indent=1)
PY
tail -3 gpurun_out/cc_trace.log
